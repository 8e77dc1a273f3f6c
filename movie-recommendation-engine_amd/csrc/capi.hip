// capi.hip -- version / error strings of the C ABI (include/pinsage_hip.h).
#include "ps_common.h"

extern "C" int ps_abi_version(void) { return 1; }

extern "C" const char *ps_error_string(int code) {
    switch (code) {
        case PS_OK: return "ok";
        case PS_EINVAL: return "invalid argument or unsupported shape";
        case PS_ELAUNCH: return "HIP launch/runtime error";
        case PS_EWORKSPACE: return "workspace too small";
        case PS_EUNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}
