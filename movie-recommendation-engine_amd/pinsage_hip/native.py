"""ctypes binding of libpinsage_hip.so (C ABI: include/pinsage_hip.h).

There is NO fallback: if the shared library is missing every op raises LibraryMissing."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("PS_HIP_LIB") or os.path.join(_HERE, "libpinsage_hip.so")   # PS_HIP_LIB: kernel experiments only
_lib = None

PS_RNG_STREAM = 0
PS_RNG_PHILOX = 1
PS_RNG_STREAM_RAW = 2
PS_RNG_STREAM_WALKS = 3
PS_WALK_HALF_BUCKETS = 0x100
PS_RELU = 1
PS_L2NORM = 2
PS_WPERM = 4
PS_OK, PS_EINVAL, PS_ELAUNCH, PS_EWORKSPACE, PS_EUNSUPPORTED = 0, -1, -2, -3, -4      # status codes (include/pinsage_hip.h)


class LibraryMissing(RuntimeError):
    pass


class NativeError(RuntimeError):
    pass


# every exported symbol of include/pinsage_hip.h (checked by tests/test_abi.py)
SYMBOLS = [
    "ps_abi_version", "ps_error_string", "ps_csr_build_workspace_bytes", "ps_csr_build", "ps_cdf_build",
    "ps_guide_build", "ps_pack_edges", "ps_bucket_build", "ps_bucket_build_half", "ps_dest_info_build", "ps_graph_stats", "ps_walk_sample", "ps_walk_sample_layers", "ps_walk_paths", "ps_uniform_offsets", "ps_mt19937_window_shift", "ps_mt19937_workspace_bytes", "ps_mt19937_chunk_log2", "ps_mt19937_random_sample", "ps_mt19937_raw_stream",
    "ps_importance_pool", "ps_permute_k", "ps_linear", "ps_lsh_encode", "ps_hamming_topk_workspace_bytes", "ps_hamming_topk",
    "ps_lsh_planes_bytes", "ps_lsh_expand", "ps_hamming_topk_mfma_workspace_bytes", "ps_hamming_topk_mfma", "ps_hamming_topk_mfma_codes",
    "ps_topk_merge", "ps_topk_merge_strided", "ps_dot_topk_workspace_bytes", "ps_dot_topk", "ps_l2_topk_workspace_bytes", "ps_l2_topk", "ps_ivf_topk_workspace_bytes", "ps_ivf_topk", "ps_spmm_csr",
]


def have_lib() -> bool:
    return os.path.exists(SO_PATH)


def lib():
    global _lib
    if _lib is None:
        if not have_lib():
            raise LibraryMissing(
                f"{SO_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the PinSage hot path.")
        _lib = C.CDLL(SO_PATH)
        _lib.ps_error_string.restype = C.c_char_p
        for name in ("ps_csr_build_workspace_bytes", "ps_hamming_topk_workspace_bytes", "ps_dot_topk_workspace_bytes",
                     "ps_l2_topk_workspace_bytes", "ps_ivf_topk_workspace_bytes", "ps_mt19937_workspace_bytes", "ps_lsh_planes_bytes",
                     "ps_hamming_topk_mfma_workspace_bytes"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_size_t
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise NativeError(f"{what}: {lib().ps_error_string(rc).decode()} (code {rc})")


class KernelTimer:
    """Optional per-launch HIP-event timing (events are recorded on the stream the kernels are
    launched on, torch's current stream; nothing synchronises until `summary`)."""

    def __init__(self):
        self.events = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.events:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0})
            d["launches"] += 1
            d["ms"] += a.elapsed_time(b)
        for d in out.values():
            d["avg_ms"] = d["ms"] / d["launches"]
        return out


_timer = None


def set_timer(t):
    global _timer
    _timer = t


def call(name, *args):
    """Invoke one C-ABI entry point and raise on a non-zero status."""
    fn = getattr(lib(), name)
    if _timer is None:
        rc = fn(*args)
    else:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        _timer.events.append((name, a, b))
    check(rc, name)


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise NativeError("libpinsage_hip takes device (HBM) pointers; got a CPU tensor")
    if not t.is_contiguous():
        raise NativeError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """torch's current HIP stream of the current device as a C pointer.  (torch.cuda.current_stream() builds a python Stream object
    through three layers of device-index helpers: 9 us a call, 14 calls per step -- a quarter of a sharded step's host time.)"""
    if _raw_stream is not None and _get_device is not None:
        return C.c_void_p(_raw_stream(_get_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    if not torch.cuda.is_available():
        raise NativeError("the PinSage hot path needs an MI355X (torch.cuda is not available); "
                          "there is no CPU fallback")
    lib()
    return torch.device("cuda", torch.cuda.current_device())


i64 = C.c_int64
i32 = C.c_int
u64 = C.c_uint64
u32 = C.c_uint32
