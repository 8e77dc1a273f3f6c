"""The C-ABI library loads and exports every symbol include/pinsage_hip.h declares (no compute:
runs without a GPU), and the product path fails loudly without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pinsage_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    from pinsage_hip import native
    if not native.have_lib():
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(native.SO_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/pinsage_hip.h but not exported"
    assert sorted(native.SYMBOLS) == syms
    lib.ps_error_string.restype = ctypes.c_char_p
    assert lib.ps_abi_version() == 1
    assert lib.ps_error_string(0) == b"ok" and b"invalid" in lib.ps_error_string(-1)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pinsage_hip import native
    from utils.random_walk import RandomWalkSampler
    from model.pinsage import ImportancePooling
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(native.NativeError):
        RandomWalkSampler(ei)
    with pytest.raises(native.NativeError):
        ImportancePooling()(torch.zeros(2, 4), [[0]], [[1.0]])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "movie-recommendation-engine_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(d, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def _header_arity():
    txt = open(os.path.join(ROOT, "include", "pinsage_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for name, params in re.findall(r"\b(ps_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        params = params.strip()
        out[name] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def test_python_call_sites_pass_as_many_arguments_as_the_header_declares():
    """ctypes calls carry no prototypes: a call site that lags behind a signature change would read garbage.  Every
    `nv.call("ps_x", ...)` and `lib.ps_x(...)` in the package (and bench/tools) is checked against the header's arity,
    and so is the C definition in csrc/ (`extern "C" ... ps_x(...)`)."""
    import ast
    arity = _header_arity()
    assert len(arity) >= 25
    checked = 0
    roots = [os.path.join(ROOT, "movie-recommendation-engine_amd"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "bench.py"),
             os.path.join(ROOT, "__graft_entry__.py")]
    files = []
    for r in roots:
        if r.endswith(".py"):
            files.append(r)
        else:
            for d, _, fs in os.walk(r):
                files += [os.path.join(d, f) for f in fs if f.endswith(".py")]
    for f in files:
        tree = ast.parse(open(f).read())
        for node in ast.walk(tree):
            if not isinstance(node, ast.Call):
                continue
            fn = node.func
            if isinstance(fn, ast.Attribute) and fn.attr == "call" and node.args and isinstance(node.args[0], ast.Constant) \
                    and isinstance(node.args[0].value, str) and node.args[0].value in arity:
                name, n = node.args[0].value, len(node.args) - 1
            elif isinstance(fn, ast.Attribute) and fn.attr in arity:
                name, n = fn.attr, len(node.args)
            else:
                continue
            assert not any(isinstance(a, ast.Starred) for a in node.args), (f, name)
            assert n == arity[name], f"{f}: {name} called with {n} arguments, header declares {arity[name]}"
            checked += 1
    assert checked >= 30
    # the definitions
    src = ""
    csrc = os.path.join(ROOT, "movie-recommendation-engine_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith(".hip"):
            src += re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f)).read())
    for name, n in arity.items():
        m = re.search(r'extern\s+"C"\s+[\w\s\*]+?\b' + name + r"\s*\(([^{;]*?)\)\s*\{", src, flags=re.S)
        assert m, f"{name} has no extern \"C\" definition"
        params = m.group(1).strip()
        got = 0 if params in ("", "void") else params.count(",") + 1
        assert got == n, f"{name}: defined with {got} parameters, declared with {n}"
