"""ctypes binding of oracle/_build/liboracle.so (the plain-C oracle).  TEST INFRASTRUCTURE
ONLY -- see oracle/pinsage_oracle.c.  `build()` compiles it on demand with gcc."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_SO_OVERRIDE = os.environ.get("PS_ORACLE_SO")          # `make -C oracle asan-test`: the sanitizer build of the same source
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "pinsage_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if _SO_OVERRIDE:
            _lib = C.CDLL(os.path.abspath(_SO_OVERRIDE))
        else:
            build()
            _lib = C.CDLL(_SO)
        _lib.orc_np_sum.restype = C.c_double
    return _lib


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed rc={rc}")


def max_threads():
    return int(lib().orc_max_threads())


def np_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return float(lib().orc_np_sum(_p(a), C.c_int64(a.shape[0])))


class Graph:
    """CSR + CDF on the host (rowptr int64[V+1], col int32[E], w fp64[E], cdf fp64[E])."""

    def __init__(self, edge_index, edge_weights=None, num_nodes=None, threads=1):
        ei = np.ascontiguousarray(edge_index, dtype=np.int64)
        E = ei.shape[1]
        V = int(ei.max()) + 1 if E else 0
        if num_nodes is not None:
            V = max(V, int(num_nodes))
        w = None if edge_weights is None else np.ascontiguousarray(edge_weights, dtype=np.float32)
        self.V, self.E = V, E
        self.rowptr = np.zeros(V + 1, dtype=np.int64)
        self.col = np.zeros(E, dtype=np.int32)
        self.w = np.zeros(E, dtype=np.float64)
        self.cdf = np.zeros(E, dtype=np.float64)
        src, dst = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
        _check(lib().orc_csr_build(_p(src), _p(dst), _p(w), C.c_int64(E), C.c_int64(V),
                                   _p(self.rowptr), _p(self.col), _p(self.w)), "orc_csr_build")
        _check(lib().orc_cdf_build(_p(self.rowptr), _p(self.w), C.c_int64(V), _p(self.cdf), C.c_int(threads)),
               "orc_cdf_build")

    @classmethod
    def from_arrays(cls, rowptr, col, cdf):
        """Wrap an existing CSR + CDF (e.g. copied back from the device) without rebuilding."""
        self = cls.__new__(cls)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.cdf = np.ascontiguousarray(cdf, dtype=np.float64)
        self.w = None
        self.V, self.E = self.rowptr.shape[0] - 1, self.col.shape[0]
        return self

    def uniform_offsets(self, nodes, W, L):
        """Offsets into the numpy stream per start node, valid on sink-free graphs."""
        nodes = np.asarray(nodes, dtype=np.int64)
        active = (self.rowptr[nodes + 1] - self.rowptr[nodes]) > 0
        off = np.zeros(nodes.shape[0], dtype=np.int64)
        off[1:] = np.cumsum(active[:-1].astype(np.int64)) * (W * L)
        return off, int(active.sum()) * W * L


def walk_sample(g: Graph, nodes, T, L=2, W=100, uniforms=None, uoff=None, philox=None, threads=1):
    nodes = np.ascontiguousarray(nodes, dtype=np.int64)
    B = nodes.shape[0]
    ids = np.empty((B, T), dtype=np.int64)
    counts = np.empty((B, T), dtype=np.int32)
    nvalid = np.empty(B, dtype=np.int32)
    weights = np.empty((B, T), dtype=np.float64)
    consumed, probes = C.c_int64(0), C.c_int64(0)
    if philox is not None:
        mode, seed, call = 1, int(philox[0]), int(philox[1])
        u, nu = None, 0
    else:
        mode, seed, call = 0, 0, 0
        u = np.ascontiguousarray(uniforms, dtype=np.float64)
        nu = u.shape[0]
    if uoff is not None:
        uoff = np.ascontiguousarray(uoff, dtype=np.int64)
    rc = lib().orc_walk_sample(_p(g.rowptr), _p(g.col), _p(g.cdf), C.c_int64(g.V), _p(nodes), C.c_int64(B),
                               C.c_int(W), C.c_int(L), C.c_int(T), C.c_int(mode), _p(u), C.c_int64(nu), _p(uoff),
                               C.c_uint64(seed), C.c_uint32(call), _p(ids), _p(counts), _p(nvalid), _p(weights),
                               C.byref(consumed), C.byref(probes), C.c_int(threads))
    _check(rc, "orc_walk_sample")
    return ids, counts, nvalid, weights, int(consumed.value), int(probes.value)


def single_walk(g: Graph, start, L, uniforms, pos=0):
    u = np.ascontiguousarray(uniforms, dtype=np.float64)
    out = np.empty(L + 1, dtype=np.int64)
    p, n = C.c_int64(pos), C.c_int(0)
    _check(lib().orc_single_walk(_p(g.rowptr), _p(g.col), _p(g.cdf), C.c_int64(int(start)), C.c_int(L), _p(u),
                                 C.byref(p), _p(out), C.byref(n)), "orc_single_walk")
    return out[:n.value].tolist(), int(p.value)


def importance_pool(x, ids, counts, nvalid, threads=1):
    x = np.ascontiguousarray(x, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    nvalid = np.ascontiguousarray(nvalid, dtype=np.int32)
    B, T = ids.shape
    out = np.empty((B, x.shape[1]), dtype=np.float32)
    _check(lib().orc_importance_pool(_p(x), C.c_int64(x.shape[0]), C.c_int(x.shape[1]), _p(ids), _p(counts),
                                     _p(nvalid), C.c_int64(B), C.c_int(T), _p(out), C.c_int(threads)), "orc_importance_pool")
    return out


def linear(x, W, b=None, x2=None, W2=None, relu=False, l2norm=False, threads=1):
    x = np.ascontiguousarray(x, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    M, K = x.shape
    N = W.shape[0]
    K2 = 0
    if x2 is not None:
        x2 = np.ascontiguousarray(x2, dtype=np.float32)
        W2 = np.ascontiguousarray(W2, dtype=np.float32)
        K2 = x2.shape[1]
    y = np.empty((M, N), dtype=np.float32)
    _check(lib().orc_linear(_p(x), _p(W), _p(b), C.c_int64(M), C.c_int(K), C.c_int(N), _p(x2), _p(W2), C.c_int(K2),
                            C.c_int(int(relu)), C.c_int(int(l2norm)), _p(y), C.c_int(threads)), "orc_linear")
    return y


def pinsage_forward(params, x, layers, threads=1):
    """Pooled branch (model/pinsage.py:217-249) on the C oracle; layers = [(ids, counts, nvalid)]*L,
    or None for the MLP branch (:205-214)."""
    p = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in params.items()}
    L = 0
    while f"convs.{L}.lin_self.weight" in p:
        L += 1
    h = linear(x, p["input_proj.weight"], p["input_proj.bias"], relu=True, threads=threads)
    for i in range(L):
        Ws, bs = p[f"convs.{i}.lin_self.weight"], p[f"convs.{i}.lin_self.bias"]
        if layers is None:
            h = linear(h, Ws, bs, relu=True, threads=threads)
            continue
        ids, counts, nvalid = layers[i]
        hn = importance_pool(h, ids, counts, nvalid, threads=threads)
        hs = linear(h, Ws, bs, threads=threads)
        Wu, bu = p[f"convs.{i}.lin_update.weight"], p[f"convs.{i}.lin_update.bias"]
        H = hs.shape[1]
        h = linear(hs, np.ascontiguousarray(Wu[:, :H]), bu, x2=hn, W2=np.ascontiguousarray(Wu[:, H:]),
                   relu=True, l2norm=True, threads=threads)
    return linear(h, p["output_proj.weight"], p["output_proj.bias"], l2norm=True, threads=threads)


def lsh_encode(x, A, threads=1):
    x = np.ascontiguousarray(x, dtype=np.float32)
    A = np.ascontiguousarray(A, dtype=np.float32)
    nbits = A.shape[0]
    codes = np.empty((x.shape[0], (nbits + 7) // 8), dtype=np.uint8)
    _check(lib().orc_lsh_encode(_p(x), C.c_int64(x.shape[0]), C.c_int(x.shape[1]), _p(A), C.c_int(nbits), _p(codes),
                                C.c_int(threads)), "orc_lsh_encode")
    return codes


def hamming_topk(qcodes, codes, k, id_offset=0, threads=1):
    q = np.ascontiguousarray(qcodes, dtype=np.uint8)
    c = np.ascontiguousarray(codes, dtype=np.uint8)
    dist = np.empty((q.shape[0], k), dtype=np.float32)
    ids = np.empty((q.shape[0], k), dtype=np.int64)
    _check(lib().orc_hamming_topk(_p(q), C.c_int64(q.shape[0]), _p(c), C.c_int64(c.shape[0]), C.c_int(c.shape[1]),
                                  C.c_int(k), C.c_int64(id_offset), _p(dist), _p(ids), C.c_int(threads)), "orc_hamming_topk")
    return dist, ids


def dot_topk(E, qidx, k, exclude_self=True, threads=1):
    E = np.ascontiguousarray(E, dtype=np.float32)
    qidx = np.ascontiguousarray(qidx, dtype=np.int64)
    vals = np.empty((qidx.shape[0], k), dtype=np.float32)
    ids = np.empty((qidx.shape[0], k), dtype=np.int64)
    _check(lib().orc_dot_topk(_p(E), C.c_int64(E.shape[0]), C.c_int(E.shape[1]), _p(qidx), C.c_int64(qidx.shape[0]),
                              C.c_int(k), C.c_int(int(exclude_self)), _p(vals), _p(ids), C.c_int(threads)), "orc_dot_topk")
    return vals, ids
