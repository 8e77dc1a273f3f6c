"""Dense / retrieval ops over libpinsage_hip.so: fp32-MFMA linear layers with fused epilogues,
LSH encode (projection + ballot bit-pack), Hamming top-k, exact dot top-k."""
from __future__ import annotations

import os

import torch

from . import native as nv


def _rowmajor(w):
    """(tensor to keep alive, leading dimension) for a 2-D fp32 matrix whose rows are contiguous
    (column slices of a row-major weight are fine: no copy, ld = stride(0))."""
    if w.dtype != torch.float32:
        raise TypeError("fp32 expected")
    if w.dim() != 2:
        raise ValueError("2-D matrix expected")
    if w.stride(1) != 1 or w.stride(0) < w.size(1):
        w = w.contiguous()
    return w, int(w.stride(0)) if w.size(0) > 1 else max(int(w.stride(0)), int(w.size(1)))


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise nv.NativeError("libpinsage_hip takes device (HBM) tensors; got a CPU tensor (no CPU fallback)")


def _ptr_view(t):
    if t is None:
        return nv.C.c_void_p(0)
    if not t.is_cuda:
        raise nv.NativeError("device tensor expected")
    return nv.C.c_void_p(t.data_ptr())


class StagedWeight:
    """A weight matrix stored in the order the GEMM kernel stages it (ps_permute_k: every group of eight k as
    0 2 4 6 1 3 5 7).  linear() / lsh_encode() take it in place of the plain matrix (PS_WPERM): same results bit for bit,
    32 fewer vector instructions per 64 MFMAs.  For matrices that are multiplied many times (model weights, LSH rotation)."""
    __slots__ = ("t", "shape")

    def __init__(self, t):
        self.t = t
        self.shape = t.shape

    def size(self, i):
        return self.t.size(i)


def stage_weight(W):
    """StagedWeight of a [N, K] device matrix, or W itself when the kernel's staged path does not apply (K % 32 != 0)."""
    if isinstance(W, StagedWeight) or W is None:
        return W
    _require_cuda(W)
    if W.dim() != 2 or W.dtype != torch.float32 or int(W.size(1)) % 32 != 0:
        return W
    Wk, ld = _rowmajor(W)
    out = torch.empty((int(Wk.size(0)), int(Wk.size(1))), dtype=torch.float32, device=W.device)
    with torch.cuda.device(W.device):
        nv.call("ps_permute_k", _ptr_view(Wk), nv.i64(int(Wk.size(0))), nv.i32(int(Wk.size(1))), nv.i32(ld), nv.ptr(out), nv.stream())
    return StagedWeight(out)


def linear(x, W, b=None, x2=None, W2=None, relu=False, l2norm=False):
    """y = epi(x @ W.T (+ x2 @ W2.T) + b): nn.Linear / F.relu / torch.cat / F.normalize of
    PinSage.forward (reference model/pinsage.py:202,235-240,248-249) in one kernel.  W / W2: matrices or StagedWeights (both
    or neither)."""
    staged = isinstance(W, StagedWeight)
    if x2 is not None and isinstance(W2, StagedWeight) != staged:
        raise ValueError("W and W2 must both be staged or both plain")
    if staged:
        W, W2 = W.t, (W2.t if W2 is not None else None)
    _require_cuda(x, W, b, x2, W2)
    x = x.contiguous()
    if x.dtype != torch.float32:
        raise TypeError("fp32 expected")
    M, K = int(x.size(0)), int(x.size(1))
    Wk, ldw = _rowmajor(W)
    N = int(Wk.size(0))
    if int(Wk.size(1)) != K:
        raise ValueError(f"shape mismatch: x [{M},{K}] vs W {tuple(Wk.shape)}")
    K2, ldw2, W2k = 0, 0, None
    if x2 is not None:
        x2 = x2.contiguous()
        W2k, ldw2 = _rowmajor(W2)
        K2 = int(x2.size(1))
        if int(W2k.size(1)) != K2 or int(W2k.size(0)) != N or int(x2.size(0)) != M:
            raise ValueError("shape mismatch in the second operand pair")
    if b is not None:
        b = b.contiguous()
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    flags = (nv.PS_RELU if relu else 0) | (nv.PS_L2NORM if l2norm else 0) | (nv.PS_WPERM if staged else 0)
    with torch.cuda.device(x.device):
        nv.call("ps_linear", nv.ptr(x), nv.i64(M), nv.i32(K), _ptr_view(Wk), nv.i32(ldw), nv.ptr(b), nv.i32(N),
                                    nv.ptr(x2), nv.i32(K2), _ptr_view(W2k), nv.i32(ldw2), nv.i32(flags), nv.ptr(y),
                                    nv.stream())
    return y


def lsh_encode(x, A):
    """codes uint8[n, nbits/8]: bit j = (x . A[j] >= 0), LSB first (faiss IndexLSH.sa_encode).  A: matrix or StagedWeight."""
    staged = isinstance(A, StagedWeight)
    if staged:
        A = A.t
    _require_cuda(x, A)
    x = x.contiguous()
    A = A.contiguous()
    n, d = int(x.size(0)), int(x.size(1))
    nbits = int(A.size(0))
    if int(A.size(1)) != d:
        raise ValueError("projection matrix must be [nbits, dim]")
    codes = torch.empty((n, nbits // 8), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        nv.call("ps_lsh_encode", nv.ptr(x), nv.i64(n), nv.i32(d), nv.ptr(A), nv.i32(nbits), nv.ptr(codes),
                                        nv.i32(nv.PS_WPERM if staged else 0), nv.stream())
    return codes


HAMMING_MAX_K = 64        # ps_hamming_topk (popcount scan); the MFMA scan serves k <= 32; beyond 64: _hamming_topk_large_k


def lsh_expand(codes):
    """Sign planes of a code table (uint8 [n, cs] -> +1/-1 bytes in MFMA fragment order, see csrc/hamming_mfma.hip);
    None when the code size is not a multiple of 4 bytes."""
    _require_cuda(codes)
    codes = codes.contiguous()
    n, cs = int(codes.size(0)), int(codes.size(1))
    nb = int(nv.lib().ps_lsh_planes_bytes(nv.i64(n), nv.i32(cs)))
    if nb == 0:
        return None
    planes = torch.empty(nb, dtype=torch.uint8, device=codes.device)
    with torch.cuda.device(codes.device):
        nv.call("ps_lsh_expand", nv.ptr(codes), nv.i64(n), nv.i32(cs), nv.ptr(planes), nv.stream())
    return planes


def hamming_mfma_supported(nq, N, cs, k):
    return int(nv.lib().ps_hamming_topk_mfma_workspace_bytes(nv.i64(nq), nv.i64(N), nv.i32(cs), nv.i32(k))) > 0


def hamming_topk(qcodes, codes, k, id_offset=0, planes=None, use_mfma=True, out=None):
    """-> (dist int32[nq,k], ids int64[nq,k]): k smallest by (distance, id), ascending.
    out=(dist, ids): write into these contiguous device tensors (a rank's candidate record, shard.py) instead of new ones.
    `planes` = lsh_expand(codes), kept by the index: the scan then runs as an exact int8 MFMA contraction when the
    shape is served (ps_hamming_topk_mfma); otherwise, and for `use_mfma=False`, the popcount kernel runs.  Both
    return the same bits."""
    _require_cuda(qcodes, codes)
    qcodes = qcodes.contiguous()
    codes = codes.contiguous()
    nq, cs = int(qcodes.size(0)), int(qcodes.size(1))
    N = int(codes.size(0))
    if out is not None:
        dist, ids = out
        if (tuple(dist.shape) != (nq, k) or tuple(ids.shape) != (nq, k) or dist.dtype != torch.int32
                or ids.dtype != torch.int64 or not dist.is_contiguous() or not ids.is_contiguous()):
            raise ValueError("out must be contiguous (int32 [nq, k], int64 [nq, k])")
    if k > HAMMING_MAX_K or cs > 128 or (cs & (cs - 1)) != 0 or cs % 4 != 0:
        # beyond the scans' shapes (k > 64; codes longer than 1024 bits or not 32 * 2^j bits): exact L2 over the +-1 images
        d, i = _hamming_topk_large_k(qcodes, codes, k, id_offset)
        if out is None:
            return d, i
        dist.copy_(d)
        ids.copy_(i)
        return dist, ids
    if out is None:
        dist = torch.empty((nq, k), dtype=torch.int32, device=qcodes.device)
        ids = torch.empty((nq, k), dtype=torch.int64, device=qcodes.device)
    L = nv.lib()
    if use_mfma and planes is not None:
        wsb = int(L.ps_hamming_topk_mfma_workspace_bytes(nv.i64(nq), nv.i64(N), nv.i32(cs), nv.i32(k)))
        if wsb > 0:
            ws = torch.empty(wsb, dtype=torch.uint8, device=qcodes.device)
            with torch.cuda.device(qcodes.device):
                # the queries go in as packed codes: the scan's workgroups build their own sign planes
                nv.call("ps_hamming_topk_mfma_codes", nv.ptr(qcodes), nv.i64(nq), nv.ptr(planes), nv.i64(N), nv.i32(cs),
                        nv.i32(k), nv.i64(id_offset), nv.ptr(dist), nv.ptr(ids), nv.ptr(ws), nv.C.c_size_t(wsb),
                        nv.stream())
            return dist, ids
    wsb = int(L.ps_hamming_topk_workspace_bytes(nv.i64(nq), nv.i64(N), nv.i32(cs), nv.i32(k)))
    ws = torch.empty(wsb, dtype=torch.uint8, device=qcodes.device)
    with torch.cuda.device(qcodes.device):
        nv.call("ps_hamming_topk", nv.ptr(qcodes), nv.i64(nq), nv.ptr(codes) if N else nv.C.c_void_p(0), nv.i64(N),
                                   nv.i32(cs), nv.i32(k), nv.i64(id_offset), nv.ptr(dist), nv.ptr(ids), nv.ptr(ws),
                                   nv.C.c_size_t(wsb), nv.stream())
    return dist, ids


LARGE_K_SIGN_BYTES = 8 << 30      # budget for the +-1 float image of the code table in the k > 64 path


def _signs(codes):
    """uint8 [n, cs] -> float32 [n, 8 cs] of +-1 (any fixed bit order: both operands use the same)"""
    shifts = torch.arange(8, device=codes.device, dtype=torch.uint8)
    bits = (codes.unsqueeze(-1) >> shifts) & 1
    return bits.reshape(codes.size(0), -1).to(torch.float32) * 2.0 - 1.0


def _hamming_topk_large_k(qcodes, codes, k, id_offset):
    """k > 64 (faiss.IndexLSH.search accepts any k): between +-1 vectors the squared L2 distance is 4 x the Hamming distance
    -- small integers, exact in fp32 -- so the exact L2 search (ps_l2_topk: fp32-MFMA GEMM + multi-sweep row top-k, any k,
    ties by id) returns the same (distance, id) order as the scans.  Costs a float image of the table (2 KiB per 512-bit
    code): meant for the occasional large request, not for the hot path."""
    N, nbits = int(codes.size(0)), int(codes.size(1)) * 8
    if N * nbits * 4 > LARGE_K_SIGN_BYTES:
        raise ValueError(f"LSH search with k = {k} > {HAMMING_MAX_K} expands the code table to floats "
                         f"({N * nbits * 4 / 2 ** 30:.1f} GiB here, limit {LARGE_K_SIGN_BYTES >> 30} GiB): split the table or the request")
    d2, ids = l2_topk(_signs(codes), _signs(qcodes), k)
    missing = ids < 0
    dist = torch.where(missing, torch.full_like(d2, 0.0), d2 * 0.25).round().to(torch.int32)
    dist[missing] = 0x7fffffff
    ids = torch.where(missing, ids, ids + int(id_offset))
    return dist, ids


def topk_merge(dist_in, ids_in):
    """[P, nq, k] candidate lists -> global [nq, k] by (distance, id)."""
    dist_in = dist_in.contiguous()
    ids_in = ids_in.contiguous()
    P, nq, k = [int(v) for v in dist_in.shape]
    dist = torch.empty((nq, k), dtype=torch.int32, device=dist_in.device)
    ids = torch.empty((nq, k), dtype=torch.int64, device=dist_in.device)
    with torch.cuda.device(dist_in.device):
        nv.call("ps_topk_merge", nv.ptr(dist_in), nv.ptr(ids_in), nv.i32(P), nv.i64(nq), nv.i32(k), nv.ptr(dist),
                                        nv.ptr(ids), nv.stream())
    return dist, ids


def topk_merge_records(records, nq, k):
    """records uint8 [P, rec]: shard p's candidate record = [nq*k int64 ids | nq*k int32 distances | pad to 16 B], exactly
    as one all-gather delivers them -> global (dist int32[nq,k], ids int64[nq,k]) by (distance, id); no repacking."""
    if records.dtype != torch.uint8 or records.dim() != 2 or not records.is_contiguous():
        raise ValueError("records must be a contiguous uint8 [P, record_bytes] tensor")
    P, rec = int(records.size(0)), int(records.size(1))
    n = int(nq) * int(k)
    if rec < 12 * n or rec % 8 != 0:
        raise ValueError("record too short or not 8-byte aligned")
    dist = torch.empty((nq, k), dtype=torch.int32, device=records.device)
    ids = torch.empty((nq, k), dtype=torch.int64, device=records.device)
    base = records.data_ptr()
    with torch.cuda.device(records.device):
        nv.call("ps_topk_merge_strided", nv.C.c_void_p(base + 8 * n), nv.i64(rec // 4), nv.C.c_void_p(base), nv.i64(rec // 8),
                nv.i32(P), nv.i64(int(nq)), nv.i32(int(k)), nv.ptr(dist), nv.ptr(ids), nv.stream())
    return dist, ids


def dot_topk(E, qidx, k, exclude_self=True):
    """Exact search: top-k of E[q] @ E.T per query row index (reference inference.py:112-118)."""
    E = E.contiguous()
    qidx = qidx.to(device=E.device, dtype=torch.int64).contiguous()
    N, D = int(E.size(0)), int(E.size(1))
    nq = int(qidx.numel())
    if k < 1:
        raise ValueError(f"k must be positive, got {k}")
    vals = torch.empty((nq, k), dtype=torch.float32, device=E.device)
    ids = torch.empty((nq, k), dtype=torch.int64, device=E.device)
    L = nv.lib()
    wsb = int(L.ps_dot_topk_workspace_bytes(nv.i64(nq), nv.i64(N), nv.i32(D), nv.i32(k)))
    ws = torch.empty(wsb, dtype=torch.uint8, device=E.device)
    with torch.cuda.device(E.device):
        nv.call("ps_dot_topk", nv.ptr(E), nv.i64(N), nv.i32(D), nv.ptr(qidx), nv.i64(nq), nv.i32(k),
                               nv.i32(int(exclude_self)), nv.ptr(vals), nv.ptr(ids), nv.ptr(ws), nv.C.c_size_t(wsb),
                               nv.stream())
    return vals, ids


def l2_topk(X, Q, k, assign=None, probe=None):
    """k nearest by squared L2 (IndexFlatL2 / IndexIVFFlat scan): -> (dist fp32[nq,k], ids int64[nq,k])."""
    X = X.contiguous()
    Q = Q.to(X.device).contiguous()
    N, D = int(X.size(0)), int(X.size(1))
    nq = int(Q.size(0))
    if k < 1:
        raise ValueError(f"k must be positive, got {k}")
    dist = torch.empty((nq, k), dtype=torch.float32, device=X.device)
    ids = torch.empty((nq, k), dtype=torch.int64, device=X.device)
    L = nv.lib()
    wsb = int(L.ps_l2_topk_workspace_bytes(nv.i64(nq), nv.i64(N), nv.i32(D), nv.i32(k)))
    ws = torch.empty(wsb, dtype=torch.uint8, device=X.device)
    words = int(probe.size(1)) if probe is not None else 0
    with torch.cuda.device(X.device):
        nv.call("ps_l2_topk", nv.ptr(X), nv.i64(N), nv.i32(D), nv.ptr(Q), nv.i64(nq), nv.i32(k), nv.ptr(assign),
                nv.ptr(probe), nv.i32(words), nv.ptr(dist), nv.ptr(ids), nv.ptr(ws), nv.C.c_size_t(wsb), nv.stream())
    return dist, ids


def ivf_topk(Xs, list_ptr, item_ids, Q, probes, k, max_list=None):
    """Inverted-file scan (ps_ivf_topk): Xs fp32 [N, D] sorted by list, list_ptr int64 [nlist + 1], item_ids int64 [N] (original
    id of every sorted row), probes int32 [nq, nprobe] -> (dist fp32 [nq, k], ids int64 [nq, k]) by (squared L2, original id).
    max_list = the longest list (computed here with one host sync when not given: an index knows it from `add`)."""
    Xs = Xs.contiguous()
    Q = Q.to(Xs.device).contiguous()
    probes = probes.to(device=Xs.device, dtype=torch.int32).contiguous()
    list_ptr = list_ptr.to(device=Xs.device, dtype=torch.int64).contiguous()
    item_ids = item_ids.to(device=Xs.device, dtype=torch.int64).contiguous()
    N, D = int(Xs.size(0)), int(Xs.size(1))
    nq, nprobe, nlist = int(Q.size(0)), int(probes.size(1)), int(list_ptr.numel()) - 1
    if k < 1:
        raise ValueError(f"k must be positive, got {k}")
    if int(probes.size(0)) != nq or int(item_ids.numel()) != N or int(Q.size(1)) != D:
        raise ValueError("shape mismatch")
    if max_list is None:
        max_list = int((list_ptr[1:] - list_ptr[:-1]).max().item()) if nlist else 0
    dist = torch.empty((nq, k), dtype=torch.float32, device=Xs.device)
    ids = torch.empty((nq, k), dtype=torch.int64, device=Xs.device)
    L = nv.lib()
    wsb = int(L.ps_ivf_topk_workspace_bytes(nv.i64(nq), nv.i64(N), nv.i32(D), nv.i32(k), nv.i32(nlist), nv.i32(nprobe), nv.i64(int(max_list))))
    ws = torch.empty(wsb, dtype=torch.uint8, device=Xs.device)
    with torch.cuda.device(Xs.device):
        nv.call("ps_ivf_topk", nv.ptr(Xs), nv.i64(N), nv.i32(D), nv.ptr(list_ptr), nv.i32(nlist), nv.i64(int(max_list)), nv.ptr(item_ids),
                nv.ptr(Q), nv.i64(nq), nv.ptr(probes), nv.i32(nprobe), nv.i32(int(k)), nv.ptr(dist), nv.ptr(ids), nv.ptr(ws),
                nv.C.c_size_t(wsb), nv.stream())
    return dist, ids


_jump_polys_dev = {}


def _jump_polys(dev):
    key = str(dev)
    if key not in _jump_polys_dev:
        from . import mtjump
        import numpy as np
        _jump_polys_dev[key] = torch.from_numpy(mtjump.jump_polynomials().view(np.int32)).to(dev).contiguous()
    return _jump_polys_dev[key]


_radix_polys_dev = {}


def _radix_polys(dev):
    key = str(dev)
    if key not in _radix_polys_dev:
        from . import mtjump
        import numpy as np
        cl2 = int(nv.lib().ps_mt19937_chunk_log2())
        _radix_polys_dev[key] = torch.from_numpy(mtjump.radix_polynomials(cl2).view(np.int32)).to(dev).contiguous()
    return _radix_polys_dev[key]


_window_polys_dev = {}


def _window_polys(dev):
    key = str(dev)
    if key not in _window_polys_dev:
        from . import mtjump
        import numpy as np
        cl2 = int(nv.lib().ps_mt19937_chunk_log2())
        shift = int(nv.lib().ps_mt19937_window_shift())
        _window_polys_dev[key] = torch.from_numpy(mtjump.window_polynomials(cl2, shift=shift).view(np.int32)).to(dev).contiguous()
    return _window_polys_dev[key]


_pending_state = []


def finish_rng_state():
    """Completes a deferred hand-back of the np.random state (mt19937_random_sample(advance='defer')): waits for the
    624-word state copy (an event, not a device synchronisation) and installs it with np.random.set_state.  Called by
    the code that asked for the deferral before it returns to the caller, so user code always sees the advanced state."""
    import numpy as np
    while _pending_state:
        name, has_gauss, cached, host_state, host_pos, ev = _pending_state.pop()
        ev.synchronize()
        np.random.set_state((name, host_state.numpy().view(np.uint32).copy(), int(host_pos.item()), has_gauss, cached))


def mt19937_random_sample(n, device, skip=0, advance=True, parallel=True, radix=True, raw=False, one_round=True, ranges=None):
    """n doubles of the process-global numpy legacy stream generated ON THE DEVICE (after skipping `skip`
    doubles); with advance=True the global np.random state is advanced exactly as
    `np.random.random_sample(skip + n)` would (reference utils/random_walk.py:79 draws these one at a time).
    advance='defer': the state comes back through an asynchronous copy that `finish_rng_state()` completes -- the host
    keeps enqueueing the kernels that consume the uniforms instead of waiting for the generator (the caller must call
    finish_rng_state() before returning to code that may touch np.random).
    raw=True (skip = 0, n >= 2^17): returns the stream as untempered MT19937 words, int32[2n + 1248], for the walk kernels'
    PS_RNG_STREAM_RAW mode (uniform i = words 2i, 2i + 1) instead of doubles.
    ranges (raw only): up to three runs (lo, hi) of uniform indices the caller will read; only those words of the buffer are
    generated (item shards: the stream positions of a rank's own start nodes), the np.random state still advances by n."""
    import numpy as np
    name, key, pos, has_gauss, cached = np.random.get_state()
    if name != "MT19937":
        raise RuntimeError("numpy global RNG is not MT19937")
    dev = torch.device(device)
    # pinned staging + asynchronous copy: a pageable `.to(dev)` waits for the stream, i.e. for the whole previous step
    if os.environ.get("PS_MT_SYNC_UPLOAD") == "1":
        st_in = torch.from_numpy(key.astype(np.uint32).view(np.int32)).to(dev)
    else:
        st_in = torch.from_numpy(key.astype(np.uint32).view(np.int32)).pin_memory().to(dev, non_blocking=True)
    st_out = torch.empty(624, dtype=torch.int32, device=dev)
    pos_out = torch.empty(1, dtype=torch.int32, device=dev)
    if raw and (skip != 0 or not parallel or n < (1 << 17)):
        raise ValueError("raw stream: skip = 0, the parallel generator and n >= 2^17 are required")
    out = (torch.empty(2 * int(n) + 1248, dtype=torch.int32, device=dev) if raw else
           torch.empty(int(n), dtype=torch.float64, device=dev))
    polys = _jump_polys(dev) if parallel else None
    rpolys = _radix_polys(dev) if (parallel and radix) else None      # radix=False: windows by doubling
    wpolys = _window_polys(dev) if (parallel and radix and one_round) else None    # one_round=False: two radix-32 rounds
    L = nv.lib()
    wsb = int(L.ps_mt19937_workspace_bytes(nv.i64(int(skip)), nv.i64(int(n)))) if parallel else 0
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if parallel else None
    if ranges is not None and not raw:
        raise ValueError("ranges: raw stream only")
    rg = np.ascontiguousarray(np.asarray(ranges, dtype=np.int64).reshape(-1, 2)) if ranges is not None and len(ranges) else None
    if rg is not None and rg.shape[0] > 3:
        rg = None                                                    # more runs than the planner takes: generate everything
    if rg is not None and os.environ.get("PS_MT_POISON") == "1":
        # debug aid (tests): words outside the requested runs stay unwritten, and the caching allocator readily hands back a
        # block that still holds a same-seed stream from an earlier call -- a kernel that read outside its runs would then see
        # correct-looking words.  Poisoned, any such read changes the sampled ids deterministically.
        out.fill_(-1)
    with torch.cuda.device(dev):
        if raw:
            nv.call("ps_mt19937_raw_stream", nv.ptr(st_in), nv.i32(int(pos)), nv.i64(int(n)), nv.ptr(out), nv.ptr(st_out),
                    nv.ptr(pos_out), nv.ptr(polys), nv.i32(int(polys.size(0))), nv.ptr(rpolys),
                    nv.i32(int(rpolys.size(0)) if rpolys is not None else 0), nv.ptr(wpolys),
                    nv.i32(int(wpolys.size(0)) if wpolys is not None else 0),
                    rg.ctypes.data_as(nv.C.c_void_p) if rg is not None else nv.C.c_void_p(0), nv.i32(int(rg.shape[0]) if rg is not None else 0),
                    nv.ptr(ws), nv.C.c_size_t(wsb), nv.stream())
        else:
          nv.call("ps_mt19937_random_sample", nv.ptr(st_in), nv.i32(int(pos)), nv.i64(int(skip)), nv.i64(int(n)),
                nv.ptr(out), nv.ptr(st_out), nv.ptr(pos_out), nv.ptr(polys),
                nv.i32(int(polys.size(0)) if polys is not None else 0), nv.ptr(rpolys),
                nv.i32(int(rpolys.size(0)) if rpolys is not None else 0), nv.ptr(wpolys),
                nv.i32(int(wpolys.size(0)) if wpolys is not None else 0), nv.ptr(ws), nv.C.c_size_t(wsb), nv.stream())
    if advance == "defer":
        finish_rng_state()                                           # at most one hand-back in flight
        host_state = torch.empty(624, dtype=torch.int32).pin_memory()
        host_pos = torch.empty(1, dtype=torch.int32).pin_memory()
        with torch.cuda.device(dev):
            host_state.copy_(st_out, non_blocking=True)
            host_pos.copy_(pos_out, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        _pending_state.append((name, has_gauss, cached, host_state, host_pos, ev))
    elif advance:
        new_key = st_out.cpu().numpy().view(np.uint32)
        np.random.set_state((name, new_key, int(pos_out.item()), has_gauss, cached))
    return out
