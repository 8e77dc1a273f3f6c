"""Tensor-native sampling ops over a DeviceGraph (the kernels behind RandomWalkSampler).

`NeighborBatch` is the device-resident result ([B,T] ids / visit counts, [B] nvalid);
`LazyNeighborList` is the python list-of-lists view the reference API returns
(utils/random_walk.py:119-142), materialised only when python code actually looks at it."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import native as nv


class NeighborBatch:
    """ids int32[B,T] (-1 pad), counts int32[B,T] (0 pad), nvalid int32[B] on the device."""

    def __init__(self, ids, counts, nvalid):
        self.ids, self.counts, self.nvalid = ids, counts, nvalid
        self._host = None

    @property
    def B(self):
        return int(self.ids.size(0))

    @property
    def T(self):
        return int(self.ids.size(1))

    def host(self):
        """(ids int64, counts int32, nvalid int32, weights fp64) as numpy; weights = count / sum(top
        counts) in fp64 exactly like python's int/int (utils/random_walk.py:113-115)."""
        if self._host is None:
            ids = self.ids.cpu().numpy().astype(np.int64)
            counts = self.counts.cpu().numpy()
            nvalid = self.nvalid.cpu().numpy()
            tot = counts.sum(axis=1, dtype=np.int64)
            w = counts.astype(np.float64) / np.maximum(tot, 1)[:, None].astype(np.float64)
            self._host = (ids, counts, nvalid, w)
        return self._host

    def to_lists(self):
        ids, counts, nvalid, w = self.host()
        nb = [list(ids[i, :nvalid[i]]) for i in range(ids.shape[0])]          # np.int64 scalars
        wt = [w[i, :nvalid[i]].tolist() for i in range(ids.shape[0])]          # python floats
        return nb, wt


class LazyNeighborList(list):
    """A real `list` (isinstance checks in the reference pass) whose items are produced from the
    device batch on first python-level access.  Our own kernels read `.batch` and never touch it.
    Until then the C-level list holds `None` placeholders of the right length, so consumers that bypass the python
    protocol (PySequence_Fast: `torch.tensor(lst)`, `np.asarray(lst)` on some paths) fail loudly on a None instead of
    silently seeing an empty list; `.materialize()` (or any python-level access) fills it."""

    EAGER_BELOW = 4096

    def __init__(self, batch: NeighborBatch, kind: str):
        super().__init__()
        self.batch = batch
        self.kind = kind
        self._done = False
        if batch.B <= self.EAGER_BELOW:       # small batches: a plain, fully populated list (C-level consumers too)
            self._fill()
        else:
            super().extend([None] * batch.B)

    def _fill(self):
        if not self._done:
            self._done = True
            nb, wt = self.batch.to_lists()
            list.__setitem__(self, slice(None), nb if self.kind == "ids" else wt)

    def materialize(self):
        """fill the list now (for code that hands it to C-level consumers)"""
        self._fill()
        return self

    def __len__(self):
        return self.batch.B

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __getitem__(self, i):
        self._fill()
        return super().__getitem__(i)

    def __eq__(self, other):
        self._fill()
        return super().__eq__(other)

    def __repr__(self):
        self._fill()
        return super().__repr__()

    def __bool__(self):
        return self.batch.B > 0

    def __reduce__(self):
        self._fill()
        return (list, (list(super().__iter__()),))


def _nodes_tensor(nodes, device, num_nodes):
    """int64 device tensor of start nodes.  Host inputs (lists / numpy / CPU tensors) are range-checked
    here like the reference's `adj_list[node]` (utils/random_walk.py:66 -> IndexError); tensors already
    on the device are not synchronised on: the kernels treat out-of-range ids as isolated nodes."""
    if isinstance(nodes, torch.Tensor) and nodes.is_cuda:
        return nodes.to(device=device, dtype=torch.int64).reshape(-1).contiguous()
    a = nodes.numpy() if isinstance(nodes, torch.Tensor) else np.asarray(nodes)
    a = a.astype(np.int64, copy=False).reshape(-1)
    if a.size and (a.min() < -num_nodes or a.max() >= num_nodes):
        raise IndexError("list index out of range")
    if a.size and a.min() < 0:
        a = np.where(a < 0, a + num_nodes, a)            # python negative indexing into adj_list
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def draw_numpy_uniforms(n, device, defer_state=False, raw=False, ranges=None):
    """n doubles of the process-global legacy numpy stream, exactly what n sequential
    `np.random.choice(..., p=...)` calls consume (utils/random_walk.py:79), staged to HBM.
    defer_state: see dense.mt19937_random_sample(advance='defer').  raw=True: the caller can consume the stream as raw
    MT19937 words (PS_RNG_STREAM_RAW); returns (tensor, is_raw) -- short requests are drawn on the host as doubles.
    ranges (raw): the runs of uniform indices the caller reads (dense.mt19937_random_sample); the rest stays unwritten."""
    if n >= (1 << 17):
        from . import dense                    # same stream, generated on the device (jump-ahead chunks)
        t = dense.mt19937_random_sample(int(n), device, advance="defer" if defer_state else True, raw=raw, ranges=ranges if raw else None)
        return (t, True) if raw else t
    if raw:
        return draw_numpy_uniforms(n, device, defer_state), False
    u = np.random.random_sample(int(n))
    t = torch.from_numpy(u)
    if n:
        t = t.pin_memory()
    return t.to(device, non_blocking=True)


def sink_walk_offsets(graph, starts, W, L, uniforms):
    """Stream position of every walk, int64[B * W], on a graph with REACHABLE SINKS, plus the number of uniforms the batch
    consumes.  The reference breaks a walk at a node without out-edges before drawing (utils/random_walk.py:65-69), so a walk
    consumes between 0 and L uniforms and walk j starts at the running count of uniforms consumed by every earlier walk:
    off_j = sum_{i<j} steps_i, steps_j = f_j(off_j).  That recurrence has exactly one solution, and it is found in parallel as a
    fixpoint: walk all B * W walks from guessed positions (ps_walk_paths), count the steps they took, prefix-sum, repeat until
    the counts reproduce themselves.  Every pass makes at least one more leading walk final, so it terminates; where the
    number of steps is decided by the graph rather than by the draw (e.g. every walk from a user lands on an item that is a
    sink) two or three passes are enough.  Exact, slower than the sink-free path, and only taken by graphs that need it."""
    dev = graph.device
    B = int(starts.numel())
    walks = starts.repeat_interleave(W).contiguous()
    ok = (starts >= 0) & (starts < graph.V)
    deg = torch.zeros(B, dtype=torch.int64, device=dev)
    deg[ok] = graph.rowptr[starts[ok] + 1] - graph.rowptr[starts[ok]]
    steps = ((deg > 0).to(torch.int64) * L).repeat_interleave(W)
    paths = torch.empty((B * W, L), dtype=torch.int32, device=dev)
    off = torch.zeros(B * W, dtype=torch.int64, device=dev)
    for _ in range(B * W + 2):
        off = (torch.cumsum(steps, 0) - steps).contiguous()
        if B * W == 0:
            break
        nv.call("ps_walk_paths", nv.ptr(graph.rowptr), nv.ptr(graph.col), nv.ptr(graph.cdf), nv.i64(graph.V),
                nv.ptr(walks), nv.i64(B * W), nv.i32(L), nv.i32(nv.PS_RNG_STREAM), nv.ptr(uniforms), nv.ptr(off),
                nv.u64(0), nv.u32(0), nv.i32(0), nv.ptr(graph.nodeinfo), nv.ptr(graph.guide), nv.ptr(paths), nv.stream())
        took = (paths >= 0).sum(dim=1)
        if torch.equal(took, steps):
            return off, int(steps.sum().item())
        steps = took
    return off, int(steps.sum().item())


def _sink_stream(graph, starts, W, L, uniforms=None):
    """(uniforms on the device, per-walk offsets) for a batch on a graph with reachable sinks; draws from np.random what the
    reference would (at most one uniform per step; the global state ends exactly where the reference leaves it)."""
    dev = graph.device
    B = int(starts.numel())
    given = uniforms is not None
    if not given:
        state = np.random.get_state()
        host = np.random.random_sample(B * W * L + 1)          # the most the batch can consume (+1: never an empty buffer)
        uniforms = torch.from_numpy(host).to(dev)
    off, total = sink_walk_offsets(graph, starts, W, L, uniforms)
    if not given:
        np.random.set_state(state)
        np.random.random_sample(total)                           # advance the global stream by what was really consumed
    return uniforms, off


def walk_sample(graph, nodes, T, W=100, L=2, rng="numpy", seed=0, call=0, uniforms=None, use_guide=True, use_packed=True, use_buckets=True,
                stream_nodes=None, use_dest=True):
    """batch_sample_neighbors on the device.  rng='numpy': the global numpy stream (bit-exact with
    the reference; graphs with reachable sinks take the per-walk stream positions of sink_walk_offsets);
    rng='philox': counter-based.
    `uniforms` (device fp64) overrides the numpy draw (tests).  `stream_nodes` (rng='numpy' only): the complete
    start-node sequence of the logical batch this call is a contiguous slice of (item shards): the stream offsets
    are computed over all of them and the whole batch's uniforms are drawn, so the ids/counts equal the matching
    rows of the unsharded call and the global np.random state ends where the unsharded call leaves it."""
    dev = graph.device
    starts = _nodes_tensor(nodes, dev, graph.V)
    B = int(starts.numel())
    ids = torch.empty((B, T), dtype=torch.int32, device=dev)
    counts = torch.empty((B, T), dtype=torch.int32, device=dev)
    nvalid = torch.empty(B, dtype=torch.int32, device=dev)
    L_ = nv.lib()
    with torch.cuda.device(dev):
        if rng == "numpy" and graph.has_reachable_sink:
            # data-dependent RNG consumption (utils/random_walk.py:65-69): per-walk stream positions by fixpoint
            if stream_nodes is not None:
                all_nodes, lo = stream_nodes
                all_t = _nodes_tensor(all_nodes, dev, graph.V)
                uniforms, off_all = _sink_stream(graph, all_t, W, L, uniforms)
                uoff = off_all[lo * W:(lo + B) * W].contiguous()
            else:
                uniforms, uoff = _sink_stream(graph, starts, W, L, uniforms)
            mode = nv.PS_RNG_STREAM_WALKS
        elif rng == "numpy":
            total = torch.empty(1, dtype=torch.int64, device=dev)
            if stream_nodes is not None:
                all_nodes, lo = stream_nodes
                all_nodes = _nodes_tensor(all_nodes, dev, graph.V)
                uoff_all = torch.empty(all_nodes.numel(), dtype=torch.int64, device=dev)
                nv.call("ps_uniform_offsets", nv.ptr(graph.rowptr), nv.i64(graph.V), nv.ptr(all_nodes),
                        nv.i64(all_nodes.numel()), nv.i32(W), nv.i32(L), nv.ptr(uoff_all), nv.ptr(total), nv.stream())
                uoff = uoff_all[lo:lo + B].contiguous()
            else:
                uoff = torch.empty(B, dtype=torch.int64, device=dev)
                nv.call("ps_uniform_offsets", nv.ptr(graph.rowptr), nv.i64(graph.V), nv.ptr(starts), nv.i64(B),
                        nv.i32(W), nv.i32(L), nv.ptr(uoff), nv.ptr(total), nv.stream())
            if uniforms is None:
                n = int(total.item())
                uniforms = draw_numpy_uniforms(n, dev)
            mode = nv.PS_RNG_STREAM
        elif rng == "philox":
            uoff, uniforms, mode = None, None, nv.PS_RNG_PHILOX
        else:
            raise ValueError("rng must be 'numpy' or 'philox'")
        nv.call("ps_walk_sample", nv.ptr(graph.rowptr), nv.ptr(graph.col), nv.ptr(graph.cdf), nv.i64(graph.V),
                                   nv.ptr(starts), nv.i64(B), nv.i32(W), nv.i32(L), nv.i32(T),
                                   nv.i32(mode | (graph.walk_flags if (use_guide and use_buckets) else 0)),
                                   nv.ptr(uniforms), nv.ptr(uoff), nv.u64(seed & (2 ** 64 - 1)), nv.u32(call),
                                   nv.ptr(graph.nodeinfo) if use_guide else nv.ptr(None),
                                   nv.ptr(graph.guide) if use_guide else nv.ptr(None),
                                   nv.ptr(graph.packed) if (use_guide and use_packed) else nv.ptr(None),
                                   nv.ptr(getattr(graph, "buckets", None)) if (use_guide and use_buckets) else nv.ptr(None),
                                   nv.ptr(getattr(graph, "dest_info", None)) if (use_guide and use_packed and use_dest) else nv.ptr(None),
                                   nv.ptr(ids), nv.ptr(counts), nv.ptr(nvalid), nv.stream())
    return NeighborBatch(ids, counts, nvalid)


def walk_sample_layers(graph, nodes, T, layers, W=100, L=2, rng="numpy", seed=0, call=0, uniforms=None,
                       stream_nodes=None, defer_state=False):
    """`layers` consecutive batch_sample_neighbors calls over the same start nodes (PinSage.get_embeddings,
    model/pinsage.py:271-275) as ONE kernel launch (ps_walk_sample_layers) -> list of NeighborBatch.  Bit-identical
    to `layers` walk_sample calls: Philox uses call, call + 1, ...; rng='numpy' draws all layers' uniforms from the
    global stream in one go (layer 0's, then layer 1's, ... as the reference consumes them) and hands the state back
    once.  `nodes` may be a python range (no host sync in the steady state).  defer_state=True leaves that hand-back to
    dense.finish_rng_state(), which the caller must invoke before it returns to user code."""
    dev = graph.device
    if rng == "numpy" and graph.has_reachable_sink:
        # a walk that reaches a sink consumes fewer uniforms: the layers are drawn one after the other, each from where the
        # previous one left the stream (what consecutive batch_sample_neighbors calls do in the reference)
        nd = torch.arange(nodes.start, nodes.stop, dtype=torch.int64, device=dev) if isinstance(nodes, range) else nodes
        sn = None
        if stream_nodes is not None:
            a0 = stream_nodes[0]
            sn = (torch.arange(a0.start, a0.stop, dtype=torch.int64, device=dev) if isinstance(a0, range) else a0, stream_nodes[1])
        return [walk_sample(graph, nd, T, W, L, rng=rng, seed=seed, call=call + r, uniforms=None, stream_nodes=sn)
                for r in range(layers)]
    as_range = nodes if isinstance(nodes, range) else None
    if as_range is not None:
        if as_range.step != 1 or as_range.start < 0 or as_range.stop > graph.V:
            raise IndexError("list index out of range")
        starts = torch.arange(as_range.start, as_range.stop, dtype=torch.int64, device=dev)
    else:
        starts = _nodes_tensor(nodes, dev, graph.V)
    B = int(starts.numel())
    ids = torch.empty((layers, B, T), dtype=torch.int32, device=dev)
    counts = torch.empty((layers, B, T), dtype=torch.int32, device=dev)
    nvalid = torch.empty((layers, B), dtype=torch.int32, device=dev)
    stride = 0
    with torch.cuda.device(dev):
        if rng == "numpy":
            # stream offsets: a property of (graph, start nodes, W * L), not of the RNG state.  For a python range of
            # start nodes (what get_embeddings and item shards sample) they are computed once per graph and kept in
            # HBM, like the CSR itself; anything else runs ps_uniform_offsets per call.
            src = stream_nodes[0] if stream_nodes is not None else (as_range if as_range is not None else starts)
            lo = stream_nodes[1] if stream_nodes is not None else 0
            key = ("uoff", src.start, src.stop, W * L) if isinstance(src, range) and src.step == 1 else None
            cache = graph.__dict__.setdefault("_stream_offsets", {})
            if key is not None and key in cache:
                uoff_all, stride = cache[key]
            else:
                all_t = (torch.arange(src.start, src.stop, dtype=torch.int64, device=dev)
                         if isinstance(src, range) else _nodes_tensor(src, dev, graph.V))
                total = torch.empty(1, dtype=torch.int64, device=dev)
                uoff_all = torch.empty(all_t.numel(), dtype=torch.int64, device=dev)
                nv.call("ps_uniform_offsets", nv.ptr(graph.rowptr), nv.i64(graph.V), nv.ptr(all_t),
                        nv.i64(all_t.numel()), nv.i32(W), nv.i32(L), nv.ptr(uoff_all), nv.ptr(total), nv.stream())
                stride = int(total.item())
                if key is not None:
                    cache[key] = (uoff_all, stride)
            uoff = uoff_all if stream_nodes is None else uoff_all[lo:lo + B].contiguous()
            mode = nv.PS_RNG_STREAM
            if uniforms is None:
                # an item shard reads only the stream positions of its own start nodes: one run per layer (its bounds are, like
                # the offsets, a property of the graph: fetched once and kept)
                runs = None
                if stream_nodes is not None and B > 0 and B < uoff_all.numel():
                    rkey = None if key is None else key + ("run", lo, B)
                    if rkey is not None and rkey in cache:
                        u_lo, u_hi = cache[rkey]
                    else:
                        u_lo = int(uoff_all[lo].item())
                        u_hi = int(uoff_all[lo + B].item()) if lo + B < uoff_all.numel() else stride
                        if rkey is not None:
                            cache[rkey] = (u_lo, u_hi)
                    if layers <= 3 and os.environ.get("PS_MT_RANGES", "1") != "0":       # PS_MT_RANGES=0: every rank generates the whole stream
                        runs = [(r * stride + u_lo, r * stride + u_hi) for r in range(layers)]
                uniforms, is_raw = draw_numpy_uniforms(layers * stride, dev, defer_state=defer_state, raw=True, ranges=runs)
                mode = nv.PS_RNG_STREAM_RAW if is_raw else nv.PS_RNG_STREAM
        elif rng == "philox":
            uoff, uniforms, mode = None, None, nv.PS_RNG_PHILOX
        else:
            raise ValueError("rng must be 'numpy' or 'philox'")
        # one launch samples up to eight layers of a start node in one wave; deeper models take further launches over the
        # next layers' calls / stream positions (same results: the layers are independent draws)
        for r0 in range(0, layers, 8):
            n = min(8, layers - r0)
            u = uniforms
            if uniforms is not None and r0:
                u = uniforms[2 * r0 * stride:] if mode == nv.PS_RNG_STREAM_RAW else uniforms[r0 * stride:]
            nv.call("ps_walk_sample_layers", nv.ptr(graph.rowptr), nv.ptr(graph.col), nv.ptr(graph.cdf), nv.i64(graph.V),
                    nv.ptr(starts), nv.i64(B), nv.i32(W), nv.i32(L), nv.i32(T), nv.i32(mode | graph.walk_flags), nv.ptr(u), nv.ptr(uoff),
                    nv.i64(stride), nv.u64(seed & (2 ** 64 - 1)), nv.u32(call + r0), nv.ptr(graph.nodeinfo), nv.ptr(graph.guide),
                    nv.ptr(graph.packed), nv.ptr(getattr(graph, "buckets", None)), nv.ptr(getattr(graph, "dest_info", None)),
                    nv.i32(n), nv.ptr(ids[r0:r0 + n]),
                    nv.ptr(counts[r0:r0 + n]), nv.ptr(nvalid[r0:r0 + n]), nv.stream())
    return [NeighborBatch(ids[r], counts[r], nvalid[r]) for r in range(layers)]


def walk_paths(graph, starts, L, rng="numpy", seed=0, call=0, walk_mod=0):
    """One walk per start node: int32[B,L] visited nodes (-1 after a sink)."""
    dev = graph.device
    st = _nodes_tensor(starts, dev, graph.V)
    B = int(st.numel())
    paths = torch.empty((B, L), dtype=torch.int32, device=dev)
    L_ = nv.lib()
    with torch.cuda.device(dev):
        if rng == "numpy" and graph.has_reachable_sink:
            uniforms, uoff = _sink_stream(graph, st, 1, L)       # one walk per start node: W = 1
            mode = nv.PS_RNG_STREAM
        elif rng == "numpy":
            uoff = torch.empty(B, dtype=torch.int64, device=dev)
            total = torch.empty(1, dtype=torch.int64, device=dev)
            nv.call("ps_uniform_offsets", nv.ptr(graph.rowptr), nv.i64(graph.V), nv.ptr(st), nv.i64(B), nv.i32(1),
                                           nv.i32(L), nv.ptr(uoff), nv.ptr(total), nv.stream())
            uniforms = draw_numpy_uniforms(int(total.item()), dev)
            mode = nv.PS_RNG_STREAM
        else:
            uoff, uniforms, mode = None, None, nv.PS_RNG_PHILOX
        nv.call("ps_walk_paths", nv.ptr(graph.rowptr), nv.ptr(graph.col), nv.ptr(graph.cdf), nv.i64(graph.V),
                                  nv.ptr(st), nv.i64(B), nv.i32(L), nv.i32(mode), nv.ptr(uniforms), nv.ptr(uoff),
                                  nv.u64(seed & (2 ** 64 - 1)), nv.u32(call), nv.i32(walk_mod), nv.ptr(graph.nodeinfo),
                                  nv.ptr(graph.guide), nv.ptr(paths), nv.stream())
    return paths


def importance_pool(x, batch: NeighborBatch = None, ids=None, counts=None, wts=None, nvalid=None, max_idx=None,
                    renorm=True):
    """out[B,H] = sum_j w_j x[ids_j]  (ImportancePooling.forward, model/pinsage.py:101-150)."""
    if batch is not None:
        ids, counts, nvalid = batch.ids, batch.counts, batch.nvalid
    x = x.contiguous()
    if x.dtype != torch.float32:
        raise TypeError("importance_pool expects fp32 features")
    B, T = int(ids.size(0)), int(ids.size(1))
    N, H = int(x.size(0)), int(x.size(1))
    out = torch.empty((B, H), dtype=torch.float32, device=x.device)
    if max_idx is None:
        max_idx = N - 1
    with torch.cuda.device(x.device):
        nv.call("ps_importance_pool", nv.ptr(x), nv.i64(N), nv.i32(H), nv.ptr(ids), nv.ptr(counts), nv.ptr(wts),
                                             nv.ptr(nvalid), nv.i64(B), nv.i32(T), nv.i64(max_idx), nv.i32(int(renorm)),
                                             nv.ptr(out), nv.stream())
    return out


class PoolFn(torch.autograd.Function):
    """Differentiable ps_importance_pool: the forward is the kernel, the backward scatter-adds w_ij * grad_out_i
    into the feature rows (d out_i / d x_r = sum over the slots j of row i with ids_ij == r of w_ij).  The weights
    are the ones the kernel used: fp32(count / total) or the given fp32 weights, restricted to the kept slots
    (j < nvalid, 0 <= id <= max_idx) and, with renorm, divided by their fp32 sum (model/pinsage.py:123-146)."""

    @staticmethod
    def forward(ctx, x, ids, counts, wts, nvalid, max_idx, renorm):
        out = importance_pool(x, ids=ids, counts=counts, wts=wts, nvalid=nvalid, max_idx=max_idx, renorm=renorm)
        ctx.save_for_backward(ids, counts if counts is not None else wts, nvalid)
        ctx.use_counts = counts is not None
        ctx.n_rows = int(x.size(0))
        ctx.max_idx = ctx.n_rows - 1 if max_idx is None else int(max_idx)
        ctx.renorm = bool(renorm)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ids, cw, nvalid = ctx.saved_tensors
        N = ctx.n_rows
        T = ids.size(1)
        slot = torch.arange(T, device=ids.device)[None, :]
        inrow = slot < nvalid[:, None]
        kept = inrow & (ids >= 0) & (ids <= ctx.max_idx)
        if ctx.use_counts:
            tot = (cw * inrow).sum(dim=1, keepdim=True).clamp(min=1)
            w = (cw.double() / tot.double()).float()
        else:
            w = cw
        w = torch.where(kept, w, torch.zeros_like(w))
        if ctx.renorm:
            s = w.sum(dim=1, keepdim=True)
            w = torch.where(s > 0, w / torch.where(s > 0, s, torch.ones_like(s)), w)
        gx = torch.zeros((N, grad_out.size(1)), dtype=grad_out.dtype, device=grad_out.device)
        rows = torch.where(kept, ids, torch.zeros_like(ids)).long()
        gx.index_add_(0, rows.reshape(-1), (grad_out[:, None, :] * w[:, :, None]).reshape(-1, grad_out.size(1)))
        return gx, None, None, None, None, None, None


def pool(x, ids=None, counts=None, wts=None, nvalid=None, max_idx=None, renorm=True):
    """importance_pool that stays on the autograd tape when `x` needs a gradient."""
    if torch.is_grad_enabled() and x.requires_grad:
        return PoolFn.apply(x, ids, counts, wts, nvalid, max_idx, renorm)
    return importance_pool(x, ids=ids, counts=counts, wts=wts, nvalid=nvalid, max_idx=max_idx, renorm=renorm)
