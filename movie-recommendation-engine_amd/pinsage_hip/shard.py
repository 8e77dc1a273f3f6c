"""Item-range sharding of the PinSage hot path over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI).

The reference is single-process (no distributed code), so this layer is new.  The item
catalogue is cut into `world` contiguous id ranges; the CSR + CDF graph is replicated (walks leave a
shard after one step).  Per GCN layer there is ONE exchange: the all-gather of the hidden rows
h^(l) (layer l+1's importance pooling gathers arbitrary global rows, model/pinsage.py:232).  Neighbour
sampling needs no communication (Philox counters are keyed by the global node id, so results do not
depend on the shard count).  LSH: every rank encodes and keeps its code shard; query codes are
all-gathered (tiny), every rank scans its shard, the [nq, k] candidates are all-gathered and merged with
the same (distance, id) order -> identical to the 1-GPU answer on every rank.

`ops` is the compute backend: HipOps (libpinsage_hip.so) in production; tests inject a CPU checker to
exercise the orchestration under gloo.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import dense, sampling


def shard_range(M, rank, world):
    chunk = (M + world - 1) // world
    lo = min(rank * chunk, M)
    return lo, min(lo + chunk, M), chunk


def _backend(group):
    return dist.get_backend(group) if dist.is_initialized() else None


class _Gather:
    """Handle of an all-gather of row shards; `.wait()` returns [world * chunk, ...]."""

    def __init__(self, out, work=None, dev=None):
        self.out, self.work, self.dev = out, work, dev

    def wait(self):
        if self.work is not None:
            self.work.wait()
        return self.out if self.dev is None else self.out.to(self.dev)


class Comm:
    """The row-shard all-gathers of one ShardedPinSage.  Receive buffers (and the zero-padded send buffer of a short last
    shard) are allocated once per (tag, shape, dtype) and reused by every step: the steady state allocates nothing and
    pads nothing.  A buffer is only rewritten by the next gather with the same tag, which is enqueued behind every kernel
    that read it (RCCL's collective stream waits for the current stream at issue).

    standin=(rank, world): no process group -- this GPU plays ONE rank of `world`; a gather copies the local shard into its
    slot of a buffer of the gathered shape (the other ranks' slots hold copies of the first local rows seen).  What a rank computes per step is
    then exactly what it computes in the real job (bench.py --config 5, tools/shard_sim.py); only the xGMI time is missing."""

    def __init__(self, group=None, standin=None):
        self.group = group
        self.standin = standin
        if standin is not None:
            self.rank, self.world = int(standin[0]), int(standin[1])
        elif dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self._bufs = {}

    def _buf(self, kind, tag, shape, dtype, device, zero=False):
        key = (kind, tag, tuple(shape), dtype, str(device))
        b = self._bufs.get(key)
        if b is None:
            b = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=device)
            self._bufs[key] = b
        return b

    def gather_rows_async(self, t, chunk, tag="rows"):
        """[n_local, ...] (n_local <= chunk) -> handle; rank r's rows start at r * chunk.  With RCCL the collective runs
        asynchronously on its own stream (overlaps kernels enqueued afterwards); gloo has no device all_gather, so
        device tensors are staged through the host synchronously (tests / rehearsal only)."""
        if self.world == 1:
            return _Gather(t)
        rest = tuple(t.shape[1:])
        if self.standin is not None:
            # the local rows go into their slot; the OTHER slots are filled once, when the buffer is made, with copies of the
            # first local rows seen, so that later kernels read data of the right kind from them (all-zero slots made the
            # query-sharded scan of rank 1 of 2 ten times slower than the real thing: half of the table tied at one distance)
            key = ("recv", tag, (self.world * chunk,) + rest, t.dtype, str(t.device))
            fresh = key not in self._bufs
            out = self._buf("recv", tag, (self.world * chunk,) + rest, t.dtype, t.device, zero=True)
            for r in (range(self.world) if fresh else (self.rank,)):
                out[r * chunk: r * chunk + t.size(0)].copy_(t)
            return _Gather(out)
        if t.size(0) == chunk and t.is_contiguous():
            src = t                                          # full shard (every rank but possibly the last): no copy
        else:
            src = self._buf("send", tag, (chunk,) + rest, t.dtype, t.device, zero=True)   # zeroed once; the tail stays zero
            src[: t.size(0)].copy_(t)
        if _backend(self.group) == "gloo" and src.is_cuda:
            out = torch.empty((self.world * chunk,) + rest, dtype=t.dtype)
            dist.all_gather_into_tensor(out, src.cpu().contiguous(), group=self.group)
            return _Gather(out, dev=t.device)
        out = self._buf("recv", tag, (self.world * chunk,) + rest, t.dtype, src.device)
        work = dist.all_gather_into_tensor(out, src, group=self.group, async_op=True)
        return _Gather(out, work)

    def gather_rows(self, t, chunk, tag="rows"):
        return self.gather_rows_async(t, chunk, tag).wait()


def all_gather_rows_async(t, chunk, group=None):
    """module-level form (tests, callers without a ShardedPinSage): fresh receive buffer per call"""
    c = Comm(group)
    return c.gather_rows_async(t, chunk)


def all_gather_rows(t, chunk, group=None):
    return all_gather_rows_async(t, chunk, group).wait()


def fused_self_update(ops, P, i, H):
    """lin_update(cat[lin_self(h), h_neigh]) = h (Wu1 Ws)^T + h_neigh Wu2^T + (Wu1 bs + bu)  (model/pinsage.py:235-239):
    the two stacked linear maps on the self path are composed (one H^3 GEMM + one GEMV, microseconds) instead of
    applied to every row; identical up to fp32 rounding (checked against the reference goldens at 1e-5).
    Nothing is cached here: parameters can be edited in place through `.data` without any version counter moving,
    so a cache keyed on (address, version) would hand back stale weights.  ShardedPinSage, which owns a parameter
    snapshot, keeps the composed weights per instance (`refresh_weights`)."""
    Ws, bs = P[f"convs.{i}.lin_self.weight"], P[f"convs.{i}.lin_self.bias"]
    Wu, bu = P[f"convs.{i}.lin_update.weight"], P[f"convs.{i}.lin_update.bias"]
    Wu1 = Wu[:, :H].contiguous()
    W1 = ops.linear(Wu1, Ws.t().contiguous(), None)                          # [H_out, H_in] = Wu1 @ Ws
    b1 = ops.linear(bs.reshape(1, -1).contiguous(), Wu1, bu).reshape(-1)     # Wu1 @ bs + bu
    return W1, b1


class HipOps:
    """The production backend: thin names over the C-ABI wrappers."""

    def sample(self, sampler, nodes, T, shard=None):
        """shard = (num_items, lo): with the numpy-stream RNG every rank draws the whole catalogue's uniforms
        (same np.random seed on all ranks) and indexes them with the global offsets of its own nodes."""
        if shard is not None and getattr(sampler, "rng", None) == "numpy":
            M, lo = shard
            all_nodes = torch.arange(M, dtype=torch.int64, device=nodes.device)
            return sampler.sample_batch(nodes, T, stream_nodes=(all_nodes, lo))
        return sampler.sample_batch(nodes, T)

    def sample_layers(self, sampler, lo, hi, T, layers, shard=None):
        """all layers' samples of the item range [lo, hi) in one launch (ps_walk_sample_layers)"""
        stream = None
        if shard is not None and getattr(sampler, "rng", None) == "numpy":
            stream = (range(shard[0]), shard[1])
        return sampler.sample_batches(range(lo, hi), T, layers, stream_nodes=stream, defer_state=True)

    def finish(self):
        """completes the deferred np.random state hand-back of sample_layers (numpy-stream mode)"""
        dense.finish_rng_state()

    def pool(self, h_full, batch, max_idx):
        return sampling.importance_pool(h_full, batch, max_idx=max_idx)

    def linear(self, x, W, b, x2=None, W2=None, relu=False, l2norm=False):
        return dense.linear(x, W, b, x2=x2, W2=W2, relu=relu, l2norm=l2norm)

    def lsh_encode(self, x, A):
        return dense.lsh_encode(x, A)

    def stage_weight(self, W):
        return dense.stage_weight(W)

    def lsh_planes(self, codes):
        return dense.lsh_expand(codes)

    def hamming_topk(self, q, codes, k, id_offset, planes=None, out=None):
        return dense.hamming_topk(q, codes, k, id_offset=id_offset, planes=planes, out=out)

    def topk_merge(self, d, i):
        return dense.topk_merge(d, i)

    def topk_merge_records(self, records, nq, k):
        """records uint8 [P, rec]: every rank's [nq*k int64 ids | nq*k int32 distances | pad] as gathered"""
        return dense.topk_merge_records(records, nq, k)


class ShardedPinSage:
    """get_embeddings + LSH build/search over `world` item shards."""

    def __init__(self, params, num_layers, sampler, num_items, ops=None, group=None, standin=None):
        """standin=(rank, world): play one rank of a `world`-rank job on this GPU without a process group (see Comm)."""
        self.P = params                   # state_dict tensors (replicated), on the compute device
        self.num_layers = num_layers
        self.sampler = sampler
        self.M = int(num_items)
        self.ops = ops if ops is not None else HipOps()
        self.group = group
        self.comm = Comm(group, standin=standin)
        self.rank, self.world = self.comm.rank, self.comm.world
        self.lo, self.hi, self.chunk = shard_range(self.M, self.rank, self.world)
        self._nodes = {}                   # device -> arange(lo, hi): made once, not per step
        self.overlap_sampling = False      # measured: no gain on MI355X (2.69 vs 2.65 ms per pass)
        self.overlap_input_proj = os.environ.get("PS_OVERLAP_INPUT_PROJ") == "1"    # numpy-stream mode: input projection beside the MT19937 generator -- measured r03: no
                                           # gain (1.224 vs 1.213 ms per embed pass: the 46 us GEMM hides, the stream hand-over costs it back)
        self.fuse_self = True
        self._streams = {}
        self._fused = {}                   # layer -> (W1, b1) composed from the snapshot `params`
        self._staged = {}                  # name -> weight in kernel staging order (snapshot too); "A": (tensor, version, staged)

    def refresh_weights(self, params=None):
        """Call after changing the parameter tensors (in place or by passing a new dict): drops the composed
        self-path weights so that the next `embed` recomputes them."""
        if params is not None:
            self.P = params
        self._fused.clear()
        self._staged.clear()

    def _fused_layer(self, i, H):
        if i not in self._fused:
            self._fused[i] = fused_self_update(self.ops, self.P, i, H)
        return self._fused[i]

    def _w(self, key, W):
        """The weight matrix `W` in the order the GEMM kernel stages it (dense.stage_weight), made once per parameter snapshot
        (refresh_weights drops them); backends without staged weights get W itself."""
        if not hasattr(self.ops, "stage_weight"):
            return W
        t = self._staged.get(key)
        if t is None:
            t = self._staged[key] = self.ops.stage_weight(W)
        return t

    def _side_stream(self, dev):
        key = str(dev)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=dev)
        return self._streams[key]

    def _sample(self, nodes, T, shard):
        try:
            return self.ops.sample(self.sampler, nodes, T, shard)
        except TypeError:                      # test backends with the 3-argument signature
            return self.ops.sample(self.sampler, nodes, T)

    # -- embeddings: PinSage.get_embeddings (model/pinsage.py:253-280) for the local item range ------
    def embed(self, x_local, T, x_full=None):
        """x_local: this rank's feature rows.  x_full (optional, replicated [M, F] features): layer-0 hidden
        rows of ALL items are then recomputed locally (one small GEMM) instead of all-gathered."""
        try:
            return self._embed(x_local, T, x_full)
        finally:
            if hasattr(self.ops, "finish"):
                self.ops.finish()                # np.random state of the numpy-stream mode, after everything is enqueued

    def _embed(self, x_local, T, x_full=None):
        ops, P = self.ops, self.P
        dev = x_local.device
        fused = (not (self.overlap_sampling and x_local.is_cuda)) and hasattr(ops, "sample_layers") \
            and hasattr(self.sampler, "sample_batches")
        nodes = None
        if not fused:                          # the fused launch takes the item range itself
            nodes = self._nodes.get(str(dev))
            if nodes is None:
                nodes = self._nodes[str(dev)] = torch.arange(self.lo, self.hi, dtype=torch.int64, device=dev)
        # fresh neighbour samples per layer, drawn in the reference's order (:271-275); no communication.
        # Layer i+1's sampling and lin_self are enqueued while layer i's hidden rows are being all-gathered.
        shard = (self.M, self.lo) if self.world > 1 else None
        # Sampling is independent of the dense layers and latency bound, the dense layers are MFMA bound: when
        # everything lives on the GPU the samples of ALL layers are drawn on a side stream (in the reference's
        # order) while the main stream runs the projections; each pooling waits for its own batch only.
        side = self._side_stream(dev) if (self.overlap_sampling and x_local.is_cuda) else None
        batches, ready = [], []
        # numpy-stream mode: the MT19937 generator that opens the sampling (a chain of low-occupancy kernels, 0.2 ms) has
        # nothing to do with the input projection, the only dense work that does not wait for the samples: the projection
        # runs beside it on a second stream (r02 measured generator || GEMMs = 0.91 ms against 1.03 ms back to back, and
        # the SAMPLER beside GEMMs = no gain, which is why only this pair is overlapped)
        early_h = None
        if (self.overlap_input_proj and fused and x_local.is_cuda and getattr(self.sampler, "rng", None) == "numpy"
                and not (x_full is not None and self.world > 1)):
            main = torch.cuda.current_stream(dev)
            aux = self._side_stream(dev)
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                early_h = ops.linear(x_local, self._w("in", P["input_proj.weight"]), P["input_proj.bias"], relu=True)
                early_ready = aux.record_event()
            early_h.record_stream(main)
        if fused:
            batches = list(ops.sample_layers(self.sampler, self.lo, self.hi, T, self.num_layers, shard))
        elif side is not None:
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(self.num_layers):
                    b = self._sample(nodes, T, shard)
                    for t in (b.ids, b.counts, b.nvalid):
                        t.record_stream(main)
                    batches.append(b)
                    ready.append(side.record_event())
        else:
            batches.append(self._sample(nodes, T, shard))
        if early_h is not None:
            torch.cuda.current_stream(dev).wait_event(early_ready)             # behind the sampling kernels enqueued above
        if x_full is not None and self.world > 1:
            h_all = ops.linear(x_full, self._w("in", P["input_proj.weight"]), P["input_proj.bias"], relu=True)
            h = h_all[self.lo:self.hi]
            pending = _Gather(h_all)
        else:
            h = early_h if early_h is not None else ops.linear(x_local, self._w("in", P["input_proj.weight"]), P["input_proj.bias"], relu=True)
            pending = self.comm.gather_rows_async(h, self.chunk, "h")
        for i in range(self.num_layers):
            if side is None and not fused and i + 1 < self.num_layers:
                batches.append(self._sample(nodes, T, shard))       # enqueued while layer i's rows are gathered
            H = h.size(1)
            Wu = P[f"convs.{i}.lin_update.weight"]
            if self.fuse_self:
                W1, b1 = self._fused_layer(i, H)
                a_in = h
            else:
                a_in = ops.linear(h, P[f"convs.{i}.lin_self.weight"], P[f"convs.{i}.lin_self.bias"])
                W1, b1 = Wu[:, :H], P[f"convs.{i}.lin_update.bias"]
            h_full = pending.wait()                                          # the per-layer exchange
            if side is not None:
                torch.cuda.current_stream(dev).wait_event(ready[i])
            h_neigh = ops.pool(h_full, batches[i], self.M - 1)
            h = ops.linear(a_in, self._w(("l1", i), W1), b1, x2=h_neigh, W2=self._w(("l2", i), Wu[:, H:]), relu=True, l2norm=True)
            if i + 1 < self.num_layers:
                pending = self.comm.gather_rows_async(h, self.chunk, "h")
        return ops.linear(h, self._w("out", P["output_proj.weight"]), P["output_proj.bias"], l2norm=True)

    # -- LSH: LSHIndex.build / .search (utils/nearest_neighbors.py:28-68) over code shards -------------
    def _staged_A(self, A):
        """the LSH rotation in kernel staging order, remade when another matrix (or a modified one) is passed"""
        if not hasattr(self.ops, "stage_weight") or not isinstance(A, torch.Tensor):
            return A
        ent = self._staged.get("A")
        if ent is None or ent[0] is not A or ent[1] != A._version:
            ent = self._staged["A"] = (A, A._version, self.ops.stage_weight(A))
        return ent[2]

    # Catalogues whose whole code table is at most this many bytes are searched QUERY-sharded: the table is all-gathered once
    # per index build (SYN-25M: 3.8 MB), every rank scans its own queries over all of it and only final results are exchanged.
    # Scanning all queries over 1 / P of a small table is what the scan is worst at (10 000 x 7 381 x 512 bit: 0.145 ms, 1 250 x
    # 59 047: 0.079 ms, no merge pass, 1 MB instead of 11 MB of candidate records per rank at P = 8).  Larger catalogues (BASELINE
    # config 5: 6.4 GB of codes) keep the code shards of SURVEY 8(e): there a rank's 12.5 M codes fill the scan.
    REPLICATE_CODES_BYTES = 64 << 20

    def build_index(self, emb_local, A):
        self.A = A
        self.codes = self.ops.lsh_encode(emb_local, self._staged_A(A))
        self.codes_all = None
        if self.world > 1 and self.M * int(self.codes.size(1)) <= self.REPLICATE_CODES_BYTES:
            self.codes_all = self.comm.gather_rows(self.codes, self.chunk, "codes")[: self.M]
        scanned = self.codes_all if self.codes_all is not None else self.codes
        # sign planes of the scanned code table for the fp4-MFMA scan (backends without them scan the packed codes)
        self.planes = self.ops.lsh_planes(scanned) if hasattr(self.ops, "lsh_planes") else None
        return self.codes

    def search(self, q_local, k):
        """q_local: this rank's query embeddings [nq_local, D] (every rank contributes the same count).
        Returns (dist int32[nq, k], ids int64[nq, k]) for ALL queries, rank-major, on every rank.
        N > 1: the scan writes its (ids | distances) straight into this rank's candidate record, ONE all-gather exchanges
        the records ([P, record] bytes, receive buffer allocated once) and ps_topk_merge_strided reads the gathered
        records in place: no stack / dtype conversion / reshape kernels around the exchange."""
        ops = self.ops
        qc = ops.lsh_encode(q_local, self._staged_A(self.A))
        nq_local = qc.size(0)
        planes = getattr(self, "planes", None)
        if self.world > 1 and getattr(self, "codes_all", None) is not None:
            # query-sharded (small catalogue): own queries over the whole table, the finished rows exchanged in one collective
            n = nq_local * k
            rec = (12 * n + 15) // 16 * 16
            mine = self.comm._buf("res", "mine", (1, rec), torch.uint8, qc.device, zero=True)
            ids_v = mine[0, : 8 * n].view(torch.int64).view(nq_local, k)
            dist_v = mine[0, 8 * n: 12 * n].view(torch.int32).view(nq_local, k)
            try:
                ops.hamming_topk(qc, self.codes_all, k, 0, planes=planes, out=(dist_v, ids_v))
            except TypeError:                                                   # test backends: plain (q, codes, k, offset)
                d, i = ops.hamming_topk(qc, self.codes_all, k, 0)
                dist_v.copy_(d)
                ids_v.copy_(i)
            allr = self.comm.gather_rows(mine, 1, "res")                        # [P, rec] bytes, rank-major = query-major
            P_ = self.world
            return (allr[:, 8 * n: 12 * n].view(torch.int32).reshape(P_ * nq_local, k),
                    allr[:, : 8 * n].view(torch.int64).reshape(P_ * nq_local, k))
        qc_all = self.comm.gather_rows(qc, nq_local, "qcodes")
        if self.world == 1:
            return (ops.hamming_topk(qc_all, self.codes, k, self.lo, planes=planes) if planes is not None
                    else ops.hamming_topk(qc_all, self.codes, k, self.lo))
        nq = qc_all.size(0)
        n = nq * k
        rec = (12 * n + 15) // 16 * 16                                        # [n x int64 ids | n x int32 dist | pad]
        mine = self.comm._buf("cand", "mine", (1, rec), torch.uint8, qc_all.device, zero=True)
        ids_v = mine[0, : 8 * n].view(torch.int64).view(nq, k)
        dist_v = mine[0, 8 * n: 12 * n].view(torch.int32).view(nq, k)
        try:
            ops.hamming_topk(qc_all, self.codes, k, self.lo, planes=planes, out=(dist_v, ids_v))
        except TypeError:                                                       # test backends: plain (q, codes, k, offset)
            d, i = ops.hamming_topk(qc_all, self.codes, k, self.lo)
            dist_v.copy_(d)
            ids_v.copy_(i)
        allc = self.comm.gather_rows(mine, 1, "cand")                           # [P, rec] bytes
        if hasattr(ops, "topk_merge_records"):
            return ops.topk_merge_records(allc, nq, k)
        d_all = torch.stack([allc[p, 8 * n: 12 * n].view(torch.int32).view(nq, k) for p in range(self.world)])
        i_all = torch.stack([allc[p, : 8 * n].view(torch.int64).view(nq, k) for p in range(self.world)])
        return ops.topk_merge(d_all, i_all)
