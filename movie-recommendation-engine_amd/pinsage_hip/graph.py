"""DeviceGraph: the sampler's adjacency as CSR + per-row fp64 CDF resident in HBM.

Replaces RandomWalkSampler._prepare_adjacency_list (reference utils/random_walk.py:33-50) and
the per-step `weights / weights.sum()` + `np.random.choice` CDF (:72-79)."""
from __future__ import annotations

import os

import torch

from . import native as nv


class DeviceGraph:
    DEST_INFO_MIN_NODES = 8 << 20        # 64 MiB of node records: beyond what the L2s and most of the MALL hold next to the row data

    def __init__(self, edge_index, edge_weights=None, device=None, buckets=None, dest_info=None):
        dev = nv.require_gpu() if device is None else torch.device(device)
        ei = torch.as_tensor(edge_index)
        if ei.dim() != 2 or ei.size(0) != 2:
            raise ValueError("edge_index must have shape [2, num_edges]")
        E = int(ei.size(1))
        ei = ei.to(device=dev, dtype=torch.int64)
        src = ei[0].contiguous()
        dst = ei[1].contiguous()
        if E:
            mx = int(ei.max().item())          # reference: max_node_idx = edge_index.max().item() + 1
            if int(ei.min().item()) < 0:
                raise ValueError("negative node index in edge_index")
        else:
            mx = -1
        V = mx + 1
        if V >= 2 ** 31 or E >= 2 ** 32:
            raise ValueError("graph too large for int32 node ids / uint32 edge ids")
        w = None
        if edge_weights is not None:
            w = torch.as_tensor(edge_weights).to(device=dev, dtype=torch.float32).contiguous()
            if w.numel() != E:
                raise ValueError("edge_weights must have one entry per edge")
        self.device = dev
        self.V, self.E = V, E
        self.rowptr = torch.empty(V + 1, dtype=torch.int64, device=dev)
        self.col = torch.empty(E, dtype=torch.int32, device=dev)
        self.cdf = torch.empty(E, dtype=torch.float64, device=dev)
        wsorted = torch.empty(E, dtype=torch.float64, device=dev)
        L = nv.lib()
        ws_bytes = int(L.ps_csr_build_workspace_bytes(nv.i64(E), nv.i64(V)))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            nv.call("ps_csr_build", nv.ptr(src), nv.ptr(dst), nv.ptr(w), nv.i64(E), nv.i64(V), nv.ptr(self.rowptr),
                                    nv.ptr(self.col), nv.ptr(wsorted), nv.ptr(ws), nv.C.c_size_t(ws_bytes), nv.stream())
            nv.call("ps_cdf_build", nv.ptr(self.rowptr), nv.ptr(wsorted), nv.i64(V), nv.ptr(self.cdf), nv.stream())
            # exact lookup accelerators for the walk kernels: packed (row start, degree) + bucket table
            self.nodeinfo = torch.empty(2 * V, dtype=torch.int32, device=dev)
            self.guide = torch.empty(E, dtype=torch.int32, device=dev)
            if E:
                nv.call("ps_guide_build", nv.ptr(self.rowptr), nv.ptr(self.cdf), nv.i64(V), nv.ptr(self.nodeinfo),
                        nv.ptr(self.guide), nv.stream())
            self.packed = torch.empty(((E + 7) // 8) * 128, dtype=torch.uint8, device=dev)
            if E:
                nv.call("ps_pack_edges", nv.ptr(self.col), nv.ptr(self.cdf), nv.ptr(self.guide), nv.i64(E),
                        nv.ptr(self.packed), nv.stream())
            # bucket records (one sector per later walk step).  Default: the 32-byte half records (four candidates' CDF entries as fp32
            # lower bounds + destinations) -- measured 5 % FASTER than the 64-byte records on SYN-25M too (0.406 against 0.431 ms per
            # two-layer launch, same box, alternating runs) at half the memory (1.6 GB instead of 3.2), and the only form that fits
            # BASELINE config 5 (64 GB beside the 66 GB graph); skipped when even they would not fit.  buckets = "full" / True builds
            # the 64-byte records (five exact candidates; kept as the cross-check), "half", False force a form.  The CSR build's
            # sort workspace is returned first: at 2 x 10^9 edges it is the size of the half records.
            del ws
            self.buckets, self.bucket_bytes = None, 0
            if E and buckets is not False:
                torch.cuda.empty_cache()
                free = torch.cuda.mem_get_info(dev)[0]
                form = {True: "full", "full": "full", "half": "half", None: None}[buckets]
                if form is None and os.environ.get("PS_GRAPH_BUCKETS") in ("full", "half"):     # experiments: force a form
                    form = os.environ["PS_GRAPH_BUCKETS"]
                if form is None:
                    form = "half" if E * 32 < (free * 3) // 5 else None
                if form == "full":
                    self.buckets, self.bucket_bytes = torch.empty(E * 64, dtype=torch.uint8, device=dev), 64
                    nv.call("ps_bucket_build", nv.ptr(self.rowptr), nv.ptr(self.col), nv.ptr(self.cdf), nv.ptr(self.guide),
                            nv.i64(V), nv.i64(E), nv.ptr(self.buckets), nv.stream())
                elif form == "half":
                    self.buckets, self.bucket_bytes = torch.empty(E * 32, dtype=torch.uint8, device=dev), 32
                    nv.call("ps_bucket_build_half", nv.ptr(self.rowptr), nv.ptr(self.col), nv.ptr(self.cdf), nv.ptr(self.guide),
                            nv.i64(V), nv.i64(E), nv.ptr(self.buckets), nv.stream())
            # destination records (8 bytes per edge, staged into LDS with a start row): a walk's second step without the gather of
            # its node record.  Built when the node records are NOT cache resident -- the default catalogue's 1.7 MB of them are L2
            # hits and the records bought nothing there (r03: 0.419 against 0.413 ms, tools/experiments/) -- i.e. for graphs of more
            # than DEST_INFO_MIN_NODES nodes (BASELINE config 5: 110 M nodes = 880 MB of node records), memory permitting;
            # dest_info = True / False (or PS_GRAPH_DEST_INFO=1 / 0) force it.
            self.dest_info = None
            want = dest_info
            if want is None and os.environ.get("PS_GRAPH_DEST_INFO") in ("0", "1"):
                want = os.environ["PS_GRAPH_DEST_INFO"] == "1"
            if E and want is not False:
                free = torch.cuda.mem_get_info(dev)[0]
                if want is True or (V > self.DEST_INFO_MIN_NODES and E * 8 < free // 4):
                    self.dest_info = torch.empty(2 * E, dtype=torch.int32, device=dev)
                    nv.call("ps_dest_info_build", nv.ptr(self.col), nv.ptr(self.nodeinfo), nv.i64(E), nv.i64(V),
                            nv.ptr(self.dest_info), nv.stream())
            flags = torch.zeros(2, dtype=torch.int64, device=dev)
            nv.call("ps_graph_stats", nv.ptr(self.rowptr), nv.ptr(self.col), nv.i64(E), nv.i64(V), nv.ptr(flags),
                                      nv.stream())
            f = flags.tolist()
        self.has_reachable_sink = bool(f[0])
        self.max_degree = int(f[1])
        self.wsorted = wsorted

    def compact(self):
        """Drop the plain `col` / `cdf` / `guide` arrays (16 bytes per edge): the walk kernels read the same values from the
        interleaved 128-byte blocks of `packed` (+ `nodeinfo`, `buckets`), which stay.  For graphs that fill the GPU (BASELINE
        config 5: 32 of 162 GB); `expand()` restores them bit for bit.  ps_walk_paths and host-side inspection need the plain
        arrays."""
        if self.packed is None or self.E == 0:
            return self
        self.col = self.cdf = self.guide = None
        self.wsorted = None
        return self

    def expand(self):
        """Rebuild `col` / `cdf` / `guide` from the packed blocks (exact copies: the blocks hold the same bits)."""
        if self.col is not None or self.E == 0:
            return self
        blk = self.packed.view(-1, 128)
        self.cdf = blk[:, :64].contiguous().view(torch.float64).reshape(-1)[: self.E].contiguous()
        self.col = blk[:, 64:96].contiguous().view(torch.int32).reshape(-1)[: self.E].contiguous()
        self.guide = blk[:, 96:128].contiguous().view(torch.int32).reshape(-1)[: self.E].contiguous()
        return self

    @property
    def walk_flags(self):
        """flags OR-ed into rng_mode of the walk launches (the form of `buckets`)"""
        return nv.PS_WALK_HALF_BUCKETS if self.bucket_bytes == 32 else 0

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in (self.rowptr, self.col, self.cdf, self.nodeinfo, self.guide, self.packed,
                                                         self.buckets, self.dest_info) if t is not None)


class TargetCSR:
    """Edges grouped by target node (what GraphConv.propagate aggregates over): rowptr / col (= source node) /
    perm (position of every CSR slot in the original edge order, for per-edge weights).  Built with the same
    stable sort as the sampler's adjacency; cached per edge_index tensor by the caller."""

    def __init__(self, edge_index, num_nodes):
        dev = nv.require_gpu() if not edge_index.is_cuda else edge_index.device
        ei = edge_index.to(device=dev, dtype=torch.int64)
        if ei.dim() != 2 or ei.size(0) != 2:
            raise ValueError("edge_index must have shape [2, num_edges]")
        E, V = int(ei.size(1)), int(num_nodes)
        if E:
            # torch / PyG raise on such input (index_add_ / x[src]); the kernels below index rowptr[V+1] and
            # x[num_nodes] with these ids, so they are checked once here (the result is cached by the caller)
            lo, hi = int(ei.min().item()), int(ei.max().item())
            if lo < 0 or hi >= V:
                raise IndexError(f"edge_index holds node id {lo if lo < 0 else hi}, outside [0, {V}) "
                                 "(global ids passed with batch-local features?)")
        src, dst = ei[0].contiguous(), ei[1].contiguous()
        self.V, self.E, self.device = V, E, dev
        self.rowptr = torch.empty(V + 1, dtype=torch.int64, device=dev)
        self.col = torch.empty(E, dtype=torch.int32, device=dev)
        order = torch.empty(E, dtype=torch.float64, device=dev)
        L = nv.lib()
        wsb = int(L.ps_csr_build_workspace_bytes(nv.i64(E), nv.i64(V)))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        # the "weight" channel carries the original edge number (exact in fp32 only below 2^24, so it is passed
        # through the fp64 output by building it in two halves)
        idx = torch.arange(E, device=dev)
        lo16 = (idx & 0xFFFF).to(torch.float32).contiguous()
        hi16 = (idx >> 16).to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            nv.call("ps_csr_build", nv.ptr(dst), nv.ptr(src), nv.ptr(lo16), nv.i64(E), nv.i64(V), nv.ptr(self.rowptr),
                    nv.ptr(self.col), nv.ptr(order), nv.ptr(ws), nv.C.c_size_t(wsb), nv.stream())
            lo_sorted = order.clone()
            nv.call("ps_csr_build", nv.ptr(dst), nv.ptr(src), nv.ptr(hi16), nv.i64(E), nv.i64(V), nv.ptr(self.rowptr),
                    nv.ptr(self.col), nv.ptr(order), nv.ptr(ws), nv.C.c_size_t(wsb), nv.stream())
        self.perm = (order.to(torch.int64) << 16) | lo_sorted.to(torch.int64)


def spmm_csr(tcsr, x, val=None):
    """out[r] = sum_e val[e] * x[col[e]] over row r of a TargetCSR (val in CSR slot order)."""
    x = x.contiguous()
    out = torch.empty((tcsr.V, int(x.size(1))), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        nv.call("ps_spmm_csr", nv.ptr(tcsr.rowptr), nv.ptr(tcsr.col), nv.ptr(val), nv.ptr(x), nv.i64(int(x.size(0))),
                nv.i32(int(x.size(1))), nv.i64(tcsr.V), nv.i64(tcsr.E), nv.ptr(out), nv.stream())
    return out
