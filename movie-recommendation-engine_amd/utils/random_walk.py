"""utils.random_walk -- drop-in for the reference module of the same name
(reference utils/random_walk.py), backed by the gfx950 kernels of libpinsage_hip.so.

Same constructor, attributes and methods as the reference's RandomWalkSampler; the python
per-node loops are replaced by one kernel launch per batch:
  _prepare_adjacency_list (:33-50)      -> ps_csr_build + ps_cdf_build (CSR + fp64 CDF in HBM)
  _single_walk (:52-83)                 -> ps_walk_paths
  sample_neighbors / batch_sample_neighbors (:85-142) -> ps_walk_sample
With rng='numpy' (default) the sampler consumes the process-global `np.random` stream exactly
like the reference (same ids, same fp64 weights, same RNG state afterwards).
"""
from __future__ import annotations

import numpy as np
import torch

from pinsage_hip.graph import DeviceGraph
from pinsage_hip import sampling


class RandomWalkSampler:
    def __init__(self, edge_index, edge_weights=None, walk_length=2, num_walks=100, p=1.0, q=1.0,
                 *, device=None, rng="numpy", seed=0):
        self.edge_index = edge_index
        self.edge_weights = edge_weights
        self.walk_length = walk_length
        self.num_walks = num_walks
        self.p = p
        self.q = q
        self.rng = rng
        self.seed = int(seed)
        self._calls = 0
        self._adj_list = None
        self._prepare_adjacency_list(device)

    def _prepare_adjacency_list(self, device=None):
        self.graph = DeviceGraph(self.edge_index, self.edge_weights, device=device)

    @classmethod
    def from_graph(cls, graph, walk_length=2, num_walks=100, rng="numpy", seed=0):
        """A sampler over an already built DeviceGraph (shares the CSR/CDF in HBM)."""
        self = cls.__new__(cls)
        self.edge_index = self.edge_weights = None
        self.walk_length, self.num_walks, self.p, self.q = walk_length, num_walks, 1.0, 1.0
        self.rng, self.seed, self._calls, self._adj_list = rng, int(seed), 0, None
        self.graph = graph
        return self

    @property
    def adj_list(self):
        """The reference's python adjacency list (utils/random_walk.py:39-50), built on demand."""
        if self._adj_list is None:
            g = self.graph
            rowptr = g.rowptr.cpu().numpy()
            col = g.col.cpu().numpy()
            w = g.wsorted.cpu().numpy()
            self._adj_list = [list(zip(col[rowptr[v]:rowptr[v + 1]].tolist(), w[rowptr[v]:rowptr[v + 1]].tolist()))
                              for v in range(g.V)]
        return self._adj_list

    # ---- tensor-native API (what model.pinsage uses) -------------------------------------
    def sample_batch(self, nodes, num_neighbors=10, uniforms=None, stream_nodes=None):
        """-> sampling.NeighborBatch on the device (ids / visit counts / nvalid).  `stream_nodes=(all_nodes, lo)`:
        `nodes` is the slice all_nodes[lo:lo+len(nodes)] of a larger logical batch (item shards, rng='numpy')."""
        call = self._calls
        self._calls += 1
        return sampling.walk_sample(self.graph, nodes, int(num_neighbors), W=self.num_walks, L=self.walk_length,
                                    rng=self.rng, seed=self.seed, call=call, uniforms=uniforms,
                                    stream_nodes=stream_nodes if self.rng == "numpy" else None)

    def sample_batches(self, nodes, num_neighbors=10, layers=2, stream_nodes=None, defer_state=False):
        """`layers` consecutive sample_batch calls over the same nodes in ONE launch (what PinSage.get_embeddings
        needs, model/pinsage.py:271-275) -> list of NeighborBatch, identical to calling sample_batch `layers` times.
        `nodes` may be a python range."""
        call = self._calls
        self._calls += layers
        return sampling.walk_sample_layers(self.graph, nodes, int(num_neighbors), int(layers), W=self.num_walks,
                                           L=self.walk_length, rng=self.rng, seed=self.seed, call=call,
                                           stream_nodes=stream_nodes if self.rng == "numpy" else None,
                                           defer_state=defer_state)

    def single_walks(self, start_nodes):
        """Batched _single_walk: int32[B, walk_length] device tensor (-1 after a sink)."""
        call = self._calls
        self._calls += 1
        return sampling.walk_paths(self.graph, start_nodes, self.walk_length, rng=self.rng, seed=self.seed, call=call)

    # ---- reference API -------------------------------------------------------------------
    def _single_walk(self, start_node):
        path = self.single_walks([int(start_node)])[0].tolist()
        walk = [start_node]
        for n in path:
            if n < 0:
                break
            walk.append(np.int64(n))
        return walk

    def sample_neighbors(self, node_idx, num_neighbors=10):
        nb, wt = self.sample_batch([int(node_idx)], num_neighbors).to_lists()
        return nb[0], wt[0]

    def batch_sample_neighbors(self, nodes, num_neighbors=10):
        if isinstance(nodes, torch.Tensor):
            nodes = nodes.tolist()
        batch = self.sample_batch(nodes, num_neighbors)
        return sampling.LazyNeighborList(batch, "ids"), sampling.LazyNeighborList(batch, "weights")

    # ---- PPR helpers: on the reference's class surface (utils/random_walk.py:144-229) but called by nothing
    # (out of scope for kernels).  Host numpy over the CSR arrays: the reference's in-place
    # ascending-node sweep is reproduced with a sorted frontier of nodes that hold residual mass, so a sweep costs
    # O(frontier edges) instead of O(V); pushes to a larger id are seen in the same sweep, to a smaller id in the
    # next one, exactly as the reference's `for node, res in enumerate(residual)` does.
    def _ppr_host_csr(self):
        if getattr(self, "_ppr_csr", None) is None:
            g = self.graph
            rowptr = g.rowptr.cpu().numpy()
            col = g.col.cpu().numpy().astype(np.int64)
            w = g.wsorted.cpu().numpy()
            share = np.empty_like(w)
            for v in np.flatnonzero(np.diff(rowptr)):
                seg = w[rowptr[v]:rowptr[v + 1]]
                share[rowptr[v]:rowptr[v + 1]] = seg / np.cumsum(seg)[-1]      # python's left-to-right sum()
            self._ppr_csr = (rowptr, col, share)
        return self._ppr_csr

    def _ppr_vector(self, source, size, alpha, sweeps):
        import heapq
        rowptr, col, share = self._ppr_host_csr()
        score = np.zeros(size)
        mass = np.zeros(size)
        score[source] = mass[source] = 1.0
        pending = [source]                                   # nodes with mass > 0 awaiting this sweep
        for _ in range(sweeps):
            heap, queued, later = list(pending), set(pending), set()
            heapq.heapify(heap)
            while heap:
                v = heapq.heappop(heap)
                queued.discard(v)
                m = mass[v]
                if not m > 0:
                    continue
                score[v] += alpha * m
                if v + 1 < rowptr.size:
                    out = (1 - alpha) * m
                    for e in range(rowptr[v], rowptr[v + 1]):
                        t = col[e]
                        mass[t] += out * share[e]
                        if t > v:
                            if t not in queued:
                                queued.add(t)
                                heapq.heappush(heap, t)
                        else:
                            later.add(t)
                mass[v] = 0
                later.discard(v)
            pending = sorted(t for t in later if mass[t] > 0)
        return score

    def compute_ppr_matrix(self, nodes, alpha=0.15, num_iterations=10):
        nodes = nodes.tolist() if isinstance(nodes, torch.Tensor) else list(nodes)
        size = max(self.graph.V, max(nodes) + 1)
        table = {}
        for s in nodes:
            vec = self._ppr_vector(s, size, alpha, num_iterations)
            table.update({(s, int(t)): float(vec[t]) for t in np.flatnonzero(vec > 0)})
        return table

    def precompute_top_neighbors(self, nodes, num_neighbors=10):
        nodes = nodes.tolist() if isinstance(nodes, torch.Tensor) else list(nodes)
        table = self.compute_ppr_matrix(nodes)
        per_source = {}
        for (s, t), v in table.items():                       # dict order = ascending target per source
            per_source.setdefault(s, []).append((t, v))
        best = {}
        for s in nodes:
            ranked = sorted(per_source.get(s, []), key=lambda tv: -tv[1])[:num_neighbors]   # stable, like reverse=True
            total = sum(v for _, v in ranked)
            best[s] = ([t for t, _ in ranked], [v / total for _, v in ranked] if ranked else [])
        return best
