"""model.pinsage -- drop-in for the reference module of the same name (reference model/pinsage.py),
with the importance-pooled branch running on the gfx950 kernels of libpinsage_hip.so.

Same classes, constructor arguments, parameter names/shapes (state_dicts of the reference load
unchanged: input_proj, convs.{i}.lin_self / lin_neigh / lin_update, output_proj) and forward
signatures.  `torch_geometric` is not needed: GraphConv keeps the reference's interface and
implements `aggr='add'` message passing itself.

What runs where
  pooled branch (model/pinsage.py:217-240,248-249)   -> ps_importance_pool + ps_linear (fp32 MFMA,
        bias/ReLU/concat/L2-normalise fused); with autograd enabled the same kernel does the
        pooling forward and torch does the (differentiable) dense layers.
  MLP branch (:205-214, what train.py uses)           -> plain torch nn.Linear (autograd), as in the reference.
  edge_index branch (:243-245, GraphConv)             -> ps_spmm_csr over the edges grouped by target node (no grad),
        torch index_add_ when autograd is needed.
"""
from __future__ import annotations

import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from pinsage_hip import dense, sampling
from pinsage_hip import graph as graph_mod
from pinsage_hip import native as nv
from pinsage_hip.shard import HipOps, fused_self_update

_OPS = HipOps()
_TCSR_CACHE = {}


def _target_csr(edge_index, num_nodes):
    """TargetCSR of an edge_index tensor, cached while that tensor is alive and unmodified."""
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes))
    hit = _TCSR_CACHE.get(key)
    if hit is None:
        if len(_TCSR_CACHE) > 8:
            _TCSR_CACHE.clear()
        hit = (graph_mod.TargetCSR(edge_index, num_nodes), edge_index)      # keeps edge_index alive: no address reuse
        _TCSR_CACHE[key] = hit
    return hit[0]


class GraphConv(nn.Module):
    """Graph convolutional layer for PinSage (reference model/pinsage.py:8-92)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.aggr = "add"
        self.lin_self = nn.Linear(in_channels, out_channels)
        self.lin_neigh = nn.Linear(in_channels, out_channels)
        self.lin_update = nn.Linear(2 * out_channels, out_channels)

    def forward(self, x, edge_index=None, edge_weight=None, importance_weights=None):
        x_self = self.lin_self(x)
        if edge_index is None:
            x_neigh = torch.zeros_like(x_self)
        else:
            x_neigh = self.propagate(edge_index, x=self.lin_neigh(x), edge_weight=edge_weight,
                                     importance_weights=importance_weights)
        x_new = self.lin_update(torch.cat([x_self, x_neigh], dim=1))
        x_new = F.relu(x_new)
        return F.normalize(x_new, p=2, dim=1)

    def propagate(self, edge_index, x, edge_weight=None, importance_weights=None):
        """aggr='add', flow source->target: out[dst] += message(x[src])."""
        if not isinstance(edge_index, torch.Tensor):
            raise TypeError("edge_index must be a LongTensor of shape [2, num_edges]")
        if x.is_cuda and not (torch.is_grad_enabled() and (x.requires_grad or
                                                             any(p.requires_grad for p in self.parameters()))):
            # inference on the GPU: CSR row gather-reduce (ps_spmm_csr) over the edges grouped by target node
            tc = _target_csr(edge_index, int(x.size(0)))
            val = None
            if edge_weight is not None or importance_weights is not None:
                w = torch.ones(tc.E, dtype=torch.float32, device=x.device)
                if edge_weight is not None:
                    w = w * edge_weight.to(x.device).float().view(-1)
                if importance_weights is not None:
                    w = w * importance_weights.to(x.device).float().view(-1)
                val = w[tc.perm].contiguous()
            return graph_mod.spmm_csr(tc, x.float(), val)
        src, dst = edge_index[0], edge_index[1]
        msg = self.message(x[src], edge_weight=edge_weight, importance_weights=importance_weights)
        out = torch.zeros_like(x)
        return out.index_add_(0, dst, msg)

    def message(self, x_j, edge_weight=None, importance_weights=None):
        msg = x_j
        if edge_weight is not None:
            msg = msg * edge_weight.view(-1, 1)
        if importance_weights is not None:
            msg = msg * importance_weights.view(-1, 1)
        return msg


# --------------------------------------------------------------------------------------------
def _lists_to_padded(x_rows, neighbors, weights):
    """The reference's per-node filtering (model/pinsage.py:108-134) done once on the host for a
    generic python list-of-lists input -> padded (ids int32[B,T], wts fp32[B,T], nvalid int32[B])."""
    max_idx = x_rows - 1
    rows_i, rows_w = [], []
    for node_neighbors, node_weights in zip(neighbors, weights):
        if isinstance(node_neighbors, (int, np.integer)):          # :110-112
            node_neighbors = [node_neighbors]
            node_weights = [1.0]
        vi, vw = [], []
        if len(node_neighbors):
            for j, idx in enumerate(node_neighbors):
                if isinstance(idx, (int, np.integer)) and idx <= max_idx:   # :124
                    ii = int(idx)
                    if ii < 0:
                        if ii < -x_rows:
                            raise IndexError(f"index {ii} is out of bounds for dimension 0 with size {x_rows}")
                        ii += x_rows                                   # python/torch negative indexing
                    vi.append(ii)
                    vw.append(float(node_weights[j]) if j < len(node_weights) else 1.0)   # :126-129
        rows_i.append(vi)
        rows_w.append(vw)
    B = len(rows_i)
    T = max(1, max((len(r) for r in rows_i), default=1))
    ids = np.full((B, T), -1, dtype=np.int32)
    wts = np.zeros((B, T), dtype=np.float32)
    nvalid = np.zeros(B, dtype=np.int32)
    for i, (vi, vw) in enumerate(zip(rows_i, rows_w)):
        nvalid[i] = len(vi)
        ids[i, :len(vi)] = vi
        wts[i, :len(vi)] = np.asarray(vw, dtype=np.float64).astype(np.float32)   # torch.tensor(python floats) is fp32 (:140)
    return ids, wts, nvalid


def _batch_of(neighbors, weights):
    """The device batch behind a pair of LazyNeighborLists (or a NeighborBatch), else None."""
    if isinstance(neighbors, sampling.NeighborBatch):
        return neighbors
    if isinstance(neighbors, sampling.LazyNeighborList) and isinstance(weights, sampling.LazyNeighborList) \
            and neighbors.batch is weights.batch:
        return neighbors.batch
    return None


class ImportancePooling(nn.Module):
    """Importance pooling for neighbourhood aggregation (reference model/pinsage.py:94-150)."""

    def __init__(self):
        super().__init__()

    def forward(self, x, neighbors, weights):
        dev = nv.require_gpu()
        xg = x if x.is_cuda else x.to(dev)
        xg = xg.float().contiguous()
        batch = _batch_of(neighbors, weights)
        if batch is not None:
            ids, counts, wts, nvalid = batch.ids, batch.counts, None, batch.nvalid
            if ids.device != xg.device:
                ids, counts, nvalid = ids.to(xg.device), counts.to(xg.device), nvalid.to(xg.device)
        else:
            ids_h, wts_h, nv_h = _lists_to_padded(int(x.size(0)), neighbors, weights)
            if ids_h.shape[0] == 0:
                raise RuntimeError("stack expects a non-empty TensorList")     # torch.stack([]) in the reference (:150)
            ids = torch.from_numpy(ids_h).to(xg.device)
            wts = torch.from_numpy(wts_h).to(xg.device)
            nvalid = torch.from_numpy(nv_h).to(xg.device)
            counts = None
        out = sampling.pool(xg, ids=ids, counts=counts, wts=wts, nvalid=nvalid)     # differentiable w.r.t. x
        return out if x.is_cuda else out.to(x.device)


class PinSage(nn.Module):
    """PinSage model (reference model/pinsage.py:151-280)."""

    _warned_misbound = False

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers=2):
        super().__init__()
        self.num_layers = num_layers
        self.input_proj = nn.Linear(in_channels, hidden_channels)
        self.convs = nn.ModuleList([GraphConv(hidden_channels, hidden_channels) for _ in range(num_layers)])
        self.importance_pooling = ImportancePooling()
        self.output_proj = nn.Linear(hidden_channels, out_channels)

    # ---- helpers ---------------------------------------------------------------------------
    def _needs_grad(self, x=None):
        """autograd has to see the layers: a parameter or the input features require a gradient"""
        return torch.is_grad_enabled() and ((x is not None and x.requires_grad) or
                                            any(p.requires_grad for p in self.parameters()))

    def _layer_inputs(self, sampled_neighbors, importance_weights, i):
        per_layer = isinstance(sampled_neighbors, list) and isinstance(importance_weights, list)   # :219
        if per_layer and len(sampled_neighbors) > i:                                               # :223
            return sampled_neighbors[i], importance_weights[i]
        return sampled_neighbors, importance_weights                                               # :228-229

    def forward(self, x, edge_index=None, sampled_neighbors=None, importance_weights=None):
        # Positional (x, neighbors, weights) calls of the reference's own drivers (inference.py:52,
        # main.py:156,215, demo.py:145) bind neighbors to `edge_index`; the reference then fails inside
        # PyG.  We accept that call form: a python list can never be an edge_index tensor.
        if isinstance(edge_index, (list, tuple)) and importance_weights is None and sampled_neighbors is not None:
            if not PinSage._warned_misbound:
                warnings.warn("PinSage.forward(x, neighbors, weights) called positionally: interpreting the "
                              "arguments as (sampled_neighbors, importance_weights)", stacklevel=2)
                PinSage._warned_misbound = True
            edge_index, sampled_neighbors, importance_weights = None, edge_index, sampled_neighbors

        if edge_index is None and (sampled_neighbors is None or importance_weights is None):
            # MLP branch (:205-214): plain torch, differentiable -- what train.py trains
            h = F.relu(self.input_proj(x))
            for i in range(self.num_layers):
                h = F.relu(self.convs[i].lin_self(h))
            return F.normalize(self.output_proj(h), p=2, dim=1)

        if edge_index is not None:
            h = F.relu(self.input_proj(x))
            for i in range(self.num_layers):
                h = self.convs[i](h, edge_index)
            return F.normalize(self.output_proj(h), p=2, dim=1)

        # ---- importance-pooled branch (:217-240, :248-249) ----
        if self._needs_grad(x):
            h = F.relu(self.input_proj(x))
            for i in range(self.num_layers):
                nb, wt = self._layer_inputs(sampled_neighbors, importance_weights, i)
                h_neigh = self.importance_pooling(h, nb, wt)
                h_self = self.convs[i].lin_self(h)
                h = F.relu(self.convs[i].lin_update(torch.cat([h_self, h_neigh], dim=1)))
                h = F.normalize(h, p=2, dim=1)
            return F.normalize(self.output_proj(h), p=2, dim=1)
        return self._forward_pooled_hip(x, sampled_neighbors, importance_weights)

    def _forward_pooled_hip(self, x, sampled_neighbors, importance_weights):
        dev = nv.require_gpu()
        xg = (x if x.is_cuda else x.to(dev)).float().contiguous()
        P = {k: (v if v.device == xg.device else v.to(xg.device)).detach().float() for k, v in self.state_dict().items()}
        h = dense.linear(xg, P["input_proj.weight"], P["input_proj.bias"], relu=True)
        for i in range(self.num_layers):
            nb, wt = self._layer_inputs(sampled_neighbors, importance_weights, i)
            h_neigh = self.importance_pooling(h, nb, wt)
            if h_neigh.size(0) != h.size(0):
                raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Expected size {h.size(0)} "
                                   f"but got size {h_neigh.size(0)} for tensor number 1 in the list.")
            H = h.size(1)
            Wu = P[f"convs.{i}.lin_update.weight"]
            # cat([lin_self(h), h_neigh]) @ Wu.T + bu == h @ (Wu[:, :H] Ws).T + h_neigh @ Wu[:, H:].T + (Wu[:, :H] bs + bu)
            # (:235-240): the stacked self path is composed once per forward, the cat never materialises
            W1, b1 = fused_self_update(_OPS, P, i, H)
            h = dense.linear(h, W1, b1, x2=h_neigh, W2=Wu[:, H:], relu=True, l2norm=True)
        e = dense.linear(h, P["output_proj.weight"], P["output_proj.bias"], l2norm=True)
        return e if x.is_cuda else e.to(x.device)

    def get_embeddings(self, x, random_walk_sampler, num_neighbors=10):
        """reference model/pinsage.py:253-280: fresh neighbour samples per layer, then the pooled forward."""
        all_neighbors, all_weights = [], []
        n = x.size(0)
        if hasattr(random_walk_sampler, "sample_batches"):
            # every layer's fresh sample of nodes 0..n-1 in one launch, same draws in the same order (:271-275)
            # the np.random state hand-back of the numpy-stream mode completes AFTER the forward pass has been enqueued
            # (the host does not wait for the stream generator before launching the kernels that follow it)
            try:
                for batch in random_walk_sampler.sample_batches(range(n), num_neighbors, self.num_layers, defer_state=True):
                    all_neighbors.append(sampling.LazyNeighborList(batch, "ids"))
                    all_weights.append(sampling.LazyNeighborList(batch, "weights"))
                return self.forward(x, edge_index=None, sampled_neighbors=all_neighbors, importance_weights=all_weights)
            finally:
                dense.finish_rng_state()
        for _ in range(self.num_layers):
            if hasattr(random_walk_sampler, "sample_batch"):
                nodes = torch.arange(n, dtype=torch.int64, device=random_walk_sampler.graph.device)
                batch = random_walk_sampler.sample_batch(nodes, num_neighbors)
                neighbors = sampling.LazyNeighborList(batch, "ids")
                weights = sampling.LazyNeighborList(batch, "weights")
            else:
                neighbors, weights = random_walk_sampler.batch_sample_neighbors(list(range(n)), num_neighbors)
            all_neighbors.append(neighbors)
            all_weights.append(weights)
        return self.forward(x, edge_index=None, sampled_neighbors=all_neighbors, importance_weights=all_weights)
