"""model.aggregators -- drop-in for the reference module of the same name (reference
model/aggregators.py).  Same five classes, constructor arguments, parameter names and forward
signatures.  The gather + weighted reduce of Mean / Weighted / Importance aggregators runs on
ps_importance_pool (one kernel for the whole batch instead of a python loop per node);
ImportanceAggregator's Linear runs once on the pooled rows (sum_i w_i (W x_i + b) = W sum_i w_i x_i + b
because the weights sum to 1), on the fp32-MFMA kernel.  Attention / MaxPooling stay torch code
(batched over a padded neighbour tensor); they are not on the north-star path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from pinsage_hip import dense, sampling
from pinsage_hip import native as nv


def _pad(neighbors, n_rows, weights=None, mode="weighted"):
    """python list-of-lists -> padded numpy (ids int32[B,T], w fp32[B,T], nvalid int32[B]).
    No range filtering here (the reference indexes features[node_neighbors] directly: out-of-range
    ids raise IndexError)."""
    B = len(neighbors)
    T = max(1, max((len(r) for r in neighbors), default=1))
    ids = np.full((B, T), -1, dtype=np.int32)
    w = np.zeros((B, T), dtype=np.float32)
    nvalid = np.zeros(B, dtype=np.int32)
    for i, nb in enumerate(neighbors):
        k = len(nb)
        if k == 0:
            continue
        a = np.asarray([int(v) for v in nb], dtype=np.int64)
        if a.max() >= n_rows or a.min() < -n_rows:
            raise IndexError(f"index {int(a.max() if a.max() >= n_rows else a.min())} is out of bounds for "
                             f"dimension 0 with size {n_rows}")
        a = np.where(a < 0, a + n_rows, a)
        ids[i, :k] = a
        nvalid[i] = k
        if mode == "mean" or weights is None:
            w[i, :k] = np.float32(1.0) / np.float32(k)
        else:
            nw = [float(v) for v in list(weights[i])[:k]]            # node_weights[:len(node_neighbors)]
            if len(nw) != k:
                raise RuntimeError(f"The size of tensor a ({k}) must match the size of tensor b ({len(nw)}) "
                                   "at non-singleton dimension 0")
            s = sum(nw)
            if s == 0:
                w[i, :k] = np.float32(1.0) / np.float32(k)           # all-zero weights -> mean (:78-80,:265-267)
            else:
                w[i, :k] = np.asarray([v / s for v in nw], dtype=np.float64).astype(np.float32)
    return ids, w, nvalid


def _pool(features, ids, w, nvalid):
    dev = nv.require_gpu()
    f = (features if features.is_cuda else features.to(dev)).float().contiguous()
    # sampling.pool keeps the result on the autograd tape when `features` needs a gradient (the reference's
    # aggregators are plain differentiable torch code, model/aggregators.py:13-91,233-287)
    return sampling.pool(f, ids=torch.from_numpy(ids).to(f.device), wts=torch.from_numpy(w).to(f.device),
                         nvalid=torch.from_numpy(nvalid).to(f.device), renorm=False)


class MeanAggregator(nn.Module):
    """reference model/aggregators.py:5-39"""

    def __init__(self):
        super().__init__()

    def forward(self, features, neighbors):
        ids, w, nvalid = _pad(neighbors, int(features.size(0)), mode="mean")
        out = _pool(features, ids, w, nvalid)
        return out if features.is_cuda else out.to(features.device)


class WeightedAggregator(nn.Module):
    """reference model/aggregators.py:41-91"""

    def __init__(self):
        super().__init__()

    def forward(self, features, neighbors, weights):
        ids, w, nvalid = _pad(neighbors, int(features.size(0)), weights)
        out = _pool(features, ids, w, nvalid)
        return out if features.is_cuda else out.to(features.device)


class AttentionAggregator(nn.Module):
    """reference model/aggregators.py:93-160 (torch; batched over a padded neighbour tensor)"""

    def __init__(self, in_channels):
        super().__init__()
        self.attention = nn.Sequential(nn.Linear(in_channels * 2, in_channels), nn.ReLU(), nn.Linear(in_channels, 1))

    def forward(self, features, neighbors, self_features=None):
        if self_features is None:
            self_features = features
        ids, _, nvalid = _pad(neighbors, int(features.size(0)), mode="mean")
        dev = features.device
        idt = torch.from_numpy(ids).to(dev).long().clamp(min=0)
        mask = torch.arange(ids.shape[1], device=dev)[None, :] < torch.from_numpy(nvalid).to(dev)[:, None]
        nbr = features[idt]                                               # [B,T,C]
        B = idt.size(0)
        selfe = self_features[:B].unsqueeze(1).expand(-1, idt.size(1), -1)
        scores = self.attention(torch.cat([selfe, nbr], dim=2)).squeeze(2)
        scores = scores.masked_fill(~mask, float("-inf"))
        has = mask.any(dim=1, keepdim=True)
        scores = torch.where(has, scores, torch.zeros_like(scores))       # avoid NaN softmax on empty rows
        attn = torch.softmax(scores, dim=1)
        attn = torch.where(mask, attn, torch.zeros_like(attn))            # rows without neighbours -> zeros
        return (nbr * attn.unsqueeze(2)).sum(dim=1)


class MaxPoolingAggregator(nn.Module):
    """reference model/aggregators.py:162-211 (torch; mlp applied once per node, masked max)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(in_channels, out_channels), nn.ReLU())

    def forward(self, features, neighbors):
        ids, _, nvalid = _pad(neighbors, int(features.size(0)), mode="mean")
        dev = features.device
        idt = torch.from_numpy(ids).to(dev).long().clamp(min=0)
        mask = torch.arange(ids.shape[1], device=dev)[None, :] < torch.from_numpy(nvalid).to(dev)[:, None]
        t = self.mlp(features)[idt]
        t = t.masked_fill(~mask.unsqueeze(2), float("-inf"))
        out = t.max(dim=1).values
        return torch.where(mask.any(dim=1, keepdim=True), out, torch.zeros_like(out))


class ImportanceAggregator(nn.Module):
    """reference model/aggregators.py:213-287: LayerNorm(sum_i w_i (W x_i + b)); zeros for empty rows."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.transform = nn.Linear(in_channels, out_channels)
        self.norm = nn.LayerNorm(out_channels)

    def forward(self, features, neighbors, importance_weights):
        ids, w, nvalid = _pad(neighbors, int(features.size(0)), importance_weights)
        pooled = _pool(features, ids, w, nvalid)
        if torch.is_grad_enabled() and (pooled.requires_grad or any(p.requires_grad for p in self.parameters())):
            t = F.linear(pooled, self.transform.weight.to(pooled.device), self.transform.bias.to(pooled.device))
        else:
            t = dense.linear(pooled, self.transform.weight.detach().to(pooled.device),
                             self.transform.bias.detach().to(pooled.device))
        normed = F.layer_norm(t, self.norm.normalized_shape, self.norm.weight.to(t.device), self.norm.bias.to(t.device),
                              self.norm.eps)
        has = torch.from_numpy(nvalid > 0).to(t.device).unsqueeze(1)
        out = torch.where(has, normed, torch.zeros_like(normed))
        return out if features.is_cuda else out.to(features.device)
