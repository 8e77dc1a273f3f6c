"""Seeded synthetic MovieLens-shaped bipartite graphs (no dataset exists offline).

Layout is the reference's (data/dataset.py:101-116): first R edge columns user->movie, next R
movie->user, user ids offset by M; weights = ratings (half stars), duplicated for both directions.
Item popularity is Zipf-Mandelbrot (max degree ~ 0.33 % of R, like ML-25M's ~81 k of 25 M), user
activity lognormal (median ~70, mean ~154 at ML-25M scale), every item has >= 1 rating."""
from __future__ import annotations

import math

import torch

ML25M = dict(num_users=162541, num_items=59047, num_ratings=25000095)
ML100K = dict(num_users=943, num_items=1682, num_ratings=100000)

_RATING_VALUES = [0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0]
_RATING_P = [.016, .031, .016, .066, .050, .196, .127, .266, .088, .144]


def bipartite_ratings(num_users, num_items, num_ratings, seed=20240601, device="cpu", zipf_c=42.0, zipf_s=1.0):
    """-> (edge_index int64[2, 2R], edge_weights fp32[2R]) on `device`."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    U, M, R = int(num_users), int(num_items), int(num_ratings)
    R_free = max(R - M, 0)
    rank = torch.arange(1, M + 1, dtype=torch.float64, device=dev)
    p_item = (rank + zipf_c).pow(-zipf_s)
    cdf_item = torch.cumsum(p_item / p_item.sum(), 0)
    act = torch.exp(torch.randn(U, generator=g, device=dev, dtype=torch.float64) * 1.255 + math.log(70.0))
    act = act.clamp(20.0, 32000.0)
    cdf_user = torch.cumsum(act / act.sum(), 0)
    items = torch.searchsorted(cdf_item, torch.rand(R_free, generator=g, device=dev, dtype=torch.float64)).clamp_(max=M - 1)
    users = torch.searchsorted(cdf_user, torch.rand(R_free, generator=g, device=dev, dtype=torch.float64)).clamp_(max=U - 1)
    # one guaranteed rating per item, spread through the row order
    items = torch.cat([items, torch.arange(M, device=dev)])
    users = torch.cat([users, torch.searchsorted(cdf_user, torch.rand(M, generator=g, device=dev, dtype=torch.float64)).clamp_(max=U - 1)])
    perm = torch.randperm(items.numel(), generator=g, device=dev)
    items, users = items[perm], users[perm]
    # popularity rank -> item id permutation so that hot items are not the low ids
    idmap = torch.randperm(M, generator=g, device=dev)
    items = idmap[items]
    users[-1] = U - 1                      # make V = M + U exactly
    cdf_r = torch.cumsum(torch.tensor(_RATING_P, dtype=torch.float64, device=dev), 0)
    cdf_r = cdf_r / cdf_r[-1]
    ridx = torch.searchsorted(cdf_r, torch.rand(items.numel(), generator=g, device=dev, dtype=torch.float64)).clamp_(max=9)
    ratings = torch.tensor(_RATING_VALUES, dtype=torch.float32, device=dev)[ridx]
    u = users + M
    edge_index = torch.stack([torch.cat([u, items]), torch.cat([items, u])]).to(torch.int64)
    edge_weights = torch.cat([ratings, ratings])
    return edge_index, edge_weights
