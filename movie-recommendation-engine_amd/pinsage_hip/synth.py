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


def bipartite_ratings(num_users, num_items, num_ratings, seed=20240601, device="cpu", zipf_c=42.0, zipf_s=1.0, unique=False):
    """-> (edge_index int64[2, 2R], edge_weights fp32[2R]) on `device`.
    unique=True: every (user, item) pair occurs at most once, like real rating logs; pairs drawn twice are
    redrawn until R distinct ones exist (the benchmark graphs keep the with-replacement draw they have been measured on).
    The draw uses torch's generator of `device`: a graph is reproducible per device type, not across CPU and GPU."""
    if unique:
        return _unique_ratings(num_users, num_items, num_ratings, seed, device, zipf_c, zipf_s)
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


def _unique_ratings(U, M, R, seed, device, zipf_c, zipf_s):
    U, M, R = int(U), int(M), int(R)
    if R > U * M:
        raise ValueError("more ratings than (user, item) pairs")
    have = None
    extra = 0
    for attempt in range(64):
        want = R + extra
        ei, ew = bipartite_ratings(U, M, max(want, M), seed=int(seed) + attempt, device=device, zipf_c=zipf_c, zipf_s=zipf_s)
        n = ei.size(1) // 2
        key = (ei[0, :n] - M) * M + ei[1, :n]                  # user * M + item, rows in the generator's shuffled order
        if have is not None:
            key = torch.cat([have[0], key])
            rat = torch.cat([have[1], ew[:n]])
        else:
            rat = ew[:n]
        # first occurrence of every pair, original order kept
        uniq, inv = torch.unique(key, return_inverse=True)
        first = torch.full((uniq.numel(),), key.numel(), dtype=torch.int64, device=key.device)
        first.scatter_reduce_(0, inv, torch.arange(key.numel(), device=key.device), reduce="amin")
        keep = torch.sort(first).values
        key, rat = key[keep], rat[keep]
        if key.numel() >= R:
            key, rat = key[:R], rat[:R]
            u, it = key // M + M, key % M
            u[-1] = M + U - 1 if not bool((key // M == U - 1).any()) else u[-1]     # V = M + U exactly
            return torch.stack([torch.cat([u, it]), torch.cat([it, u])]).to(torch.int64), torch.cat([rat, rat])
        have = (key, rat)
        extra = max(R - key.numel(), 1024) * 2
    raise RuntimeError("could not draw enough distinct pairs")


def write_movielens_csv(out_dir, num_users, num_items, num_ratings, seed=20240601, device="cpu"):
    """Writes the graph as the MovieLens files the reference reads (data/dataset.py:46-70): ratings.csv
    (userId,movieId,rating,timestamp; ids 1-based, rows in the generator's order, distinct pairs) and movies.csv
    (movieId,title,genres).  Returns the two paths.  `ingest.build_graph_from_csv(ratings.csv)` rebuilds the graph with the
    reference's first-appearance numbering."""
    import os

    import numpy as np
    import pandas as pd
    ei, ew = bipartite_ratings(num_users, num_items, num_ratings, seed=seed, device=device, unique=True)
    R = ei.size(1) // 2
    users = (ei[0, :R] - int(num_items) + 1).cpu().numpy()
    items = (ei[1, :R] + 1).cpu().numpy()
    os.makedirs(out_dir, exist_ok=True)
    rp, mp = os.path.join(out_dir, "ratings.csv"), os.path.join(out_dir, "movies.csv")
    pd.DataFrame({"userId": users, "movieId": items, "rating": ew[:R].cpu().numpy().astype(np.float64),
                  "timestamp": 1_500_000_000 + np.arange(R, dtype=np.int64)}).to_csv(rp, index=False)
    genres = ["Action", "Comedy", "Drama", "Sci-Fi", "Thriller", "Romance"]
    mid = np.arange(1, int(num_items) + 1)
    pd.DataFrame({"movieId": mid, "title": [f"Synthetic Movie {i} ({1990 + i % 30})" for i in mid],
                  "genres": [genres[i % 6] + "|" + genres[(i // 6) % 6] for i in mid]}).to_csv(mp, index=False)
    return rp, mp
