"""Hard-negative sampling on the walk kernel.

`NegativeSampler.sample_hard_negatives` of the reference (data/negative_sampler.py:44-99) ranks, per query item,
the nodes visited by 100 `_single_walk` calls by visit count (dict order = first-visit order, stable sort) and
picks negatives from a rank window with `np.random.choice`, everything on the process-global numpy RNG.
Here the 100 walks + ranking of one query are one ps_walk_sample launch (T = 100 * walk_length keeps the whole
ranking); the host-side `np.random.choice` calls are made exactly as in the reference and in the same order, so
with rng='numpy' the returned indices and the final RNG state are identical to the reference's."""
from __future__ import annotations

import numpy as np
import torch

from . import sampling

NUM_WALKS = 100            # `for _ in range(100)` at data/negative_sampler.py:67


def sample_hard_negatives(sampler, num_movies, query_indices, num_hard_samples=5, max_rank=5000, min_rank=2000):
    """-> LongTensor [len(query_indices), num_hard_samples] on query_indices.device."""
    if sampler is None:
        raise ValueError("RandomWalkSampler is required for hard negative sampling")
    all_movie_indices = list(range(int(num_movies)))
    L = int(sampler.walk_length)
    hard = []
    for idx in query_indices.cpu().numpy():
        call = sampler._calls
        sampler._calls += 1
        batch = sampling.walk_sample(sampler.graph, [int(idx)], NUM_WALKS * L, W=NUM_WALKS, L=L, rng=sampler.rng,
                                     seed=sampler.seed, call=call)
        ids, _, nvalid, _ = batch.host()
        ranked = ids[0, : int(nvalid[0])]                                   # by count desc, first visit first
        candidates = [int(v) for v in ranked[min_rank:max_rank] if v < num_movies]   # `item in all_movie_indices`
        if not candidates:
            sampled = np.random.choice(all_movie_indices, size=num_hard_samples, replace=False)
        else:
            sampled = np.random.choice(candidates, size=min(num_hard_samples, len(candidates)), replace=False)
            if len(sampled) < num_hard_samples:
                additional = np.random.choice([i for i in all_movie_indices if i not in sampled],
                                              size=num_hard_samples - len(sampled), replace=False)
                sampled = np.concatenate([sampled, additional])
        hard.append(sampled)
    return torch.tensor(np.asarray(hard), device=query_indices.device)
