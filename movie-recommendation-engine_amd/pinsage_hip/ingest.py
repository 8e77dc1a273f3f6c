"""Graph ingest from MovieLens-style rating rows: the edge_index / edge_weights layout of the
reference's `MovieLensDataset._create_mappings` + `build_graph` (data/dataset.py:77-123), vectorised.

The reference maps ids to indices in FIRST-APPEARANCE order (`Series.unique()` + enumerate, :80-86) and then
looks every row up in a python dict (25 M dict lookups per column on ML-25M).  `pandas.factorize` yields exactly
that numbering in one pass."""
from __future__ import annotations

import numpy as np
import pandas as pd
import torch


def build_graph_from_ratings(user_ids, movie_ids, ratings):
    """-> (edge_index int64[2, 2R], edge_weights fp32[2R], movie_ids_by_index, user_ids_by_index).
    First R columns user->movie, next R movie->user; user indices are offset by the number of movies
    (data/dataset.py:101-116)."""
    movie_idx, movie_uniques = pd.factorize(np.asarray(movie_ids), sort=False)     # first-appearance order
    user_idx, user_uniques = pd.factorize(np.asarray(user_ids), sort=False)
    m = torch.from_numpy(movie_idx.astype(np.int64))
    u = torch.from_numpy(user_idx.astype(np.int64)) + int(len(movie_uniques))
    edge_index = torch.stack([torch.cat([u, m]), torch.cat([m, u])], dim=0)
    r = torch.as_tensor(np.asarray(ratings), dtype=torch.float)
    edge_weights = torch.cat([r, r])
    return edge_index, edge_weights, movie_uniques, user_uniques


def build_graph_from_csv(ratings_csv):
    """ratings.csv with columns userId, movieId, rating (data/dataset.py:46-58)."""
    df = pd.read_csv(ratings_csv, usecols=["userId", "movieId", "rating"])
    return build_graph_from_ratings(df["userId"].values, df["movieId"].values, df["rating"].values)
