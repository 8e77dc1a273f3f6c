#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); the GPU box never runs it.
The fixtures are data only (inputs + the reference's outputs as arrays).  Re-run with
`python tests/golden/make_golden.py`; outputs are deterministic (seeded).

Reference modules exercised: utils/random_walk.py (RandomWalkSampler), model/pinsage.py
(ImportancePooling, PinSage; torch_geometric is absent here, so a 2-name in-process stub
provides the `MessagePassing` base class -- the pooled/MLP branches never touch PyG),
model/aggregators.py, utils/evaluation.py (generate_recommendations).
utils/nearest_neighbors.py cannot be imported (faiss absent) -> no LSH golden exists.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("PINSAGE_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _import_reference():
    tg = types.ModuleType("torch_geometric")
    tgn = types.ModuleType("torch_geometric.nn")
    tgu = types.ModuleType("torch_geometric.utils")

    class MessagePassing(torch.nn.Module):
        def __init__(self, aggr="add"):
            super().__init__()

    tgn.MessagePassing = MessagePassing
    tgu.to_dense_batch = None
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tgn, "torch_geometric.utils": tgu})
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir("/tmp")  # config.py makes ./checkpoints ./output on import
    from utils.random_walk import RandomWalkSampler
    from model.pinsage import PinSage, ImportancePooling
    from model import aggregators
    from utils.evaluation import generate_recommendations
    os.chdir(cwd)
    return RandomWalkSampler, PinSage, ImportancePooling, aggregators, generate_recommendations


def bipartite(M, U, R, seed, weights="half", isolated_items=(), hub=None):
    """edge_index/edge_weights laid out like data/dataset.py:101-116: first R columns
    user->movie, next R movie->user, users offset by M."""
    rs = np.random.RandomState(seed)
    items = rs.randint(0, M, size=R)
    users = rs.randint(0, U, size=R)
    if hub is not None:       # item `hub[0]` rated by users 0..hub[1]-1
        items = np.concatenate([np.full(hub[1], hub[0]), items])
        users = np.concatenate([np.arange(hub[1]) % U, users])
    keep = ~np.isin(items, np.asarray(isolated_items, dtype=np.int64))
    items, users = items[keep], users[keep]
    # make sure the largest user index exists so that V = M + U
    items = np.concatenate([items, [0 if 0 not in isolated_items else 1]])
    users = np.concatenate([users, [U - 1]])
    n = items.shape[0]
    if weights == "half":
        r = rs.randint(1, 11, size=n).astype(np.float32) * 0.5
    elif weights == "float":
        r = (rs.random_sample(n) * 4.9 + 0.1).astype(np.float32)
    else:
        r = None
    u = users + M
    ei = np.stack([np.concatenate([u, items]), np.concatenate([items, u])]).astype(np.int64)
    ew = None if r is None else np.concatenate([r, r]).astype(np.float32)
    return ei, ew


def pad_lists(nbrs, wts, T):
    B = len(nbrs)
    ids = np.full((B, T), -1, dtype=np.int64)
    w = np.zeros((B, T), dtype=np.float64)
    nv = np.zeros(B, dtype=np.int32)
    for i, (a, b) in enumerate(zip(nbrs, wts)):
        assert len(a) == len(b) <= T
        nv[i] = len(a)
        ids[i, :len(a)] = [int(v) for v in a]
        w[i, :len(a)] = b
    return ids, w, nv


def main():
    RandomWalkSampler, PinSage, ImportancePooling, aggregators, generate_recommendations = _import_reference()
    out = {}

    # ---------------- G1: sampler -------------------------------------------------
    cases = [
        # name, graph kwargs, np seed, list of (W, L, T), nodes
        ("A", dict(M=30, U=20, R=200, seed=1, weights="half", isolated_items=(7, 19)), 0,
         [(100, 2, 5), (100, 2, 10), (100, 2, 50)], "items"),
        ("B", dict(M=25, U=18, R=150, seed=2, weights=None), 42, [(10, 3, 10)], "items"),
        ("C", dict(M=12, U=9, R=60, seed=3, weights="float"), 42, [(7, 1, 3), (100, 2, 10)], "items"),
        ("D", dict(M=8, U=1200, R=300, seed=4, weights="half", hub=(0, 1100)), 0, [(100, 2, 10)], "items"),
        ("E", dict(M=20, U=15, R=120, seed=5, weights="half"), 42, [(100, 2, 10)], "tensor_mixed"),
    ]
    for name, gk, npseed, wlts, nodesel in cases:
        ei, ew = bipartite(**gk)
        out[f"g1_{name}_edge_index"] = ei
        if ew is not None:
            out[f"g1_{name}_edge_weights"] = ew
        M = gk["M"]
        np.random.seed(npseed)
        out[f"g1_{name}_npseed"] = np.int64(npseed)
        for ci, (W, L, T) in enumerate(wlts):
            s = RandomWalkSampler(torch.from_numpy(ei), None if ew is None else torch.from_numpy(ew),
                                  walk_length=L, num_walks=W)
            if nodesel == "items":
                nodes = list(range(M))
            else:
                rs = np.random.RandomState(9)
                nodes = torch.from_numpy(rs.permutation(ei.max() + 1)[:17].astype(np.int64))
            nb, wt = s.batch_sample_neighbors(nodes, T)
            ids, w, nv = pad_lists(nb, wt, T)
            pre = f"g1_{name}_{ci}_"
            out[pre + "WLT"] = np.array([W, L, T], dtype=np.int64)
            out[pre + "nodes"] = np.asarray(nodes, dtype=np.int64)
            out[pre + "ids"] = ids
            out[pre + "weights"] = w
            out[pre + "nvalid"] = nv
        out[f"g1_{name}_tail"] = np.float64(np.random.random_sample())  # RNG position check

    # directed graph with a sink: consumption is data dependent (utils/random_walk.py:68-69)
    ei = np.array([[0, 0, 1, 2, 2, 3], [1, 2, 2, 3, 4, 0]], dtype=np.int64)  # node 4 = sink
    ew = np.array([1.0, 2.0, 1.5, 1.0, 3.0, 1.0], dtype=np.float32)
    np.random.seed(7)
    s = RandomWalkSampler(torch.from_numpy(ei), torch.from_numpy(ew), walk_length=3, num_walks=20)
    nb, wt = s.batch_sample_neighbors([0, 1, 2, 3, 4], 4)
    ids, w, nv = pad_lists(nb, wt, 4)
    out.update(g1_S_edge_index=ei, g1_S_edge_weights=ew, g1_S_ids=ids, g1_S_weights=w, g1_S_nvalid=nv,
               g1_S_tail=np.float64(np.random.random_sample()))

    # ---------------- G6: _single_walk sequences ---------------------------------
    ei, ew = bipartite(M=15, U=10, R=80, seed=6, weights="half")
    s = RandomWalkSampler(torch.from_numpy(ei), torch.from_numpy(ew), walk_length=4, num_walks=3)
    np.random.seed(11)
    walks = [s._single_walk(n) for n in [0, 3, 16, 24, 3, 0]]
    out.update(g6_edge_index=ei, g6_edge_weights=ew,
               g6_starts=np.array([0, 3, 16, 24, 3, 0], dtype=np.int64),
               g6_walks=np.array([[int(v) for v in wk] for wk in walks], dtype=np.int64))

    # ---------------- G2: ImportancePooling --------------------------------------
    ei, ew = bipartite(M=30, U=20, R=200, seed=1, weights="half", isolated_items=(7, 19))
    s = RandomWalkSampler(torch.from_numpy(ei), torch.from_numpy(ew), walk_length=2, num_walks=100)
    np.random.seed(3)
    nb, wt = s.batch_sample_neighbors(list(range(30)), 10)
    ids, w, nv = pad_lists(nb, wt, 10)
    torch.manual_seed(0)
    h_items = torch.randn(30, 32)
    h_all = torch.randn(50, 32)
    pool = ImportancePooling()
    out.update(g2_ids=ids, g2_weights=w, g2_nvalid=nv, g2_h_items=h_items.numpy(), g2_h_all=h_all.numpy(),
               g2_out_items=pool(h_items, nb, wt).numpy(), g2_out_all=pool(h_all, nb, wt).numpy())
    # scalar-int / empty / missing-weight rows (model/pinsage.py:110-117,126-129)
    odd_n = [3, [1, 2, 40], [], [5, 6, 7]]
    odd_w = [0.3, [0.5, 0.25, 0.25], [], [0.7]]
    out["g2_odd_out"] = pool(h_items, odd_n, odd_w).numpy()

    # ---------------- G3: PinSage.forward ----------------------------------------
    torch.manual_seed(2)
    model = PinSage(16, 32, 8, num_layers=2).eval()
    x = torch.randn(30, 16)
    np.random.seed(5)
    lists = [s.batch_sample_neighbors(list(range(30)), 10) for _ in range(2)]
    with torch.no_grad():
        e_pool = model(x, edge_index=None, sampled_neighbors=[l[0] for l in lists],
                       importance_weights=[l[1] for l in lists])
        e_shared = model(x, edge_index=None, sampled_neighbors=tuple(lists[0][0]),
                         importance_weights=tuple(lists[0][1]))
        e_mlp = model(x)
        np.random.seed(5)
        e_get = model.get_embeddings(x, s, num_neighbors=10)
    for k, v in model.state_dict().items():
        out["g3_param_" + k] = v.numpy()
    out["g3_x"] = x.numpy()
    for li, (a, b) in enumerate(lists):
        ids, w, nv = pad_lists(a, b, 10)
        out[f"g3_l{li}_ids"], out[f"g3_l{li}_weights"], out[f"g3_l{li}_nvalid"] = ids, w, nv
    out.update(g3_edge_index=ei, g3_edge_weights=ew, g3_e_pool=e_pool.numpy(), g3_e_shared=e_shared.numpy(),
               g3_e_mlp=e_mlp.numpy(), g3_e_get=e_get.numpy())

    # ---------------- G4: aggregators --------------------------------------------
    torch.manual_seed(4)
    f = torch.randn(12, 8)
    nbrs = [[1, 2, 3], [], [0], [4, 5, 6, 7, 8], [9, 10], [11, 0, 1]]
    wts = [[0.5, 0.25, 0.25], [], [2.0], [1.0, 2.0, 3.0, 4.0, 5.0], [0.0, 0.0], [0.1, 0.7, 0.2]]
    ia = aggregators.ImportanceAggregator(8, 6).eval()
    at = aggregators.AttentionAggregator(8).eval()
    mp = aggregators.MaxPoolingAggregator(8, 6).eval()
    with torch.no_grad():
        out["g4_mean"] = aggregators.MeanAggregator()(f, nbrs).numpy()
        out["g4_weighted"] = aggregators.WeightedAggregator()(f, nbrs, wts).numpy()
        out["g4_importance"] = ia(f, nbrs, wts).numpy()
        out["g4_attention"] = at(f, nbrs).numpy()
        out["g4_maxpool"] = mp(f, nbrs).numpy()
    out["g4_features"] = f.numpy()
    T = 5
    ids = np.full((6, T), -1, dtype=np.int64)
    w = np.zeros((6, T), dtype=np.float64)
    nv = np.zeros(6, dtype=np.int32)
    for i, (a, b) in enumerate(zip(nbrs, wts)):
        nv[i] = len(a); ids[i, :len(a)] = a; w[i, :len(b)] = b
    out.update(g4_ids=ids, g4_weights=w, g4_nvalid=nv)
    for nm, m in (("ia", ia), ("at", at), ("mp", mp)):
        for k, v in m.state_dict().items():
            out[f"g4_{nm}_{k}"] = v.numpy()

    # ---------------- G5: exact top-k --------------------------------------------
    torch.manual_seed(6)
    emb = torch.nn.functional.normalize(torch.randn(200, 16), dim=1)
    qs = np.array([0, 17, 199, 42], dtype=np.int64)
    out["g5_emb"] = emb.numpy()
    out["g5_queries"] = qs
    out["g5_top11"] = np.stack([generate_recommendations(emb.clone(), int(q), k=11) for q in qs])
    out["g5_top5_incl"] = np.stack([generate_recommendations(emb.clone(), int(q), k=5, exclude_query=False) for q in qs])

    path = os.path.join(OUT, "reference_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)/1024:.1f} KiB")


if __name__ == "__main__":
    main()
