"""GPU parity of the pooled forward, dense layers, aggregators, LSH and exact search: HIP kernels
through the reference-shaped classes vs the reference's golden outputs and the CPU oracle.
Tolerances: fp32 embeddings rtol 1e-5 / atol 2e-6 (north_star: 1e-5 rel); codes / ids bit-exact."""
import numpy as np
import pytest
import torch

from conftest import bipartite_graph

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 2e-6


def _model_from_golden(g):
    from model.pinsage import PinSage
    m = PinSage(16, 32, 8, num_layers=2)
    sd = {k[len("g3_param_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("g3_param_")}
    assert set(sd) == set(m.state_dict())                      # same parameter names as the reference
    m.load_state_dict(sd)
    return m.eval()


def _lists(g, li):
    ids, w, nv = g[f"g3_l{li}_ids"], g[f"g3_l{li}_weights"], g[f"g3_l{li}_nvalid"]
    nb = [[np.int64(v) for v in ids[i, :nv[i]]] for i in range(ids.shape[0])]
    wt = [w[i, :nv[i]].tolist() for i in range(ids.shape[0])]
    return nb, wt


def test_forward_golden_all_branches(golden):
    g = golden
    m = _model_from_golden(g)
    x = torch.from_numpy(g["g3_x"])
    l0, l1 = _lists(g, 0), _lists(g, 1)
    for dev in ("cpu", "cuda"):                                 # CPU tensors are staged through the GPU
        mm, xx = m.to(dev), x.to(dev)
        with torch.no_grad():
            e = mm(xx, edge_index=None, sampled_neighbors=[l0[0], l1[0]], importance_weights=[l0[1], l1[1]])
            np.testing.assert_allclose(e.cpu().numpy(), g["g3_e_pool"], rtol=RTOL, atol=ATOL)
            e = mm(xx, edge_index=None, sampled_neighbors=tuple(l0[0]), importance_weights=tuple(l0[1]))
            np.testing.assert_allclose(e.cpu().numpy(), g["g3_e_shared"], rtol=RTOL, atol=ATOL)
            e = mm(xx)
            np.testing.assert_allclose(e.cpu().numpy(), g["g3_e_mlp"], rtol=RTOL, atol=ATOL)
            with pytest.warns(UserWarning) if dev == "cpu" else _nullcontext():
                e = mm(xx, [l0[0], l1[0]], [l0[1], l1[1]])      # the drivers' positional call form
            np.testing.assert_allclose(e.cpu().numpy(), g["g3_e_pool"], rtol=RTOL, atol=ATOL)
        # autograd-enabled pooled branch (pool kernel forward + torch dense layers)
        e = mm(xx, edge_index=None, sampled_neighbors=[l0[0], l1[0]], importance_weights=[l0[1], l1[1]])
        np.testing.assert_allclose(e.detach().cpu().numpy(), g["g3_e_pool"], rtol=RTOL, atol=ATOL)
        e.sum().backward()
        assert mm.convs[0].lin_update.weight.grad is not None and mm.input_proj.weight.grad.abs().sum() > 0
        mm.zero_grad()


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def test_get_embeddings_golden(golden):
    from utils.random_walk import RandomWalkSampler
    g = golden
    m = _model_from_golden(g).cuda()
    s = RandomWalkSampler(torch.from_numpy(g["g3_edge_index"]), torch.from_numpy(g["g3_edge_weights"]),
                          walk_length=2, num_walks=100)
    np.random.seed(5)
    with torch.no_grad():
        e = m.get_embeddings(torch.from_numpy(g["g3_x"]).cuda(), s, num_neighbors=10)
    np.testing.assert_allclose(e.cpu().numpy(), g["g3_e_get"], rtol=RTOL, atol=ATOL)


def test_importance_pooling_list_api_golden(golden):
    from model.pinsage import ImportancePooling
    g = golden
    pool = ImportancePooling()
    h = torch.from_numpy(g["g2_h_items"]).cuda()
    odd_n = [3, [1, 2, 40], [], [5, 6, 7]]
    odd_w = [0.3, [0.5, 0.25, 0.25], [], [0.7]]
    np.testing.assert_allclose(pool(h, odd_n, odd_w).cpu().numpy(), g["g2_odd_out"], rtol=RTOL, atol=ATOL)
    ids, w, nv = g["g2_ids"], g["g2_weights"], g["g2_nvalid"]
    nb = [[np.int64(v) for v in ids[i, :nv[i]]] for i in range(30)]
    wt = [w[i, :nv[i]].tolist() for i in range(30)]
    for tag in ("items", "all"):
        out = pool(torch.from_numpy(g[f"g2_h_{tag}"]).cuda(), nb, wt)
        np.testing.assert_allclose(out.cpu().numpy(), g[f"g2_out_{tag}"], rtol=RTOL, atol=ATOL)


def test_importance_pool_shape_fuzz_both_kernels(monkeypatch):
    """ps_importance_pool over shapes around its kernels' boundaries -- T = 1 / 16 / 17 / 50 / 64 (four-rows-per-wave kernel with one
    and four 16-entry pages) and 65 / 100 (one wave per row), H below, at and beyond one 256-column sweep and not a multiple of
    it, visit counts or given fp32 weights, with and without renormalisation, ids that are dropped (beyond max_idx, negative),
    rows with nothing kept, a row count that is not a multiple of four -- against an fp64 restatement of
    ImportancePooling.forward (model/pinsage.py:101-150), and the two kernels against each other."""
    from pinsage_hip import sampling
    rs = np.random.RandomState(11)
    for T, H, B, N in [(1, 32, 7, 50), (10, 256, 1001, 400), (16, 260, 130, 90), (17, 100, 66, 300), (50, 256, 513, 2000),
                       (64, 512, 65, 70), (65, 64, 40, 60), (100, 256, 33, 500), (10, 4, 9, 12)]:
        x = rs.standard_normal((N, H)).astype(np.float32)
        ids = rs.randint(-1, N + N // 3 + 1, size=(B, T)).astype(np.int32)          # some beyond max_idx, some negative
        nvalid = rs.randint(0, T + 1, size=B).astype(np.int32)
        nvalid[rs.randint(0, B, size=max(1, B // 10))] = 0
        counts = rs.randint(1, 30, size=(B, T)).astype(np.int32)
        wts = (rs.random_sample((B, T)) + 0.05).astype(np.float32)
        max_idx = N - 1 - rs.randint(0, 5)
        xt, it, nt = torch.from_numpy(x).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(nvalid).cuda()
        for use_counts in (True, False):
            for renorm in (True, False):
                kw = dict(ids=it, counts=torch.from_numpy(counts).cuda() if use_counts else None,
                          wts=None if use_counts else torch.from_numpy(wts).cuda(), nvalid=nt, max_idx=max_idx, renorm=renorm)
                monkeypatch.delenv("PS_POOL_ROWS_PER_WAVE", raising=False)
                a = sampling.importance_pool(xt, **kw).cpu().numpy()
                monkeypatch.setenv("PS_POOL_ROWS_PER_WAVE", "1")
                b = sampling.importance_pool(xt, **kw).cpu().numpy()
                ref = np.zeros((B, H))
                for i in range(B):
                    k = int(nvalid[i])
                    if use_counts:
                        tot = counts[i, :k].sum()
                        w = np.array([np.float32(np.float64(c) / np.float64(tot)) for c in counts[i, :k]], dtype=np.float32)
                    else:
                        w = wts[i, :k].copy()
                    keep = (ids[i, :k] >= 0) & (ids[i, :k] <= max_idx)
                    w = np.where(keep, w, np.float32(0))
                    sw = np.float32(w.sum(dtype=np.float64))
                    if renorm and sw > 0:
                        w = w / sw
                    ref[i] = (x[np.where(keep, ids[i, :k], 0)].astype(np.float64) * w[:, None].astype(np.float64)).sum(0)
                np.testing.assert_allclose(a, ref, rtol=2e-5, atol=2e-6, err_msg=str((T, H, B, use_counts, renorm)))
                np.testing.assert_allclose(b, ref, rtol=2e-5, atol=2e-6, err_msg=str((T, H, B, use_counts, renorm)))
                if T <= 16:
                    assert np.array_equal(a, b), (T, H, B, use_counts, renorm)          # the same operations in the same order


def test_pool_backward_matches_torch():
    from model.pinsage import ImportancePooling
    torch.manual_seed(0)
    x = torch.randn(40, 24, device="cuda", requires_grad=True)
    nb = [[int(v) for v in np.random.RandomState(i).randint(0, 60, size=i % 7)] for i in range(40)]
    wt = [[float(v) for v in np.random.RandomState(100 + i).random_sample(i % 7) + 0.1] for i in range(40)]
    out = ImportancePooling()(x, nb, wt)
    out.pow(2).sum().backward()
    gx = x.grad.clone()
    x2 = x.detach().clone().requires_grad_(True)
    rows = []
    for a, b in zip(nb, wt):
        v = [(i, w) for i, w in zip(a, b) if i <= 39]
        if not v:
            rows.append(torch.zeros(24, device="cuda"))
            continue
        wv = torch.tensor([w for _, w in v], device="cuda")
        wv = wv / wv.sum()
        rows.append((x2[[i for i, _ in v]] * wv[:, None]).sum(0))
    torch.stack(rows).pow(2).sum().backward()
    np.testing.assert_allclose(gx.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("M,K,N,K2", [(1000, 128, 256, 0), (777, 256, 256, 256), (513, 256, 128, 0), (300, 256, 64, 0),
                                        (65, 20, 6, 0), (129, 33, 300, 7), (5000, 16, 32, 32)])
def test_linear_vs_oracle(M, K, N, K2):
    from oracle import c_oracle as co
    from pinsage_hip import dense
    rs = np.random.RandomState(M + N)
    x = rs.standard_normal((M, K)).astype(np.float32)
    Wfull = (rs.standard_normal((N, K + K2)) / np.sqrt(K + K2)).astype(np.float32)
    b = rs.standard_normal(N).astype(np.float32)
    x2 = rs.standard_normal((M, K2)).astype(np.float32) if K2 else None
    Wd = torch.from_numpy(Wfull).cuda()
    for relu, l2 in ((False, False), (True, False), (True, True)):
        ref = co.linear(x, np.ascontiguousarray(Wfull[:, :K]), b, x2=x2,
                        W2=np.ascontiguousarray(Wfull[:, K:]) if K2 else None, relu=relu, l2norm=l2, threads=8)
        y = dense.linear(torch.from_numpy(x).cuda(), Wd[:, :K], torch.from_numpy(b).cuda(),
                         x2=torch.from_numpy(x2).cuda() if K2 else None, W2=Wd[:, K:] if K2 else None,
                         relu=relu, l2norm=l2)
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
        if not l2:
            # same k-ordered fma chain as the oracle: results are in fact bit-identical before the norm
            assert np.array_equal(y.cpu().numpy(), ref)


def test_aggregators_golden(golden):
    from model import aggregators as A
    g = golden
    f = torch.from_numpy(g["g4_features"]).cuda()
    nbrs = [[1, 2, 3], [], [0], [4, 5, 6, 7, 8], [9, 10], [11, 0, 1]]
    wts = [[0.5, 0.25, 0.25], [], [2.0], [1.0, 2.0, 3.0, 4.0, 5.0], [0.0, 0.0], [0.1, 0.7, 0.2]]
    ia, at, mp = A.ImportanceAggregator(8, 6), A.AttentionAggregator(8), A.MaxPoolingAggregator(8, 6)
    for nm, m in (("ia", ia), ("at", at), ("mp", mp)):
        sd = {k[len(f"g4_{nm}_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"g4_{nm}_")}
        assert set(sd) == set(m.state_dict())
        m.load_state_dict(sd)
        m.cuda().eval()
    with torch.no_grad():
        np.testing.assert_allclose(A.MeanAggregator()(f, nbrs).cpu().numpy(), g["g4_mean"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(A.WeightedAggregator()(f, nbrs, wts).cpu().numpy(), g["g4_weighted"], rtol=RTOL, atol=ATOL)
        got = ia(f, nbrs, wts).cpu().numpy()
        err = np.abs(got - g["g4_importance"]) / (np.abs(g["g4_importance"]) + 1e-6)
        print("ImportanceAggregator max rel err", float(err.max()))
        np.testing.assert_allclose(got, g["g4_importance"], rtol=RTOL, atol=ATOL)             # north_star: 1e-5 rel
        np.testing.assert_allclose(at(f, nbrs).cpu().numpy(), g["g4_attention"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(mp(f, nbrs).cpu().numpy(), g["g4_maxpool"], rtol=1e-5, atol=1e-6)
    with pytest.raises(IndexError):
        A.MeanAggregator()(f, [[99]])


@pytest.mark.parametrize("d,nbits", [(128, 256), (256, 512), (64, 64), (48, 128), (64, 2048), (40, 1312)])   # beyond 1024 bits / not 32 * 2^j bits: the L2-over-signs path
def test_lsh_codes_and_search_bit_exact(d, nbits):
    from oracle import c_oracle as co
    from utils.nearest_neighbors import LSHIndex, lsh_rotation_matrix
    rs = np.random.RandomState(d)
    N = 3000
    emb = rs.standard_normal((N, d)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    emb[100] = emb[7]; emb[2500] = emb[7]                       # exact duplicates -> distance ties, id order
    idx = LSHIndex(d, nbits, 16)
    idx.build(torch.from_numpy(emb))
    assert idx.index.ntotal == N
    A = lsh_rotation_matrix(d, nbits)
    ref_codes = co.lsh_encode(emb, A, threads=8)
    assert np.array_equal(idx.index.codes.cpu().numpy(), ref_codes)            # bit-exact codes
    for k in (1, 10, 11, 50):
        dist, ids = idx.search(emb[:200], k)
        rd, ri = co.hamming_topk(ref_codes[:200], ref_codes, k, threads=8)
        assert dist.dtype == np.float32 and ids.dtype == np.int64
        assert np.array_equal(ids, ri) and np.array_equal(dist, rd)
    d1, i1 = idx.search(emb[7:8], 4)
    assert i1[0, :3].tolist() == [7, 100, 2500] and d1[0, 0] == 0
    # ntotal < k padding like faiss (-1 ids)
    small = LSHIndex(d, nbits, 16)
    small.build(emb[:5])
    d2, i2 = small.search(emb[:3], 8)
    assert np.all(i2[:, 5:] == -1) and np.all(i2[:, :5] >= 0)


def test_topk_merge_equals_single_shard():
    from pinsage_hip import dense
    rs = np.random.RandomState(1)
    codes = torch.from_numpy(rs.randint(0, 256, size=(4000, 32)).astype(np.uint8)).cuda()
    q = codes[:300]
    d0, i0 = dense.hamming_topk(q, codes, 11)
    parts = [dense.hamming_topk(q, codes[s:s + 1000], 11, id_offset=s) for s in range(0, 4000, 1000)]
    dm, im = dense.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(dm, d0) and torch.equal(im, i0)


def test_exact_topk_golden_and_oracle(golden):
    from oracle import c_oracle as co
    from utils.nearest_neighbors import generate_recommendations, ExactIndex
    g = golden
    emb = torch.from_numpy(g["g5_emb"])
    for qi, q in enumerate(g["g5_queries"]):
        assert np.array_equal(generate_recommendations(emb, int(q), k=11), g["g5_top11"][qi])
        assert np.array_equal(generate_recommendations(emb, int(q), k=5, exclude_query=False), g["g5_top5_incl"][qi])
    rs = np.random.RandomState(3)
    E = rs.standard_normal((5000, 64)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    ex = ExactIndex(64)
    ex.build(E)
    q = np.arange(0, 5000, 37)
    vals, ids = ex.search_indices(q, 11)
    rv, ri = co.dot_topk(E, q, 11, threads=8)
    assert np.array_equal(ids.cpu().numpy(), ri)
    np.testing.assert_allclose(vals.cpu().numpy(), rv, rtol=1e-6, atol=1e-7)


def test_device_mt19937_matches_numpy():
    """Serial and jump-ahead (parallel chunk) generation of numpy's global stream, incl. odd word positions,
    skipped prefixes (multi-GPU shards) and the advanced global state."""
    from pinsage_hip import dense
    cases = [(42, 0, 5000, 0), (0, 3, 1249, 0), (123, 311, 300000, 0), (7, 1, 1, 0), (5, 10, 200000, 123457),
             (11, 77, 4096, 5000000), (13, 624, 131072, 0), (17, 5, 1 << 20, 1 << 22), (9, 0, 11809400, 0),
             # chunk windows by matrix-core jump products (33..1024 chunks): after a skipped prefix (base jump first), 33 chunks
             # (one source in the second round), and the largest table that path serves (1008 chunks; beyond: radix rounds)
             (19, 3, 5_000_000, 5_000_000), (21, 0, 2_150_000, 0), (23, 1, 66_000_000, 0), (29, 0, 68_000_000, 0)]
    for seed, burn, n, skip in cases:
        np.random.seed(seed)
        np.random.random_sample(burn)
        if burn % 2:
            np.random.randint(0, 10)                               # odd word position
        st = np.random.get_state()
        np.random.random_sample(skip)
        ref = np.random.random_sample(n)
        tail = np.random.random_sample()
        np.random.set_state(st)
        out = dense.mt19937_random_sample(n, "cuda", skip=skip).cpu().numpy()
        assert np.array_equal(out, ref), (seed, burn, n, skip)
        assert np.random.random_sample() == tail, (seed, burn, n, skip)     # global state advanced identically
    # the serial kernel gives the same stream
    np.random.seed(3)
    a = dense.mt19937_random_sample(200000, "cuda", advance=False, parallel=False)
    b = dense.mt19937_random_sample(200000, "cuda", advance=False, parallel=True)
    assert torch.equal(a, b)
    # one-round windows (the default for 33..512 chunks) and the two radix-32 rounds agree: doubles, raw words, numpy state
    for seed, burn, n in ((31, 0, 11_809_400), (37, 3, 2_200_000), (41, 1, 33_000_000)):
        outs = []
        for one_round in (True, False):
            np.random.seed(seed)
            np.random.random_sample(burn)
            d = dense.mt19937_random_sample(n, "cuda", one_round=one_round)
            t1 = np.random.random_sample()
            np.random.seed(seed)
            np.random.random_sample(burn)
            r = dense.mt19937_random_sample(n, "cuda", raw=True, one_round=one_round)
            t2 = np.random.random_sample()
            outs.append((d, r[:2 * n], t1, t2))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert outs[0][2] == outs[1][2] == outs[0][3] == outs[1][3]
        rs = np.random.RandomState(seed)
        rs.random_sample(burn)
        assert np.array_equal(outs[0][0].cpu().numpy(), rs.random_sample(n)) and rs.random_sample() == outs[0][2]
    # ranged requests (a rank of an item-sharded job reads only its own start nodes' stream positions): the words of the runs
    # equal the whole stream's, and the state handed back is the post-whole-stream state
    n = 23_618_800
    np.random.seed(5)
    full = dense.mt19937_random_sample(n, "cuda", raw=True)
    tail = np.random.random_sample()
    h = n // 2
    for runs in ([(0, n // 8)], [(3 * (h // 8), 4 * (h // 8)), (h + 3 * (h // 8), h + 4 * (h // 8))], [(n - 1000, n)], [(5, 6)],
                 [(n // 3, n // 3 + 10), (n // 3 + 70_000, n // 3 + 70_010), (n - 5, n)], [(h - 40_000, h + 40_000)], [(0, n)]):
        np.random.seed(5)
        part = dense.mt19937_random_sample(n, "cuda", raw=True, ranges=runs)
        assert np.random.random_sample() == tail, runs
        for lo, hi in runs:
            assert torch.equal(part[2 * lo:2 * hi], full[2 * lo:2 * hi]), (runs, lo, hi)
    # chunk windows by doubling (no radix-16 table) and by radix-16 rounds agree, incl. > 16 and > 256 chunks
    for n in (3_000_000, 20_000_000):
        np.random.seed(21)
        c = dense.mt19937_random_sample(n, "cuda", advance=False, radix=False)
        d = dense.mt19937_random_sample(n, "cuda", advance=False, radix=True)
        assert torch.equal(c, d)
        assert np.array_equal(d[-1000:].cpu().numpy(), np.random.RandomState(21).random_sample(n)[-1000:])


def test_inference_style_flow_unchanged_call_sites():
    """The call pattern of the reference's inference driver, written against the drop-in classes: per 1024-item
    batch, `num_layers` x sampler.batch_sample_neighbors(batch_indices, T), the POSITIONAL model(x, neighbors,
    weights) call (inference.py:43-52), then LSHIndex(dim, bits, tables).build / .search(q[1, D], k) and dropping
    the query itself (inference.py:73-76, 120-127).  Checked against the oracle on the same numpy stream."""
    from oracle import c_oracle as co
    from pinsage_hip import synth
    from utils.random_walk import RandomWalkSampler
    from utils.nearest_neighbors import LSHIndex
    from model.pinsage import PinSage
    dev = torch.device("cuda")
    ei, ew = synth.bipartite_ratings(500, 1500, 40000, seed=3)
    M = 1500
    sampler = RandomWalkSampler(ei, ew, walk_length=2, num_walks=100)
    torch.manual_seed(1)
    model = PinSage(32, 64, 32, num_layers=2).to(dev).eval()
    feats = torch.randn(M, 32)
    np.random.seed(42)
    embs = []
    with torch.no_grad():
        for i in range(0, M, 1024):
            batch_indices = torch.arange(i, min(i + 1024, M))
            all_neighbors, all_weights = [], []
            for _ in range(model.num_layers):
                nb, wt = sampler.batch_sample_neighbors(batch_indices, 10)
                all_neighbors.append(nb)
                all_weights.append(wt)
            # batch-local features with global neighbour ids, exactly what the driver passes
            out = model(feats[i:i + 1024].to(dev), all_neighbors, all_weights)
            embs.append(out.cpu())
    emb = torch.cat(embs)
    assert emb.shape == (M, 32)
    # oracle with the same stream and the same batch-local semantics (ids <= len(batch)-1 kept, pinsage.py:124)
    cg = co.Graph(ei.numpy(), ew.numpy())
    rs = np.random.RandomState(42)
    params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    refs = []
    for i in range(0, M, 1024):
        nodes = np.arange(i, min(i + 1024, M))
        layers = []
        for _ in range(2):
            uoff, n = cg.uniform_offsets(nodes, 100, 2)
            ids, counts, nv, _, _, _ = co.walk_sample(cg, nodes, 10, 2, 100, uniforms=rs.random_sample(n))
            layers.append((ids, counts, nv))
        refs.append(co.pinsage_forward(params, feats[i:i + 1024].numpy(), layers))
    np.testing.assert_allclose(emb.numpy(), np.concatenate(refs), rtol=RTOL, atol=ATOL)
    assert np.random.random_sample() == rs.random_sample()
    index = LSHIndex(32, 64, 16)
    index.build(emb)
    q = 17
    _, indices = index.search(emb[q].unsqueeze(0).numpy(), k=11)
    rec = [idx for idx in indices[0] if idx != q][:10]
    assert len(rec) == 10 and q not in rec and indices[0][0] == q


def test_config0_ml100k_shape_end_to_end_vs_oracle():
    """BASELINE configs[0]: ML-100K-shaped graph, d=64, 1 GCN layer, T=5 neighbours, exact brute-force search
    (K = 11): the whole path against the oracle on the same numpy stream."""
    from oracle import c_oracle as co
    from pinsage_hip import synth
    from utils.random_walk import RandomWalkSampler
    from utils.nearest_neighbors import ExactIndex
    from model.pinsage import PinSage
    dev = torch.device("cuda")
    ei, ew = synth.bipartite_ratings(**synth.ML100K, seed=20240601)
    M = synth.ML100K["num_items"]
    sampler = RandomWalkSampler(ei, ew, walk_length=2, num_walks=100)
    torch.manual_seed(2)
    model = PinSage(128, 256, 64, num_layers=1).to(dev).eval()
    torch.manual_seed(1)
    x = torch.randn(M, 128)
    np.random.seed(42)
    with torch.no_grad():
        emb = model.get_embeddings(x.to(dev), sampler, num_neighbors=5)
    cg = co.Graph(ei.numpy(), ew.numpy())
    rs = np.random.RandomState(42)
    uoff, n = cg.uniform_offsets(np.arange(M), 100, 2)
    ids, counts, nv, _, _, _ = co.walk_sample(cg, np.arange(M), 5, 2, 100, uniforms=rs.random_sample(n))
    params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ref = co.pinsage_forward(params, x.numpy(), [(ids, counts, nv)], threads=8)
    np.testing.assert_allclose(emb.cpu().numpy(), ref, rtol=RTOL, atol=ATOL)
    ex = ExactIndex(64)
    ex.build(emb)
    q = np.arange(0, M, 13)
    vals, got = ex.search_indices(q, 11)
    rv, ri = co.dot_topk(emb.cpu().numpy(), q, 11, threads=8)
    np.testing.assert_allclose(vals.cpu().numpy(), rv, rtol=1e-6, atol=1e-7)
    agree = (got.cpu().numpy() == ri).mean()
    assert agree == 1.0, agree


def test_lsh_num_bits_not_multiple_of_32():
    from oracle import c_oracle as co
    from utils.nearest_neighbors import LSHIndex, lsh_rotation_matrix
    rs = np.random.RandomState(8)
    emb = rs.standard_normal((700, 40)).astype(np.float32)
    for nbits in (24, 72, 200):
        idx = LSHIndex(40, nbits, 16)
        idx.build(emb)
        d, i = idx.search(emb[:50], 9)
        codes = co.lsh_encode(emb, lsh_rotation_matrix(40, nbits))           # nbits real bits per code
        rd, ri = co.hamming_topk(codes[:50], codes, 9)
        assert np.array_equal(i, ri) and np.array_equal(d, rd)


# ---- GraphConv edge branch (model/pinsage.py:53-54, 70-92).  torch_geometric is absent here, so the
# reference's propagate cannot be run: parity unpinned by goldens; checked against PyG's documented aggr='add'
# semantics (out[dst] += w_e * x[src]) evaluated in fp64 on the CPU.
@pytest.mark.parametrize("H,weights", [(32, "both"), (256, "edge"), (7, "none"), (64, "imp")])
def test_spmm_csr_matches_scatter_add(H, weights):
    from pinsage_hip import graph as G
    rng = np.random.default_rng(5)
    V, E = 3000, 60000
    src = rng.integers(0, V, E)
    dst = rng.integers(0, V, E)
    dst[:9000] = 17                                           # one long row (cut into atomic slices)
    dst[dst == 5] = 6                                         # and an empty one
    ei = torch.from_numpy(np.stack([src, dst]))
    x = torch.from_numpy(rng.standard_normal((V, H)).astype(np.float32))
    ew = torch.from_numpy(rng.random(E).astype(np.float32)) if weights in ("both", "edge") else None
    iw = torch.from_numpy(rng.random(E).astype(np.float32)) if weights in ("both", "imp") else None
    w64 = torch.ones(E, dtype=torch.float64)
    for t in (ew, iw):
        if t is not None:
            w64 = w64 * t.double()
    ref = torch.zeros((V, H), dtype=torch.float64).index_add_(0, ei[1], x.double()[ei[0]] * w64[:, None])
    tc = G.TargetCSR(ei.cuda(), V)
    # the CSR is the stable sort by target: perm lists the original edge numbers row by row, ascending inside a row
    perm = tc.perm.cpu().numpy()
    assert np.array_equal(perm, np.argsort(dst, kind="stable"))
    assert np.array_equal(tc.col.cpu().numpy(), src[perm])
    val = None if weights == "none" else (w64.float().cuda()[tc.perm]).contiguous()
    out = G.spmm_csr(tc, x.cuda(), val).cpu()
    assert torch.all(out[5] == 0)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-4 if H == 32 else 2e-4)


def test_graphconv_edge_branch_hip_vs_torch():
    from model.pinsage import PinSage
    torch.manual_seed(3)
    V, E = 500, 4000
    ei = torch.randint(0, V, (2, E))
    x = torch.randn(V, 16)
    m = PinSage(16, 32, 8, num_layers=2).eval()
    with torch.no_grad():
        ref = m(x, edge_index=ei)                              # CPU: torch index_add_ path
        out = m.cuda()(x.cuda(), edge_index=ei.cuda()).cpu()    # GPU, no grad: ps_spmm_csr
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)
    # with autograd the differentiable path runs and gradients reach the parameters
    m.train()
    e = m(x.cuda(), edge_index=ei.cuda())
    e.square().sum().backward()
    assert m.convs[0].lin_neigh.weight.grad is not None


@pytest.mark.parametrize("P,k", [(2, 11), (5, 11), (8, 11), (20, 11), (30, 11), (3, 1), (4, 64)])
def test_topk_merge_vs_lexsort(P, k):
    """ps_topk_merge over P candidate lists (16-lane selection for P*k <= 256, one wave per query beyond) vs a numpy
    (distance, id) sort; lists may be short (-1 / INT32_MAX padding) and distances tie across lists."""
    from pinsage_hip import dense
    rs = np.random.RandomState(P * 100 + k)
    nq = 257
    d = rs.randint(0, 40, size=(P, nq, k)).astype(np.int32)
    ids = np.stack([rs.permutation(100000)[:P * k].reshape(P, k) for _ in range(nq)], axis=1).astype(np.int64)
    short = rs.rand(P, nq, k) < 0.15
    ids[short] = -1
    d[short] = 0x7fffffff
    ids[:, 0] = -1; d[:, 0] = 0x7fffffff                                  # a query without any candidate
    dm, im = dense.topk_merge(torch.from_numpy(d).cuda(), torch.from_numpy(ids).cuda())
    dm, im = dm.cpu().numpy(), im.cpu().numpy()
    for q in range(nq):
        dd, ii = d[:, q].reshape(-1), ids[:, q].reshape(-1)
        ok = ii >= 0
        order = np.lexsort((ii[ok], dd[ok]))[:k]
        ed = np.full(k, 0x7fffffff, np.int32); ei = np.full(k, -1, np.int64)
        ed[:order.size] = dd[ok][order]; ei[:order.size] = ii[ok][order]
        assert np.array_equal(dm[q], ed) and np.array_equal(im[q], ei), q


# ---- ADVICE r1 (medium) regressions ----------------------------------------------------------------------------
def test_graphconv_rejects_out_of_range_edge_index():
    """global ids with batch-local features must raise like torch / PyG do, not corrupt device memory"""
    from model.pinsage import PinSage
    from pinsage_hip import graph as G
    m = PinSage(16, 32, 8, num_layers=1).eval().cuda()
    x = torch.randn(100, 16, device="cuda")
    bad = torch.tensor([[0, 5, 250], [1, 2, 3]], device="cuda")            # source id 250 >= 100
    with torch.no_grad(), pytest.raises(IndexError):
        m(x, edge_index=bad)
    with pytest.raises(IndexError):
        G.TargetCSR(torch.tensor([[0, 1], [2, -1]], device="cuda"), 100)
    with torch.no_grad():
        ok = m(x, edge_index=torch.tensor([[0, 5, 99], [1, 2, 3]], device="cuda"))
    assert torch.isfinite(ok).all()


def _torch_weighted_rows(feat, nb, wt, mean=False):
    rows = []
    for a, b in zip(nb, wt):
        if not a:
            rows.append(torch.zeros(feat.size(1), device=feat.device))
            continue
        w = torch.full((len(a),), 1.0 / len(a), device=feat.device) if mean else \
            torch.tensor(b[:len(a)], device=feat.device) / torch.tensor(b[:len(a)], device=feat.device).sum()
        rows.append((feat[a] * w[:, None]).sum(0))
    return torch.stack(rows)


def test_aggregators_are_differentiable_wrt_features():
    """reference model/aggregators.py:13-91,233-287 are plain torch code: gradients reach `features`"""
    from model.aggregators import ImportanceAggregator, MeanAggregator, WeightedAggregator
    torch.manual_seed(4)
    rs = np.random.RandomState(8)
    nb = [[int(v) for v in rs.randint(0, 50, size=i % 6)] for i in range(50)]
    wt = [[float(v) for v in rs.random_sample(i % 6) + 0.05] for i in range(50)]
    base = torch.randn(50, 24, device="cuda")
    for name in ("mean", "weighted", "importance"):
        x = base.clone().requires_grad_(True)
        x2 = base.clone().requires_grad_(True)
        if name == "mean":
            out = MeanAggregator()(x, nb)
            ref = _torch_weighted_rows(x2, nb, wt, mean=True)
        elif name == "weighted":
            out = WeightedAggregator()(x, nb, wt)
            ref = _torch_weighted_rows(x2, nb, wt)
        else:
            agg = ImportanceAggregator(24, 12).cuda()
            out = agg(x, nb, wt)
            pooled = _torch_weighted_rows(x2, nb, wt)
            t = torch.nn.functional.layer_norm(agg.transform(pooled), (12,), agg.norm.weight, agg.norm.bias, agg.norm.eps)
            has = torch.tensor([len(a) > 0 for a in nb], device="cuda")[:, None]
            ref = torch.where(has, t, torch.zeros_like(t))
        assert out.grad_fn is not None, name
        out.pow(2).sum().backward()
        ref.pow(2).sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(x.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=name)


def test_pinsage_pooled_forward_grad_reaches_input_with_frozen_params(golden):
    """all parameters frozen + x.requires_grad: the result must stay on the autograd tape (saliency, feature
    learning); before the fix the fused HIP path returned a detached tensor"""
    g = golden
    m = _model_from_golden(g).cuda()
    for p in m.parameters():
        p.requires_grad_(False)
    nb0, wt0 = _lists(g, 0)
    nb1, wt1 = _lists(g, 1)
    x = torch.from_numpy(g["g3_x"]).cuda().requires_grad_(True)
    e = m(x, sampled_neighbors=[nb0, nb1], importance_weights=[wt0, wt1])
    assert e.grad_fn is not None
    np.testing.assert_allclose(e.detach().cpu().numpy(), g["g3_e_pool"], rtol=RTOL, atol=ATOL)
    e.square().sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0
    with torch.no_grad():
        e2 = m(x, sampled_neighbors=[nb0, nb1], importance_weights=[wt0, wt1])      # fused HIP path, same values
    np.testing.assert_allclose(e2.cpu().numpy(), e.detach().cpu().numpy(), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("M,K,N,K2,relu,l2", [(30000, 128, 256, 0, True, False), (40001, 256, 256, 256, True, True),
                                              (59047, 256, 256, 0, False, True), (24576, 64, 512, 32, False, False)])
def test_dma_gemm_is_bit_identical_to_the_register_staged_gemm(M, K, N, K2, relu, l2):
    """gemm_dma_kernel (operands through LDS-DMA, one barrier per K step; opt-in with PS_GEMM_DMA=1 because it measured
    slower) vs gemm_f32_kernel.  Same MFMA, same k order, same epilogue: outputs must be BIT-identical (the latter is held
    to the fmaf-chain oracle by test_linear_vs_oracle), for ps_linear and for the LSH sign pack."""
    import os
    from pinsage_hip import dense
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    x2 = torch.randn(M, K2, generator=g).cuda() if K2 else None
    W2 = (torch.randn(N, K2, generator=g) / K2 ** 0.5).cuda() if K2 else None
    if N <= 256:
        y_reg = dense.linear(x, W, b, x2=x2, W2=W2, relu=relu, l2norm=l2)
    c_reg = dense.lsh_encode(x, W)
    os.environ["PS_GEMM_DMA"] = "1"
    try:
        if N <= 256:
            y_dma = dense.linear(x, W, b, x2=x2, W2=W2, relu=relu, l2norm=l2)
        c_dma = dense.lsh_encode(x, W)
    finally:
        del os.environ["PS_GEMM_DMA"]
    if N <= 256:
        assert torch.equal(y_dma, y_reg)
    assert torch.equal(c_dma, c_reg)


@pytest.mark.parametrize("M,K,N,K2,relu,l2", [(7381, 256, 256, 256, True, True), (7381, 128, 256, 0, True, False),
                                              (7381, 256, 256, 0, False, True), (2500, 64, 512, 32, False, False),
                                              (33, 32, 129, 96, True, True), (1, 256, 256, 256, True, True),
                                              (16384, 96, 200, 160, False, True)])
def test_shard_gemm_is_bit_identical_to_the_register_staged_gemm(M, K, N, K2, relu, l2, monkeypatch):
    """gemm_shard_kernel (small launches: both operands through LDS-DMA into a ring of two or three images, one barrier per K step)
    vs gemm_f32_kernel / gemm_f32_pkernel (PS_GEMM_SHARD=0), which test_linear_vs_oracle holds to the fmaf-chain oracle: the
    same MFMA, k order and epilogue, so the outputs must be BIT-identical -- for ps_linear and for the LSH sign pack, plain and
    image-order weights (views with a row stride of K + K2), both ring depths, ragged last tiles in both dimensions, K != K2."""
    from pinsage_hip import dense
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K + K2, generator=g) / (K + K2) ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    x2 = torch.randn(M, K2, generator=g).cuda() if K2 else None
    W1, W2 = W[:, :K], (W[:, K:] if K2 else None)
    A = torch.randn(512 if N > 256 else 256, K, generator=g).cuda()

    def run():
        out = [dense.lsh_encode(x, A), dense.lsh_encode(x, dense.stage_weight(A))]
        if N <= 256 or not l2:
            out.append(dense.linear(x, W1, b, x2=x2, W2=W2, relu=relu, l2norm=l2))
            out.append(dense.linear(x, dense.stage_weight(W1), b, x2=x2, W2=dense.stage_weight(W2) if K2 else None, relu=relu, l2norm=l2))
        return out
    monkeypatch.setenv("PS_GEMM_SHARD", "0")
    ref = run()
    for depth in ("2", "3"):
        monkeypatch.setenv("PS_GEMM_SHARD", depth)
        for r, y in zip(ref, run()):
            assert torch.equal(r, y), depth
    monkeypatch.delenv("PS_GEMM_SHARD")
    for r, y in zip(ref, run()):                                   # the launcher's own choice
        assert torch.equal(r, y)


@pytest.mark.parametrize("M,N", [(70, 256), (25000, 256), (90, 128), (50, 64)])
def test_fused_norm_quotient_is_the_ieee_division(M, N):
    """The fused row-normalise epilogue (csrc/dense_mfma.hip) divides by fma corrections of x * RN(1/norm) on its fast path
    and by v_div when a wave sees a value outside that path's domain: both must be the IEEE quotient
    x / max(||x||, 1e-12) of F.normalize (reference model/pinsage.py:240,249), bit for bit.  Rows have at most two
    nonzeros in different lanes, so the norm itself is independent of the summation order (RN(RN(a^2) + RN(b^2))) and
    numpy float32 restates it exactly; W = I makes the GEMM exact.  Covered: ordinary values, quotients next to 1, ratios
    below 2^-60 (slow path), norms above 2^40 and below 2^-40 (slow path), the 1e-12 clamp, zero rows, subnormal
    inputs, a negative zero input -- in all four tile shapes (N = 64 / 128, 256 with few / many rows)."""
    from pinsage_hip import dense
    rs = np.random.RandomState(M + N)
    x = np.zeros((M, N), dtype=np.float32)
    a = (rs.standard_normal(M) * np.exp2(rs.randint(-20, 20, size=M))).astype(np.float32)
    b = (rs.standard_normal(M) * np.exp2(rs.randint(-20, 20, size=M))).astype(np.float32)
    c0, c1 = rs.randint(0, 32, size=M), 32 + rs.randint(0, N - 32, size=M)      # different lanes (c1 != c0 mod 32 enforced below)
    c1 = np.where(c1 % 32 == c0, c1 + 1 - 2 * (c1 % 32 == 31), c1)
    x[np.arange(M), c0], x[np.arange(M), c1] = a, b
    special = {1: (3.0, 4.0), 2: (1.0, 1e-30), 3: (1e25, 3e24), 4: (0.0, 0.0), 5: (1e-20, 0.0), 6: (1e-13, 2e-13),
               7: (1e-42, 3e-41), 8: (-0.0, 5.0), 9: (np.float32(1.0) - np.float32(2.0 ** -24), 2.0 ** -13), 10: (2e-19, 1.0),
               11: (1e-38, 1e-38), 12: (3e38, 0.0), 13: (-7.5, 1e-22)}
    for r, (u, v) in special.items():
        if r < M:
            x[r] = 0
            x[r, 0], x[r, 33] = u, v
    eye = torch.eye(N, device="cuda")
    y = dense.linear(torch.from_numpy(x).cuda(), eye, None, l2norm=True).cpu().numpy()
    x = x + np.float32(0.0)                                      # the fma chain of the GEMM starts at +0: -0 comes out as +0
    with np.errstate(over="ignore", under="ignore"):
        sq = x * x                                               # RN(a^2) per element, float32
        nrm = np.sqrt(sq.sum(axis=1, dtype=np.float32))          # <= 2 nonzeros: one rounding, order-free
        want = x / np.maximum(nrm, np.float32(1e-12))[:, None]
    want = want.astype(np.float32)
    bad = np.argwhere(y.view(np.uint32) != want.view(np.uint32))
    assert y.dtype == np.float32 and bad.size == 0, [(int(r), int(c), float(x[r, c]), float(nrm[r]), float(y[r, c]), float(want[r, c]))
                                                      for r, c in bad[:8]]


def test_staged_weights_are_bit_identical():
    """PS_WPERM: weights stored once in the kernel's staging order (dense.stage_weight -> ps_permute_k) give the same bits as the
    plain matrices for every tile shape the launcher picks (64 x 128 / 64 x 256 / 32 x 256, persistent and one-tile), with and
    without the second operand pair, ReLU, fused row norm; the LSH projection too.  K % 32 != 0 keeps the plain path."""
    from pinsage_hip import dense
    g = torch.Generator().manual_seed(12)
    for M in (59047, 20000, 7381, 65, 1):
        for (K, N, K2, relu, l2) in ((128, 256, 0, True, False), (256, 256, 256, True, True), (256, 256, 0, False, True), (64, 96, 32, False, False)):
            x = torch.randn(M, K, generator=g).cuda()
            W = (torch.randn(N, K + K2, generator=g) / 8).cuda()
            b = torch.randn(N, generator=g).cuda()
            x2 = torch.randn(M, K2, generator=g).cuda() if K2 else None
            W1, W2 = W[:, :K], (W[:, K:] if K2 else None)                       # views with row stride K + K2
            S1, S2 = dense.stage_weight(W1), dense.stage_weight(W2)
            assert isinstance(S1, dense.StagedWeight) and (W2 is None or isinstance(S2, dense.StagedWeight))
            ref = dense.linear(x, W1, b, x2=x2, W2=W2, relu=relu, l2norm=l2)
            got = dense.linear(x, S1, b, x2=x2, W2=S2, relu=relu, l2norm=l2)
            assert torch.equal(ref, got), (M, K, N, K2)
    emb = torch.nn.functional.normalize(torch.randn(5000, 256, generator=g), dim=1).cuda()
    A = torch.randn(512, 256, generator=g).cuda()
    assert torch.equal(dense.lsh_encode(emb, A), dense.lsh_encode(emb, dense.stage_weight(A)))
    Wodd = torch.randn(40, 48, generator=g).cuda()                               # K = 48: not a multiple of 32 -> stays plain
    assert dense.stage_weight(Wodd) is Wodd
    with pytest.raises(ValueError):
        dense.linear(torch.randn(4, 64).cuda(), dense.stage_weight(torch.randn(8, 64).cuda()), None,
                     x2=torch.randn(4, 32).cuda(), W2=torch.randn(8, 32).cuda())
