"""SURVEY 8f-1 on the GPU: exact L2 search (faiss.IndexFlatL2), the IVF probe restriction (faiss.IndexIVFFlat behind
WeakANDIndex) and the benchmark harness (reference utils/nearest_neighbors.py:70-254), through the reference-shaped
classes and the bare `ps_l2_topk` wrapper.

faiss is absent (parity unpinned at that boundary): ids are compared with an fp64 numpy restatement of
"k smallest by (squared L2, id)", the IVF restriction with the same restatement over the probed lists only, and
the SYN-25M-size case through size-independent properties (self is nearest, ascending, masked results lie in the
probed lists)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _probe_bits(probe_lists, nlist):
    nq = probe_lists.shape[0]
    bits = np.zeros((nq, (nlist + 31) // 32), dtype=np.uint32)
    for r in range(nq):
        for l in probe_lists[r]:
            bits[r, l >> 5] |= np.uint32(1) << np.uint32(l & 31)
    return bits


def test_l2_topk_and_ivf_masked_search():
    """ps_l2_topk: exact squared-L2 k-NN (IndexFlatL2, utils/nearest_neighbors.py:89,176) and the probed-lists
    restriction (IndexIVFFlat, :92) vs a numpy restatement: ids identical (by (distance, id)), distances 1e-4."""
    from pinsage_hip import dense
    rs = np.random.RandomState(5)
    N, D, nq, k, nlist = 4000, 48, 300, 10, 37
    X = rs.standard_normal((N, D)).astype(np.float32)
    Q = X[:nq] + 0.05 * rs.standard_normal((nq, D)).astype(np.float32)
    assign = rs.randint(0, nlist, size=N).astype(np.int32)
    d2 = ((Q[:, None, :].astype(np.float64) - X[None, :, :].astype(np.float64)) ** 2).sum(-1)
    order = np.argsort(d2, axis=1, kind="stable")[:, :k]
    dist, ids = dense.l2_topk(torch.from_numpy(X).cuda(), torch.from_numpy(Q).cuda(), k)
    assert np.array_equal(ids.cpu().numpy(), order)
    np.testing.assert_allclose(dist.cpu().numpy(), np.take_along_axis(d2, order, 1), rtol=1e-4, atol=1e-4)
    probe_lists = np.stack([rs.permutation(nlist)[:5] for _ in range(nq)])          # 5 random lists per query
    bits = _probe_bits(probe_lists, nlist)
    vis = (probe_lists[:, :, None] == assign[None, None, :]).any(axis=1)
    order_m = np.argsort(np.where(vis, d2, np.inf), axis=1, kind="stable")[:, :k]
    dist, ids = dense.l2_topk(torch.from_numpy(X).cuda(), torch.from_numpy(Q).cuda(), k,
                              assign=torch.from_numpy(assign).cuda(),
                              probe=torch.from_numpy(bits.view(np.int32)).cuda())
    assert np.array_equal(ids.cpu().numpy(), order_m)


def test_l2_topk_fewer_visible_items_than_k_pads():
    """a query whose probed lists hold fewer than k items gets (-1, FLT_MAX) padding, like faiss (:137)."""
    from pinsage_hip import dense
    rs = np.random.RandomState(9)
    X = rs.standard_normal((64, 16)).astype(np.float32)
    assign = np.arange(64, dtype=np.int32) % 16                                     # 4 items per list
    bits = _probe_bits(np.array([[3]]), 16)
    dist, ids = dense.l2_topk(torch.from_numpy(X).cuda(), torch.from_numpy(X[:1]).cuda(), 10,
                              assign=torch.from_numpy(assign).cuda(), probe=torch.from_numpy(bits.view(np.int32)).cuda())
    ids = ids.cpu().numpy()[0]
    assert set(ids[:4].tolist()) == {3, 19, 35, 51} and np.all(ids[4:] == -1)
    assert np.all(dist.cpu().numpy()[0, 4:] >= 3.0e38)


def test_weakand_index_and_benchmark_harness(capsys):
    from utils.nearest_neighbors import WeakANDIndex, benchmark_search_methods
    rs = np.random.RandomState(2)
    centers = rs.standard_normal((40, 32)).astype(np.float32) * 3
    emb = (centers[rs.randint(0, 40, size=6000)] + 0.3 * rs.standard_normal((6000, 32))).astype(np.float32)
    idx = WeakANDIndex(32, num_partitions=50)
    idx.build(emb)
    assert idx.index.ntotal == 6000 and idx.index.is_trained and idx.quantizer.ntotal == 50
    d, i = idx.search(emb[:64], k=10)
    assert d.shape == (64, 10) and i.dtype == np.int64 and d.dtype == np.float32
    assert idx.index.nprobe == 20                                                   # min(num_partitions, 20), :134
    assert np.array_equal(i[:, 0], np.arange(64)) and np.all(np.abs(d[:, 0]) < 1e-3)     # self is nearest
    assert np.all(np.diff(d, axis=1) >= -1e-6)
    res = benchmark_search_methods(torch.from_numpy(emb), emb[:128], k=10)
    assert list(res) == ["exact", "lsh", "ivf"]
    for m in res.values():
        assert set(m) >= {"distances", "indices", "search_time", "index_size", "method"} and m["index_size"] == 6000
        assert m["indices"].shape == (128, 10) and m["search_time"] > 0
    assert [res[m]["method"] for m in res] == ["Exact (Brute Force)", "Locality-Sensitive Hashing", "Weak AND (IVF)"]
    assert "recall" not in res["exact"]
    assert res["ivf"]["recall"] > 0.9 and 0.0 < res["lsh"]["recall"] <= 1.0
    # recall is the reference's definition (:238-246): mean |set(exact_row) & set(method_row)| / k
    want = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(res["exact"]["indices"], res["lsh"]["indices"])])
    assert abs(res["lsh"]["recall"] - want) < 1e-12
    out = capsys.readouterr().out
    for line in ("Benchmarking exact search...", "Benchmarking lsh search...", "Benchmarking ivf search...",
                 "Built LSH index with 6000 embeddings", "Built Weak AND index with 6000 embeddings",
                 "\nBenchmark Results:\n-----------------\nExact (Brute Force):\n  Search time: ",
                 "  Index size: 6000 vectors", "  Locality-Sensitive Hashing recall@10: ",
                 "  Weak AND (IVF) recall@10: "):
        assert line in out, line
    # a subset of methods without 'exact': no recall is computed (:233), unknown names are skipped
    res2 = benchmark_search_methods(emb, emb[:8], k=5, methods=["lsh", "nope"])
    assert list(res2) == ["lsh"] and "recall" not in res2["lsh"] and res2["lsh"]["indices"].shape == (8, 5)


def test_ivf_search_full_catalogue_properties():
    """SYN-25M-sized catalogue (59 047 items, d = 128): WeakANDIndex with the reference's defaults
    (100 partitions, nprobe 20).  Properties: every query that is an item finds itself first at distance ~0,
    distances ascend, every hit lies in one of the query's probed lists, and the result equals the exact L2
    search restricted to those lists."""
    from pinsage_hip import dense
    from utils.nearest_neighbors import WeakANDIndex
    g = torch.Generator().manual_seed(11)
    M, D, nq, k = 59047, 128, 2048, 11
    cent = torch.randn(300, D, generator=g)
    emb = torch.nn.functional.normalize(cent[torch.randint(0, 300, (M,), generator=g)] +
                                        0.35 * torch.randn(M, D, generator=g), dim=1)
    idx = WeakANDIndex(D)
    idx.build(emb)
    q = emb[:nq]
    d, i = idx.search(q, k=k)
    assert d.shape == (nq, k) and np.all(i >= 0)
    assert np.array_equal(i[:, 0], np.arange(nq)) and np.all(np.abs(d[:, 0]) < 1e-4)
    assert np.all(np.diff(d, axis=1) >= -1e-6)
    ivf = idx.index
    lists = ivf._coarse(q.cuda(), ivf.nprobe).cpu().numpy()                         # [nq, 20]
    assign = ivf.assign.cpu().numpy()
    assert np.all((assign[i][:, :, None] == lists[:, None, :]).any(axis=2))         # masked subset of probed lists
    # against the unrestricted exact search: a hit of the exact search that lives in a probed list must be found
    de, ie = dense.l2_topk(emb.cuda(), q.cuda(), k)
    ie = ie.cpu().numpy()
    inprobe = (assign[ie][:, :, None] == lists[:, None, :]).any(axis=2)
    for r in range(0, nq, 97):
        assert set(ie[r][inprobe[r]].tolist()) <= set(i[r].tolist())


@pytest.mark.parametrize("k", [33, 50, 200, 700])
def test_exact_and_l2_search_any_k(k):
    """torch.topk / faiss accept any k (the reference's evaluation asks for hit-rate@500, utils/evaluation.py:5): beyond 32
    keys per lane the kernel sweeps the similarity row again, admitting only keys after the last one emitted.  ids vs fp64
    numpy restatements of "k best by (value, id)"; k > N pads with (-inf / FLT_MAX, -1)."""
    from pinsage_hip import dense
    rs = np.random.RandomState(k)
    N, D, nq = 600, 24, 37
    X = rs.standard_normal((N, D)).astype(np.float32)
    X[5] = X[77]; X[400] = X[77]                                                     # exact ties: broken by id
    Q = X[:nq] + 0.1 * rs.standard_normal((nq, D)).astype(np.float32)
    d2 = ((Q[:, None, :].astype(np.float64) - X[None, :, :].astype(np.float64)) ** 2).sum(-1)
    dist, ids = dense.l2_topk(torch.from_numpy(X).cuda(), torch.from_numpy(Q).cuda(), k)
    ids, dist = ids.cpu().numpy(), dist.cpu().numpy()
    kk = min(k, N)
    got_d2 = np.take_along_axis(d2, np.where(ids[:, :kk] < 0, 0, ids[:, :kk]), 1)
    assert np.all(ids[:, :kk] >= 0) and np.all(np.diff(got_d2, axis=1) >= -1e-3)    # ascending in the fp64 distance
    for r in range(nq):                                                              # a permutation of the fp64 k best (fp32 ties aside)
        assert len(set(ids[r, :kk].tolist())) == kk
        worst = np.sort(d2[r])[kk - 1]
        assert np.all(d2[r, ids[r, :kk]] <= worst + 1e-3)
    if k > N:
        assert np.all(ids[:, N:] == -1) and np.all(dist[:, N:] >= 3.0e38)
    # exact inner-product search over item rows, self excluded (inference.py:112-118)
    E = torch.nn.functional.normalize(torch.from_numpy(X), dim=1).cuda()
    qidx = torch.arange(nq)
    vals, ii = dense.dot_topk(E, qidx, k, exclude_self=True)
    sim = (E.double() @ E.double().T).cpu().numpy()[:nq]
    sim[np.arange(nq), np.arange(nq)] = -np.inf
    ii, vals = ii.cpu().numpy(), vals.cpu().numpy()
    kk = min(k, N - 1)
    for r in range(nq):
        assert len(set(ii[r, :kk].tolist())) == kk and r not in ii[r, :kk]
        best = np.sort(sim[r])[::-1][kk - 1]
        assert np.all(sim[r, ii[r, :kk]] >= best - 1e-5)
        assert np.all(np.diff(vals[r, :kk]) <= 1e-6)                                 # descending


def test_inverted_file_scan_equals_masked_scan():
    """ps_ivf_topk (items sorted by list, only the probed lists' row blocks multiplied and swept) vs the masked full product
    (ps_l2_topk with assign / probe) and the fp64 numpy restatement: identical ids, identical fp32 distances; ragged and
    EMPTY lists, a list larger than a column tile, k below / above the items a query sees, queries in arbitrary order."""
    from pinsage_hip import dense
    rs = np.random.RandomState(17)
    N, D, nq, nlist, nprobe = 5000, 40, 333, 41, 6
    X = rs.standard_normal((N, D)).astype(np.float32)
    X[100] = X[7]; X[4000] = X[7]                                                   # exact ties: broken by ORIGINAL id
    Q = np.concatenate([X[:200] + 0.05 * rs.standard_normal((200, D)).astype(np.float32), rs.standard_normal((nq - 200, D)).astype(np.float32)])
    Q[0] = X[7]
    assign = rs.randint(0, nlist, size=N).astype(np.int32)
    assign[assign == 13] = 12                                                       # list 13 is empty, list 12 large
    assign[:700] = 5                                                                # > 128 x 5 items: spans several column tiles
    assign[[7, 100, 4000]] = 5
    probe_lists = np.stack([rs.permutation(nlist)[:nprobe] for _ in range(nq)]).astype(np.int32)
    probe_lists[0, 0] = 5
    probe_lists[1] = [13, 13, 13, 13, 13, 13]                                       # probes only the empty list -> all padding
    order = np.argsort(assign, kind="stable")
    list_ptr = np.zeros(nlist + 1, dtype=np.int64)
    list_ptr[1:] = np.cumsum(np.bincount(assign, minlength=nlist))
    Xd, Qd = torch.from_numpy(X).cuda(), torch.from_numpy(Q).cuda()
    d2 = ((Q[:, None, :].astype(np.float64) - X[None, :, :].astype(np.float64)) ** 2).sum(-1)
    vis = (probe_lists[:, :, None] == assign[None, None, :]).any(axis=1)
    for k in (1, 10, 40, 300):
        di, ii = dense.ivf_topk(Xd[torch.from_numpy(order).cuda()], torch.from_numpy(list_ptr), torch.from_numpy(order), Qd,
                                torch.from_numpy(probe_lists), k)
        bits = _probe_bits(np.where(probe_lists == 13, 12, probe_lists) if False else probe_lists, nlist)
        dm, im = dense.l2_topk(Xd, Qd, k, assign=torch.from_numpy(assign).cuda(), probe=torch.from_numpy(bits.view(np.int32)).cuda())
        assert torch.equal(ii, im) and torch.equal(di, dm), k
        ii = ii.cpu().numpy()
        nvis = vis.sum(1)
        want = np.argsort(np.where(vis, d2, np.inf), axis=1, kind="stable")[:, :k]
        for r in range(nq):
            kk = min(k, nvis[r])
            assert np.all(ii[r, kk:] == -1)
            if r == 0:                                                               # the three copies of X[7] tie at distance 0
                assert ii[r, :3].tolist() == [7, 100, 4000] or k < 3
        # fp32 vs fp64 order can differ only between near-ties; compare as sets with the distance bound
        for r in range(0, nq, 7):
            kk = min(k, nvis[r])
            if kk:
                assert np.all(d2[r, ii[r, :kk]] <= d2[r, want[r, kk - 1]] + 1e-3)
    assert np.all(ii[1] == -1)


def test_ivf_index_uses_the_inverted_file_and_is_3x_faster_than_flat():
    """same ids as the masked search through the class, and at the reference's defaults (59 047 x 128,
    10 000 queries, 100 lists, nprobe 20) the inverted-file search is ~3.3x faster than the flat L2 search (asserted: >= 2.6x)."""
    import time
    from utils.nearest_neighbors import WeakANDIndex, _DeviceFlatL2
    g = torch.Generator().manual_seed(3)
    M, D, nq, k = 59047, 128, 10000, 11
    cent = torch.randn(300, D, generator=g)
    emb = torch.nn.functional.normalize(cent[torch.randint(0, 300, (M,), generator=g)] + 0.35 * torch.randn(M, D, generator=g), dim=1).cuda()
    idx = WeakANDIndex(D)
    idx.build(emb)
    idx.index.nprobe = 20
    q = emb[torch.randperm(M, generator=g)[:nq].cuda()].contiguous()
    d1, i1 = idx.index.search_device(q, k)
    idx.index.masked = True
    d2, i2 = idx.index.search_device(q, k)
    idx.index.masked = False
    assert torch.equal(i1, i2) and torch.equal(d1, d2)
    flat = _DeviceFlatL2(D)
    flat.add(emb)

    def timed(fn):                       # best of five rounds of five calls, device time (events): boxes differ, and a timing
        best = 1e9                       # assertion must not fail on a slow moment of a shared host
        for _ in range(5):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5 * 1e-3)
        return best

    t_ivf = timed(lambda: idx.index.search_device(q, k))
    t_flat = timed(lambda: flat.search_device(q, k))
    print(f"IVF (nlist 100, nprobe 20) {t_ivf * 1e3:.3f} ms, flat L2 {t_flat * 1e3:.3f} ms, x{t_flat / t_ivf:.2f}")
    # measured 3.2-3.4 x (profiles/README.md, tools/ivf_probe.py); the assertion leaves room for box-to-box variation
    assert t_flat / t_ivf >= 2.6
