"""BASELINE-size runs (SYN-25M: 59 047 items, 162 541 users, 50 M directed edges).

Oracle parity at full size (`test_full_catalogue_equals_the_oracle`: the multithreaded C oracle does the whole catalogue --
every start item, both GCN layers, the pooled forward, LSH encode, thousands of queries -- in seconds): sampled ids / visit
counts / fp64 weights, LSH codes and top-k (distance, id) bit-exact in both RNG modes, embeddings to 1e-5.  Beside it the
size-independent properties: structural invariants of the CSR / CDF, sampler invariants + determinism + batching / shard
invariance, encode -> search round trips, sharded search == unsharded search."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    from pinsage_hip import synth
    from pinsage_hip.graph import DeviceGraph
    dev = torch.device("cuda")
    ei, ew = synth.bipartite_ratings(**synth.ML25M, device=dev)
    g = DeviceGraph(ei, ew)
    return g, ei, ew


@pytest.fixture(scope="module")
def host_graph(big):
    """host copy of the device-built CSR + CDF for the C oracle (the build itself is held to the oracle's by
    test_hip_sampler.py::test_csr_cdf_bit_exact_vs_oracle and, below, on the heaviest rows)"""
    from oracle import c_oracle as co
    g = big[0]
    return co.Graph.from_arrays(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.cdf.cpu().numpy())


def _oracle_threads():
    from oracle import c_oracle as co
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, co.max_threads(), 64))


@pytest.mark.parametrize("T,D,nbits", [(10, 128, 256), (50, 256, 512), (10, 256, 512)])   # BASELINE configs 2, 3, the headline
def test_full_catalogue_equals_the_oracle(big, host_graph, T, D, nbits):
    """Every start item of SYN-25M x 2 GCN layers against the C oracle (reference utils/random_walk.py:85-142,
    model/pinsage.py:217-249, utils/nearest_neighbors.py:28-68): neighbour ids, visit counts, nvalid and the fp64
    importance weights bit-exact in Philox mode and in the reference's np.random stream mode (incl. the np.random state
    afterwards), embeddings through the class surface (PinSage.get_embeddings) within 1e-5 of the oracle's forward over the
    same samples, LSH codes of those embeddings bit-exact, top-11 (distance, id) of 4 096 queries over all codes bit-exact
    (LSHIndex.build / .search)."""
    from oracle import c_oracle as co
    from pinsage_hip import sampling
    from utils.nearest_neighbors import LSHIndex, lsh_rotation_matrix
    from utils.random_walk import RandomWalkSampler
    from model.pinsage import PinSage
    g = big[0]
    cg = host_graph
    M, W, L, LAYERS, k = 59047, 100, 2, 2, 11
    thr = _oracle_threads()
    nodes = np.arange(M)

    def same(batch, ref):
        ids, counts, nv, wts = ref[:4]
        hi, hc, hn, hw = batch.host()                         # what batch_sample_neighbors' lists are made of
        assert np.array_equal(hi, ids) and np.array_equal(hc, counts) and np.array_equal(hn, nv)
        valid = np.arange(T)[None, :] < nv[:, None]
        assert np.array_equal(hw[valid], wts[valid])          # fp64: count / sum(top counts), random_walk.py:113-115

    # ---- Philox mode (what item shards and config 5 use) ----
    got = sampling.walk_sample_layers(g, range(M), T, LAYERS, W, L, rng="philox", seed=42, call=0)
    for call in range(LAYERS):
        same(got[call], co.walk_sample(cg, nodes, T, L, W, philox=(42, call), threads=thr))
    # the heaviest item row (81 K edges) and the heaviest user row are among the start nodes / second steps above; the hub
    # rows as START nodes of a scattered batch too
    deg = g.rowptr[1:] - g.rowptr[:-1]
    hubs = torch.cat([torch.topk(deg[:M], 64).indices, M + torch.topk(deg[M:], 64).indices])
    hb = sampling.walk_sample(g, hubs, T, W, L, rng="philox", seed=7, call=3)
    same(hb, co.walk_sample(cg, hubs.cpu().numpy(), T, L, W, philox=(7, 3), threads=thr))
    # ---- the reference's RNG mode through the class surface ----
    torch.manual_seed(2)
    model = PinSage(128, 256, D, LAYERS).to(g.device).eval()
    x = torch.randn(M, 128, device=g.device)
    smp = RandomWalkSampler.from_graph(g, L, W, rng="numpy", seed=0)
    np.random.seed(42)
    with torch.no_grad():
        emb = model.get_embeddings(x, smp, T)
    tail = np.random.random_sample()
    rs = np.random.RandomState(42)
    uoff, n = cg.uniform_offsets(nodes, W, L)
    assert n == M * W * L
    layers = [co.walk_sample(cg, nodes, T, L, W, uniforms=rs.random_sample(n), uoff=uoff, threads=thr) for _ in range(LAYERS)]
    assert tail == rs.random_sample()                         # np.random ends where the reference leaves it
    np.random.seed(42)
    got = smp.sample_batches(range(M), T, LAYERS)             # the launch get_embeddings made, again
    for r in range(LAYERS):
        same(got[r], layers[r])
    np.random.seed(42)
    one = smp.sample_batch(torch.arange(M, device=g.device), T)     # and as one batch_sample_neighbors call
    same(one, layers[0])
    # ---- embeddings: the oracle's pooled forward over the same samples ----
    params = {kk: v.detach().cpu().numpy() for kk, v in model.state_dict().items()}
    ref = co.pinsage_forward(params, x.cpu().numpy(), [l[:3] for l in layers], threads=thr)
    np.testing.assert_allclose(emb.cpu().numpy(), ref, rtol=1e-5, atol=2e-6)
    # ---- LSH: codes of the GPU's embeddings and the scan over all of them ----
    idx = LSHIndex(D, nbits, 16)
    idx.build(emb)
    A = lsh_rotation_matrix(D, nbits)
    codes = co.lsh_encode(emb.cpu().numpy(), A, threads=thr)
    assert np.array_equal(idx.index.codes.cpu().numpy(), codes)
    q = np.concatenate([np.arange(2048), np.arange(2048, M, (M - 2048) // 2048)[:2048]])
    dist, ids = idx.search(emb[torch.from_numpy(q).to(g.device)], k)
    rd, ri = co.hamming_topk(codes[q], codes, k, threads=thr)
    assert np.array_equal(ids, ri) and np.array_equal(dist, rd)
    assert ids.dtype == np.int64 and dist.dtype == np.float32


def test_graph_invariants_full_size(big):
    g, ei, ew = big
    M = 59047
    assert g.V == 59047 + 162541 and g.E == ei.size(1) == 2 * 25000095
    deg = g.rowptr[1:] - g.rowptr[:-1]
    assert int(deg.min()) >= 1 and not g.has_reachable_sink and g.max_degree == int(deg.max())
    assert torch.equal(deg, torch.bincount(ei[0], minlength=g.V))
    # bipartite: item rows point at users and vice versa
    row_of_edge = torch.repeat_interleave(torch.arange(g.V, device=deg.device), deg)
    assert bool(((row_of_edge < M) != (g.col.long() < M)).all())
    # every row's CDF is non-decreasing, ends at exactly 1.0, first entry > 0
    last = g.cdf[(g.rowptr[1:] - 1)]
    assert bool((last == 1.0).all())
    d = g.cdf[1:] - g.cdf[:-1]
    is_row_start = torch.zeros(g.E, dtype=torch.bool, device=d.device)
    is_row_start[g.rowptr[:-1]] = True
    assert bool((d[~is_row_start[1:]] >= 0).all()) and bool((g.cdf[g.rowptr[:-1]] > 0).all())
    # guide: a valid lower bound of every bucket (start <= searchsorted(cdf, j/deg))
    e = torch.randint(0, g.E, (200000,), device=d.device)
    row = row_of_edge[e]
    lo = g.rowptr[row]
    j = e - lo
    start = lo + g.guide[e].long()
    t = j.double() / deg[row].double()
    assert bool((start >= lo).all()) and bool((start < g.rowptr[row + 1]).all())
    prev_ok = (start == lo) | (g.cdf[(start - 1).clamp(min=0)] <= t)
    assert bool(prev_ok.all())


@pytest.mark.parametrize("T", [10, 50])                            # BASELINE configs 2 / 3
def test_sampler_properties_full_size(big, T):
    from pinsage_hip import sampling
    g, ei, ew = big
    M, W, L = 59047, 100, 2
    nodes = torch.arange(M, device=g.device)
    a = sampling.walk_sample(g, nodes, T, W, L, rng="philox", seed=42, call=3)
    b = sampling.walk_sample(g, nodes, T, W, L, rng="philox", seed=42, call=3)
    assert torch.equal(a.ids, b.ids) and torch.equal(a.counts, b.counts)            # deterministic
    # batching / shard invariance: any slice of start nodes gives the same rows
    for lo, hi in ((0, 1024), (30000, 37381), (59000, 59047)):
        s = sampling.walk_sample(g, nodes[lo:hi], T, W, L, rng="philox", seed=42, call=3)
        assert torch.equal(s.ids, a.ids[lo:hi]) and torch.equal(s.counts, a.counts[lo:hi])
    nv = a.nvalid.long()
    ar = torch.arange(T, device=g.device)[None, :]
    valid = ar < nv[:, None]
    if T == 10:
        assert bool((nv == T).all())                               # 200 visits over >= 20-degree rows: always >= 10 distinct
    else:
        assert bool((nv >= 10).all()) and bool((nv <= T).all()) and int((nv == T).sum()) > M // 2
    assert bool((a.ids[valid] >= 0).all()) and bool((a.ids[~valid] == -1).all())
    c = a.counts.long()
    assert bool((c[valid] >= 1).all()) and bool((c[~valid] == 0).all()) and bool((c.sum(1) <= W * L).all())
    assert bool((c[:, :-1] >= c[:, 1:]).all())                     # sorted by visit count, descending
    # no duplicate ids inside a row
    srt = torch.sort(a.ids, dim=1).values
    assert bool(((srt[:, 1:] != srt[:, :-1]) | (srt[:, 1:] < 0)).all())
    # visited nodes are within 2 hops: step-1 nodes are users adjacent to the start item
    users_first = a.ids >= M
    i, jx = torch.nonzero(users_first & valid, as_tuple=True)
    pick = torch.randint(0, i.numel(), (20000,), device=g.device)
    si, su = i[pick], a.ids[i[pick], jx[pick]].long()
    # (item si, user su) must be an edge: binary search the item's row for su is not possible (rows keep edge
    # order), so test membership through the user's row length > 0 and the reverse edge count > 0
    lo, hi = g.rowptr[si], g.rowptr[si + 1]
    found = torch.zeros_like(si, dtype=torch.bool)
    for k in range(0, 64):                                          # probe the first 64 entries + hub rows skipped
        idx = (lo + k).clamp(max=g.E - 1)
        found |= (g.col[idx].long() == su) & (lo + k < hi)
    small = (hi - lo) <= 64
    assert bool(found[small].all())
    # different call index -> different samples (fresh draws per layer)
    d = sampling.walk_sample(g, nodes, T, W, L, rng="philox", seed=42, call=4)
    assert not torch.equal(d.ids, a.ids)
    # numpy-stream mode at full size: device MT19937 stream == numpy's, same results for both uniform sources
    np.random.seed(9)
    ref_u = np.random.random_sample(4096 * W * L)
    np.random.seed(9)
    from pinsage_hip import dense
    u = dense.mt19937_random_sample(4096 * W * L, g.device)
    assert np.array_equal(u.cpu().numpy(), ref_u)
    x = sampling.walk_sample(g, nodes[:4096], T, W, L, rng="numpy", uniforms=u)
    y = sampling.walk_sample(g, nodes[:4096], T, W, L, rng="numpy", uniforms=torch.from_numpy(ref_u).to(g.device), use_guide=False)
    assert torch.equal(x.ids, y.ids) and torch.equal(x.counts, y.counts)


@pytest.mark.parametrize("D,nbits", [(256, 512), (128, 256)])      # BASELINE configs 3 / 2
def test_lsh_roundtrip_and_sharded_search_full_size(D, nbits):
    from pinsage_hip import dense
    from utils.nearest_neighbors import lsh_rotation_matrix
    dev = torch.device("cuda")
    M, k = 59047, 11
    g = torch.Generator(device=dev).manual_seed(0)
    emb = torch.nn.functional.normalize(torch.randn(M, D, generator=g, device=dev), dim=1)
    A = torch.from_numpy(lsh_rotation_matrix(D, nbits)).to(dev)
    codes = dense.lsh_encode(emb, A)
    assert codes.shape == (M, nbits // 8)
    # linearity of the projection sign: encode(-x) is the bitwise complement wherever x.a != 0
    neg = dense.lsh_encode(-emb[:4096], A)
    assert int(torch.bitwise_and(neg, codes[:4096]).ne(0).sum()) < 16            # only exact zeros may share bits
    q = torch.arange(0, M, 7, device=dev)[:8192]
    planes = dense.lsh_expand(codes)
    dist, ids = dense.hamming_topk(codes[q], codes, k, planes=planes)            # the fp4-MFMA scan
    dpc, ipc = dense.hamming_topk(codes[q], codes, k, use_mfma=False)            # the popcount scan
    assert torch.equal(dist, dpc) and torch.equal(ids, ipc)
    assert bool((ids[:, 0] == q).all()) and bool((dist[:, 0] == 0).all())        # an item is its own nearest code
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())                              # ascending distances
    same = dist[:, 1:] == dist[:, :-1]
    assert bool((ids[:, 1:][same] > ids[:, :-1][same]).all())                     # ties by ascending id
    # distances are the true Hamming distances of the returned ids
    x = torch.bitwise_xor(codes[q][:, None, :], codes[ids])
    true = torch.zeros_like(dist)
    for b in range(8):
        true += ((x >> b) & 1).sum(dim=2).int()
    assert torch.equal(true, dist)
    # 8 shards + merge == unsharded
    chunk = (M + 7) // 8
    parts = [dense.hamming_topk(codes[q], codes[s:s + chunk].contiguous(), k, id_offset=s,
                                planes=dense.lsh_expand(codes[s:s + chunk].contiguous())) for s in range(0, M, chunk)]
    dm, im = dense.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(dm, dist) and torch.equal(im, ids)
    # the multi-GPU exchange format: every shard's scan writes (ids | distances) straight into its candidate record, the
    # gathered records are merged in place (ps_topk_merge_strided) -- same answer, no repacking
    nq, n = q.numel(), q.numel() * k
    rec = (12 * n + 15) // 16 * 16
    records = torch.zeros((8, rec), dtype=torch.uint8, device=dev)
    for p, s in enumerate(range(0, M, chunk)):
        out = (records[p, 8 * n: 12 * n].view(torch.int32).view(nq, k), records[p, : 8 * n].view(torch.int64).view(nq, k))
        part = codes[s:s + chunk].contiguous()
        dense.hamming_topk(codes[q], part, k, id_offset=s, planes=dense.lsh_expand(part), out=out)
    dr, ir = dense.topk_merge_records(records, nq, k)
    assert torch.equal(dr, dist) and torch.equal(ir, ids)


@pytest.mark.parametrize("T,D,rng", [(10, 256, "philox"), (10, 128, "numpy"), (50, 256, "numpy")])
def test_embeddings_full_size_unit_norm_and_shard_invariance(big, T, D, rng):
    """get_embeddings at BASELINE configs 2 / 3 sizes in both RNG modes: the class API (fused two-layer sampling) equals
    the sharded pipeline on one rank and an explicit layer-by-layer evaluation; in numpy mode the global np.random state
    ends where 2 * M * W * L draws leave it."""
    from pinsage_hip import sampling
    from pinsage_hip.shard import ShardedPinSage
    from utils.random_walk import RandomWalkSampler
    from model.pinsage import PinSage
    g, ei, ew = big
    M = 59047
    torch.manual_seed(2)
    model = PinSage(128, 256, D, 2).to(g.device).eval()
    x = torch.randn(M, 128, device=g.device)
    smp = RandomWalkSampler.from_graph(g, 2, 100, rng=rng, seed=42)
    with torch.no_grad():
        np.random.seed(4)
        e1 = model.get_embeddings(x, smp, T)
        tail = np.random.random_sample()
        smp._calls = 0
        params = {k: v.detach() for k, v in model.state_dict().items()}
        np.random.seed(4)
        e2 = ShardedPinSage(params, 2, smp, M).embed(x, T)
        assert np.random.random_sample() == tail
        smp._calls = 0
        np.random.seed(4)
        lists = [smp.sample_batch(torch.arange(M, device=g.device), T) for _ in range(2)]      # two separate launches
        nb = [sampling.LazyNeighborList(b, "ids") for b in lists]
        wt = [sampling.LazyNeighborList(b, "weights") for b in lists]
        e3 = model(x, sampled_neighbors=nb, importance_weights=wt)
        assert np.random.random_sample() == tail
    if rng == "numpy":
        np.random.seed(4)
        np.random.random_sample(2 * M * 100 * 2)                                    # every item has out-edges here
        assert np.random.random_sample() == tail
    assert torch.equal(e1, e2) and torch.equal(e1, e3)
    np.testing.assert_allclose(e1.norm(dim=1).cpu().numpy(), 1.0, rtol=0, atol=1e-5)
    assert bool(torch.isfinite(e1).all())


def test_config5_shaped_graph_without_bucket_records():
    """BASELINE config 5 (100 M items / 1 B edges) cannot keep the 64-byte bucket records (128 GB); DeviceGraph then
    leaves them out and the walk kernel takes the packed-block search for every step.  The largest graph of that
    shape that fits the test budget (4 M items, 400 K users, 40 M ratings = 80 M directed edges), Philox uniforms as
    config 5 prescribes: the no-bucket path must return exactly what the bucket path returns, for one and for two
    fused layers, and the plain-array path (no packed blocks) too."""
    from pinsage_hip import sampling, synth
    from pinsage_hip.graph import DeviceGraph
    dev = torch.device("cuda")
    M, U, R = 4_000_000, 400_000, 40_000_000
    ei, ew = synth.bipartite_ratings(U, M, R, seed=7, device=dev)
    g = DeviceGraph(ei, ew, buckets=False)
    assert g.buckets is None and g.E == 2 * R
    nodes = torch.randint(0, M, (300_000,), device=dev)
    a = sampling.walk_sample(g, nodes, 10, 100, 2, rng="philox", seed=42, call=0)
    two = sampling.walk_sample_layers(g, nodes, 10, 2, 100, 2, rng="philox", seed=42, call=0)
    assert torch.equal(two[0].ids, a.ids) and torch.equal(two[0].counts, a.counts)
    plain = sampling.walk_sample(g, nodes[:50_000], 10, 100, 2, rng="philox", seed=42, call=0, use_packed=False)
    assert torch.equal(plain.ids, a.ids[:50_000]) and torch.equal(plain.counts, a.counts[:50_000])
    del g
    torch.cuda.empty_cache()
    gb = DeviceGraph(ei, ew, buckets=True)
    assert gb.buckets is not None
    b = sampling.walk_sample(gb, nodes, 10, 100, 2, rng="philox", seed=42, call=0)
    b1 = sampling.walk_sample(gb, nodes, 10, 100, 2, rng="philox", seed=42, call=1)
    assert torch.equal(a.ids, b.ids) and torch.equal(a.counts, b.counts) and torch.equal(a.nvalid, b.nvalid)
    assert torch.equal(two[1].ids, b1.ids) and torch.equal(two[1].counts, b1.counts)
    # the 32-byte half records config 5 itself uses (64 GB at 2 x 10^9 edges): same rows again, one launch and fused layers
    del gb
    torch.cuda.empty_cache()
    gh = DeviceGraph(ei, ew, buckets="half", dest_info=True)  # + the destination records config 5's 110 M nodes switch on by themselves
    assert gh.bucket_bytes == 32 and gh.dest_info is not None and gh.dest_info.numel() == 2 * gh.E
    h = sampling.walk_sample(gh, nodes, 10, 100, 2, rng="philox", seed=42, call=0)
    h2 = sampling.walk_sample_layers(gh, nodes, 10, 2, 100, 2, rng="philox", seed=42, call=0)
    assert torch.equal(a.ids, h.ids) and torch.equal(a.counts, h.counts) and torch.equal(a.nvalid, h.nvalid)
    assert torch.equal(h2[1].ids, b1.ids) and torch.equal(h2[1].counts, b1.counts)
    # items of this graph may be unrated (isolated): they return the empty result on both paths
    # the graph as config 5 keeps it while it steps: plain col / cdf / guide dropped (the packed blocks hold the same values), then
    # restored bit for bit
    col0, cdf0, guide0 = gh.col.clone(), gh.cdf.clone(), gh.guide.clone()
    gh.compact()
    assert gh.col is None and gh.cdf is None and gh.guide is None
    hc = sampling.walk_sample(gh, nodes, 10, 100, 2, rng="philox", seed=42, call=0)
    assert torch.equal(a.ids, hc.ids) and torch.equal(a.counts, hc.counts)
    gh.expand()
    assert torch.equal(gh.col, col0) and torch.equal(gh.cdf, cdf0) and torch.equal(gh.guide, guide0)
    gb = gh
    iso = (gb.rowptr[nodes + 1] - gb.rowptr[nodes]) == 0
    assert bool((a.nvalid[iso] == 0).all()) and bool((a.nvalid[~iso] > 0).all())


def test_whole_step_is_hipgraph_capturable(big):
    """include/pinsage_hip.h promises that every call only enqueues work on the given stream (no allocation, no
    synchronisation, no global state): one whole step -- two-layer sampling, pooling, the four GEMMs, LSH encode + sign
    planes, the three Hamming passes and the merge -- is captured into ONE hipGraph and replayed; every replay must
    return exactly what the eager launches return (Philox mode: the numpy-stream mode hands the RNG state back to the
    host, which is a synchronisation by definition)."""
    from pinsage_hip.shard import ShardedPinSage
    from utils.nearest_neighbors import lsh_rotation_matrix
    from utils.random_walk import RandomWalkSampler
    from model.pinsage import PinSage
    g, ei, ew = big
    M, T, D, nbits, k, nq = 59047, 10, 256, 512, 11, 10000
    dev = g.device
    torch.manual_seed(2)
    model = PinSage(128, 256, D, 2).to(dev).eval()
    params = {n: v.detach().float().contiguous() for n, v in model.state_dict().items()}
    x = torch.randn(M, 128, device=dev)
    A = torch.from_numpy(lsh_rotation_matrix(D, nbits)).to(dev)
    smp = RandomWalkSampler.from_graph(g, 2, 100, rng="philox", seed=42)
    pipe = ShardedPinSage(params, 2, smp, M)

    def step():
        smp._calls = 0                                   # the Philox call index is a launch argument: baked into the graph
        emb = pipe.embed(x, T)
        pipe.build_index(emb, A)
        d, i = pipe.search(emb[:nq], k)
        return emb, d, i

    with torch.no_grad():
        e0, d0, i0 = step()                              # eager (also warms allocator pools and one-time attributes)
        step()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            step()                                       # warm-up on the capture stream
            with torch.cuda.graph(graph, stream=side):
                e1, d1, i1 = step()
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            e1.zero_(); d1.zero_(); i1.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(e1, e0) and torch.equal(d1, d0) and torch.equal(i1, i0)


def test_device_ingest_full_size_equals_host_factorize():
    """SURVEY 8f-3 at ML-25M size: 25 M rating rows -> edge_index on the device (sort-unique + scatter-min first rows)
    equals the host pandas.factorize path (itself pinned to the reference's build_graph by golden G8), and DeviceGraph
    accepts the result."""
    from pinsage_hip.graph import DeviceGraph
    from pinsage_hip.ingest import build_graph_from_ratings, build_graph_from_ratings_device
    g = torch.Generator().manual_seed(3)
    R = 25_000_095
    users = torch.randint(1, 162_542, (R,), generator=g) * 7 + 3            # sparse raw ids, like userId / movieId
    movies = torch.randint(1, 59_048, (R,), generator=g) * 13 + 1
    ratings = torch.randint(1, 11, (R,), generator=g).float() * 0.5
    ei_d, ew_d, mu_d, uu_d = build_graph_from_ratings_device(users, movies, ratings)
    ei_h, ew_h, mu_h, uu_h = build_graph_from_ratings(users.numpy(), movies.numpy(), ratings.numpy())
    assert torch.equal(ei_d.cpu(), ei_h) and torch.equal(ew_d.cpu(), ew_h)
    assert np.array_equal(mu_d.cpu().numpy(), np.asarray(mu_h)) and np.array_equal(uu_d.cpu().numpy(), np.asarray(uu_h))
    dg = DeviceGraph(ei_d, ew_d)
    assert dg.E == 2 * R and dg.V == len(mu_h) + len(uu_h) and not dg.has_reachable_sink


def test_config5_full_scale_shard_sampling_and_oracle_parity():
    """BASELINE config 5 at its real size, as one of its eight ranks sees it: the replicated graph of 100 M items, 10 M users
    and 10^9 ratings (2 x 10^9 directed edges: 66 GB + the 64 GB of 32-byte half bucket records, since the 128 GB of full
    records do not fit), Philox uniforms, item shards of 12.5 M.  Every shard of the catalogue is sampled (both GCN layers
    in one launch) and held to the size-independent properties; the C oracle then replays ~1 100 start nodes -- the
    maximum-degree item rows (1.4 M edges: 21-probe searches), the maximum-degree user rows and random items of every shard
    -- on a host copy of the CSR + CDF, and the pinned python oracle rebuilds the CDF of the heaviest rows from their
    weights (reference utils/random_walk.py:72-79: w / w.sum(), cumsum, searchsorted)."""
    import os
    import time
    from oracle import c_oracle as co
    from oracle import pinsage_oracle as orc
    from pinsage_hip import sampling, synth
    from pinsage_hip.graph import DeviceGraph
    dev = torch.device("cuda")
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info(dev)[0]
    if free < 200e9:
        pytest.skip(f"config 5 needs ~170 GB of HBM while the graph is built; {free / 1e9:.0f} GB free")
    U, M, R, P = 10_000_000, 100_000_000, 1_000_000_000, 8
    T, W, L = 10, 100, 2
    t0 = time.time()
    ei, ew = synth.bipartite_ratings(U, M, R, seed=20240601, device=dev)
    g = DeviceGraph(ei, ew)                                # 128 GB of 64-byte records do not fit: the 32-byte half records (64 GB)
    del ei, ew
    torch.cuda.empty_cache()
    print(f"config 5 graph: V={g.V} E={g.E} max degree {g.max_degree}, {g.nbytes() / 1e9:.1f} GB resident "
          f"({g.bucket_bytes}-byte bucket records), built in {time.time() - t0:.1f} s", flush=True)
    assert g.V == M + U and g.E == 2 * R and g.bucket_bytes == 32 and not g.has_reachable_sink
    assert g.dest_info is not None                         # 880 MB of node records are not cache resident: destination records on (16 GB)
    deg = g.rowptr[1:] - g.rowptr[:-1]
    assert int(deg[:M].min()) >= 1 and g.max_degree == int(deg.max()) > 1_000_000
    chunk = M // P
    arT = torch.arange(T, device=dev)[None, :]
    gen = torch.Generator(device=dev).manual_seed(5)
    picked = []                                           # random start items of every shard for the oracle replay
    for r in range(P):
        lo, hi = r * chunk, (r + 1) * chunk
        two = sampling.walk_sample_layers(g, range(lo, hi), T, 2, W, L, rng="philox", seed=42, call=0)
        if r == 0:                                        # determinism: the same launch again
            again = sampling.walk_sample_layers(g, range(lo, hi), T, 2, W, L, rng="philox", seed=42, call=0)
            for x, y in zip(two, again):
                assert torch.equal(x.ids, y.ids) and torch.equal(x.counts, y.counts) and torch.equal(x.nvalid, y.nvalid)
            del again
        for layer, b in enumerate(two):
            nv = b.nvalid.long()
            valid = arT < nv[:, None]
            assert bool((nv >= 1).all()) and bool((nv <= T).all())           # every item has out-edges
            assert bool(((b.ids >= 0) & (b.ids < g.V))[valid].all()) and bool((b.ids[~valid] == -1).all())
            c = b.counts.long()
            assert bool((c[valid] >= 1).all()) and bool((c[~valid] == 0).all()) and bool((c.sum(1) <= W * L).all())
            assert bool((c[:, :-1] >= c[:, 1:]).all())                       # visit counts descending
            srt = torch.sort(b.ids, dim=1).values
            assert bool(((srt[:, 1:] != srt[:, :-1]) | (srt[:, 1:] < 0)).all())   # no duplicate ids in a row
            del srt, c, valid
            # slice / batching invariance: a sub-range of the shard and a scattered batch give the same rows
            off = int(torch.randint(0, chunk - 4096, (1,), generator=gen, device=dev))
            s = sampling.walk_sample(g, torch.arange(lo + off, lo + off + 4096, device=dev), T, W, L, rng="philox", seed=42,
                                     call=layer)
            assert torch.equal(s.ids, b.ids[off:off + 4096]) and torch.equal(s.counts, b.counts[off:off + 4096])
            idx = torch.randint(0, chunk, (2048,), generator=gen, device=dev)
            s = sampling.walk_sample(g, lo + idx, T, W, L, rng="philox", seed=42, call=layer)
            assert torch.equal(s.ids, b.ids[idx]) and torch.equal(s.counts, b.counts[idx])
        assert not torch.equal(two[0].ids, two[1].ids)                       # fresh draws per layer
        picked.append(lo + torch.randint(0, chunk, (128,), generator=gen, device=dev))
        del two
    # ---- oracle parity on the heaviest rows + random items of every shard ----
    host_gb = (os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")) / 1e9
    if host_gb < 64:
        pytest.skip(f"oracle replay needs a 25 GB host copy of the CSR + CDF; this host has {host_gb:.0f} GB")
    top_items = torch.topk(deg[:M], 32).indices
    top_users = M + torch.topk(deg[M:], 32).indices
    nodes = torch.cat([top_items, top_users] + picked).cpu().numpy()
    # the CDF of the four heaviest rows, rebuilt by the pinned python oracle from the row's weights
    for v in [int(top_items[0]), int(top_items[1]), int(top_users[0]), int(top_users[1])]:
        lo, hi = int(g.rowptr[v]), int(g.rowptr[v + 1])
        w = g.wsorted[lo:hi].cpu().numpy()
        ref = orc.cdf_from_csr(np.array([0, hi - lo], dtype=np.int64), w)
        assert np.array_equal(g.cdf[lo:hi].cpu().numpy(), ref), f"CDF of row {v} (degree {hi - lo})"
    t0 = time.time()
    cg = co.Graph.from_arrays(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.cdf.cpu().numpy())
    print(f"host copy of rowptr / col / cdf: {time.time() - t0:.1f} s", flush=True)
    dn = torch.from_numpy(nodes).to(dev)
    threads = max(1, min(16, co.max_threads()))
    for call in (0, 1):
        got = sampling.walk_sample(g, dn, T, W, L, rng="philox", seed=42, call=call)
        ids, counts, nvalid, _, _, probes = co.walk_sample(cg, nodes, T, L, W, philox=(42, call), threads=threads)
        assert np.array_equal(got.ids.cpu().numpy().astype(np.int64), ids)
        assert np.array_equal(got.counts.cpu().numpy(), counts) and np.array_equal(got.nvalid.cpu().numpy(), nvalid)
    print(f"oracle parity on {nodes.size} start nodes (degrees up to {int(deg[top_items[0]])}), {probes} CDF probes", flush=True)
