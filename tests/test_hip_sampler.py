"""GPU parity of the random-walk sampler: HIP kernels (through the C ABI / the reference-shaped
RandomWalkSampler class) against the golden vectors of the reference and the C oracle.
Bit-exact: CSR, fp64 CDF, neighbour ids, visit counts, fp64 weights, numpy RNG position."""
import numpy as np
import pytest
import torch

from conftest import bipartite_graph

pytestmark = pytest.mark.gpu

G1 = {"A": 3, "B": 1, "C": 2, "D": 1, "E": 1}


def _sampler(g, name, W, L, **kw):
    from utils.random_walk import RandomWalkSampler
    ei = torch.from_numpy(g[f"{name}_edge_index"])
    ew = torch.from_numpy(g[f"{name}_edge_weights"]) if f"{name}_edge_weights" in g.files else None
    return RandomWalkSampler(ei, ew, walk_length=L, num_walks=W, **kw)


@pytest.mark.parametrize("name", sorted(G1))
def test_golden_batch_sample_neighbors(golden, name):
    g = golden
    np.random.seed(int(g[f"g1_{name}_npseed"]))
    for ci in range(G1[name]):
        pre = f"g1_{name}_{ci}_"
        W, L, T = [int(v) for v in g[pre + "WLT"]]
        s = _sampler(g, f"g1_{name}", W, L)
        nodes = g[pre + "nodes"]
        nodes = torch.from_numpy(nodes) if name == "E" else nodes.tolist()
        nb, wt = s.batch_sample_neighbors(nodes, T)
        assert isinstance(nb, list) and isinstance(wt, list) and len(nb) == len(nodes)
        ref_ids, ref_w, ref_nv = g[pre + "ids"], g[pre + "weights"], g[pre + "nvalid"]
        for i in range(len(nodes)):
            k = int(ref_nv[i])
            assert [int(v) for v in nb[i]] == ref_ids[i, :k].tolist()
            assert all(isinstance(v, np.integer) for v in nb[i])
            assert wt[i] == ref_w[i, :k].tolist()                         # fp64 bit-exact
    assert np.random.random_sample() == float(g[f"g1_{name}_tail"])      # global RNG advanced identically


def test_golden_single_walk(golden):
    g = golden
    s = _sampler(g, "g6", 3, 4)
    np.random.seed(11)
    for st, ref in zip(g["g6_starts"], g["g6_walks"]):
        assert [int(v) for v in s._single_walk(int(st))] == ref.tolist()


def test_sink_graph_numpy_mode_matches_the_reference_and_philox_the_oracle(golden):
    """A directed graph with a reachable sink: the reference's walk breaks there before drawing (utils/random_walk.py:65-69), so
    its RNG consumption is data dependent.  rng='numpy' reproduces the reference's own output (golden G1 "S": ids, fp64 weights
    and the np.random position afterwards) through per-walk stream positions found as a fixpoint (sampling.sink_walk_offsets);
    rng='philox' is held to the oracle."""
    from oracle import c_oracle as co
    g = golden
    s = _sampler(g, "g1_S", 20, 3)
    assert s.graph.has_reachable_sink
    np.random.seed(7)
    nb, wt = s.batch_sample_neighbors([0, 1, 2, 3, 4], 4)
    tail = np.random.random_sample()
    for i in range(5):
        k = int(g["g1_S_nvalid"][i])
        assert [int(v) for v in nb[i]] == g["g1_S_ids"][i, :k].tolist() and len(nb[i]) == k
        assert wt[i] == g["g1_S_weights"][i, :k].tolist()                 # fp64 bit-exact
    assert tail == float(g["g1_S_tail"])                                  # the global stream advanced by the uniforms really consumed
    # the same batch node by node (sample_neighbors) and walk by walk (_single_walk) consumes the stream identically
    np.random.seed(7)
    for i in range(5):
        ids_i, w_i = s.sample_neighbors(i, 4)
        k = int(g["g1_S_nvalid"][i])
        assert [int(v) for v in ids_i] == g["g1_S_ids"][i, :k].tolist() and w_i == g["g1_S_weights"][i, :k].tolist()
    assert np.random.random_sample() == float(g["g1_S_tail"])
    cg = co.Graph(g["g1_S_edge_index"], g["g1_S_edge_weights"])
    np.random.seed(3)
    u = np.random.RandomState(3).random_sample(64)
    pos = 0
    for start in (0, 2, 4, 1, 3, 0):
        ref, pos = co.single_walk(cg, start, 3, u, pos)
        assert [int(v) for v in s._single_walk(start)] == ref
    assert np.random.random_sample() == u[pos]
    s = _sampler(g, "g1_S", 20, 3, rng="philox", seed=99)
    b = s.sample_batch([0, 1, 2, 3, 4], 4)
    ids, counts, nv, w, _, _ = co.walk_sample(cg, [0, 1, 2, 3, 4], 4, 3, 20, philox=(99, 0))
    assert np.array_equal(b.ids.cpu().numpy(), ids) and np.array_equal(b.counts.cpu().numpy(), counts)
    assert np.array_equal(b.nvalid.cpu().numpy(), nv)


@pytest.mark.parametrize("L", [2, 3])
def test_sink_graphs_numpy_mode_equals_the_sequential_oracle(L):
    """Larger graphs with sinks against the C oracle's strictly sequential stream consumption (uoff = NULL: exact for any
    graph): (a) a user -> item graph (every item is a sink: the number of steps is decided by the graph, two passes), (b) a
    random directed graph in which a third of the nodes have no out-edges (the number of steps depends on the draws), (c) the
    two-layer get_embeddings launch and an item shard of it.  ids, counts, fp64 weights and the stream position afterwards."""
    from oracle import c_oracle as co
    from pinsage_hip import sampling
    from pinsage_hip.graph import DeviceGraph
    rs = np.random.RandomState(4)
    W, T = 30, 6
    graphs = []
    M, U, R = 300, 200, 6000
    items, users = rs.randint(0, M, R), rs.randint(0, U, R) + M
    graphs.append((np.stack([users, items]).astype(np.int64), (rs.randint(1, 11, R) * 0.5).astype(np.float32), M + U))
    V, E = 400, 5000
    src = rs.randint(0, V, E)
    src = src[src % 3 != 0]                                             # nodes = 0 mod 3 never appear as a source: sinks
    dst = rs.randint(0, V, src.size)
    graphs.append((np.stack([src, dst]).astype(np.int64), (rs.random_sample(src.size) * 3 + 0.25).astype(np.float32), V))
    for ei, ew, V_ in graphs:
        g = DeviceGraph(torch.from_numpy(ei), torch.from_numpy(ew))
        assert g.has_reachable_sink
        cg = co.Graph(ei, ew, num_nodes=g.V)
        nodes = np.concatenate([np.arange(g.V), rs.randint(0, g.V, 100)])
        u = np.random.RandomState(12).random_sample(nodes.size * W * L + 8)
        ids, counts, nv, wts, used, _ = co.walk_sample(cg, nodes, T, L, W, uniforms=u)      # sequential consumption
        assert 0 < used < nodes.size * W * L
        np.random.seed(12)
        b = sampling.walk_sample(g, nodes, T, W, L, rng="numpy")
        assert np.random.random_sample() == u[used]
        hi, hc, hn, hw = b.host()
        assert np.array_equal(hi, ids) and np.array_equal(hc, counts) and np.array_equal(hn, nv)
        valid = np.arange(T)[None, :] < nv[:, None]
        assert np.array_equal(hw[valid], wts[valid])
        # two consecutive calls (get_embeddings' per-layer samples) in one walk_sample_layers call, and a shard of them
        nd = np.arange(min(g.V, 256))
        u2 = np.random.RandomState(5).random_sample(2 * nd.size * W * L + 8)
        r0 = co.walk_sample(cg, nd, T, L, W, uniforms=u2)
        r1 = co.walk_sample(cg, nd, T, L, W, uniforms=u2[r0[4]:])
        np.random.seed(5)
        two = sampling.walk_sample_layers(g, range(nd.size), T, 2, W, L, rng="numpy")
        assert np.random.random_sample() == u2[r0[4] + r1[4]]
        for got, ref in zip(two, (r0, r1)):
            assert np.array_equal(got.ids.cpu().numpy().astype(np.int64), ref[0]) and np.array_equal(got.counts.cpu().numpy(), ref[1])
        np.random.seed(5)
        part = sampling.walk_sample_layers(g, range(100, 180), T, 2, W, L, rng="numpy", stream_nodes=(range(nd.size), 100))
        assert np.random.random_sample() == u2[r0[4] + r1[4]]
        for got, ref in zip(part, (r0, r1)):
            assert np.array_equal(got.ids.cpu().numpy().astype(np.int64), ref[0][100:180])


@pytest.mark.parametrize("weights", ["half", "float", None])
def test_csr_cdf_bit_exact_vs_oracle(weights):
    from oracle import c_oracle as co
    from pinsage_hip.graph import DeviceGraph
    rs = np.random.RandomState(5)
    ei, ew = bipartite_graph(300, 200, 20000, 11, weights)
    # a hub row longer than numpy's 8192-element summation buffer
    hub_u = np.arange(9000) % 200 + 300
    extra = np.stack([np.concatenate([hub_u, np.zeros(9000, dtype=np.int64)]),
                      np.concatenate([np.zeros(9000, dtype=np.int64), hub_u])])
    ei = np.concatenate([ei, extra], axis=1)
    if ew is not None:
        xw = (rs.random_sample(9000) * 4.9 + 0.1).astype(np.float32) if weights == "float" else \
            rs.randint(1, 11, size=9000).astype(np.float32) * 0.5
        ew = np.concatenate([ew, xw, xw])
    g = DeviceGraph(torch.from_numpy(ei), None if ew is None else torch.from_numpy(ew))
    cg = co.Graph(ei, ew, threads=4)
    assert np.array_equal(g.rowptr.cpu().numpy(), cg.rowptr)
    assert np.array_equal(g.col.cpu().numpy(), cg.col)
    assert np.array_equal(g.cdf.cpu().numpy(), cg.cdf)          # fp64 bit-exact incl. the hub row
    assert not g.has_reachable_sink and g.max_degree == int(np.diff(cg.rowptr).max())


@pytest.mark.parametrize("W,L,T", [(100, 2, 10), (100, 2, 50), (64, 1, 5), (33, 5, 7), (200, 3, 20), (1, 1, 1)])
def test_sampler_vs_c_oracle_seeded(W, L, T):
    from oracle import c_oracle as co
    from utils.random_walk import RandomWalkSampler
    ei, ew = bipartite_graph(2000, 1500, 120000, 3, "half")
    cg = co.Graph(ei, ew, threads=4)
    nodes = np.arange(2000)
    uoff, n = cg.uniform_offsets(nodes, W, L)
    rs = np.random.RandomState(42)
    u = rs.random_sample(n)
    ids, counts, nv, w, _, _ = co.walk_sample(cg, nodes, T, L, W, uniforms=u, uoff=uoff, threads=8)
    s = RandomWalkSampler(torch.from_numpy(ei), torch.from_numpy(ew), walk_length=L, num_walks=W)
    np.random.seed(42)
    b = s.sample_batch(nodes, T)
    assert np.array_equal(b.ids.cpu().numpy(), ids)
    assert np.array_equal(b.counts.cpu().numpy(), counts)
    assert np.array_equal(b.nvalid.cpu().numpy(), nv)
    assert np.array_equal(b.host()[3], w)
    assert np.random.random_sample() == rs.random_sample()
    # philox mode, same graph
    s2 = RandomWalkSampler(torch.from_numpy(ei), torch.from_numpy(ew), walk_length=L, num_walks=W, rng="philox", seed=7)
    s2.sample_batch(nodes[:3], T)                                  # call 0
    b2 = s2.sample_batch(nodes, T)                                 # call 1
    ids2, counts2, nv2, _, _, _ = co.walk_sample(cg, nodes, T, L, W, philox=(7, 1), threads=8)
    assert np.array_equal(b2.ids.cpu().numpy(), ids2) and np.array_equal(b2.counts.cpu().numpy(), counts2)


def test_sampler_determinism_and_errors():
    from utils.random_walk import RandomWalkSampler
    ei, ew = bipartite_graph(500, 300, 20000, 9, "half")
    s = RandomWalkSampler(torch.from_numpy(ei), torch.from_numpy(ew), rng="philox", seed=1)
    a = s.sample_batch(np.arange(500), 10)
    s._calls = 0
    b = s.sample_batch(np.arange(500), 10)
    assert torch.equal(a.ids, b.ids) and torch.equal(a.counts, b.counts)
    with pytest.raises(IndexError):
        s.batch_sample_neighbors([0, 800], 10)                     # reference: adj_list[800] IndexError
    nb, wt = s.batch_sample_neighbors([], 10)
    assert len(nb) == 0 and list(nb) == []


def test_importance_pool_golden_and_oracle(golden):
    from oracle import c_oracle as co
    from pinsage_hip import sampling
    from test_oracle_golden import _counts_from_weights
    g = golden
    counts = _counts_from_weights(g["g2_weights"], g["g2_nvalid"])
    dev = torch.device("cuda")
    ids = torch.from_numpy(g["g2_ids"].astype(np.int32)).to(dev)
    cn = torch.from_numpy(counts).to(dev)
    nvl = torch.from_numpy(g["g2_nvalid"]).to(dev)
    for tag in ("items", "all"):
        x = torch.from_numpy(g[f"g2_h_{tag}"]).to(dev)
        out = sampling.importance_pool(x, ids=ids, counts=cn, nvalid=nvl)
        np.testing.assert_allclose(out.cpu().numpy(), g[f"g2_out_{tag}"], rtol=1e-5, atol=1e-6)
    # larger seeded case vs the C oracle, H = 256 (one 1 KiB row per wave instruction) and odd H
    rs = np.random.RandomState(0)
    for H, T in ((256, 50), (100, 50), (7, 50), (64, 150), (256, 64), (32, 65)):   # T > 64: the chunked metadata path
        N, B = 5000, 3000
        x = rs.standard_normal((N, H)).astype(np.float32)
        idn = rs.randint(0, 2 * N, size=(B, T)).astype(np.int64)   # ~half out of range (user ids dropped)
        cnt = rs.randint(1, 20, size=(B, T)).astype(np.int32)
        nvv = rs.randint(0, T + 1, size=B).astype(np.int32)
        ref = co.importance_pool(x, idn, cnt, nvv, threads=8)
        out = sampling.importance_pool(torch.from_numpy(x).to(dev), ids=torch.from_numpy(idn.astype(np.int32)).to(dev),
                                       counts=torch.from_numpy(cnt).to(dev), nvalid=torch.from_numpy(nvv).to(dev))
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


def test_guide_table_is_exact_on_skewed_weights():
    """The bucket-table lookup must return searchsorted's answer for any weights: rows whose weights span
    9 orders of magnitude force long forward scans and the bisection fallback."""
    from oracle import c_oracle as co
    from pinsage_hip import sampling
    from pinsage_hip.graph import DeviceGraph
    rs = np.random.RandomState(4)
    ei, _ = bipartite_graph(400, 300, 60000, 13, None)
    n = ei.shape[1] // 2
    w = np.exp(rs.uniform(np.log(1e-4), np.log(1e5), size=n)).astype(np.float32)
    w[rs.randint(0, n, size=50)] = 1e7                       # a few dominant edges
    ew = np.concatenate([w, w])
    g = DeviceGraph(torch.from_numpy(ei), torch.from_numpy(ew), buckets="full")
    cg = co.Graph(ei, ew, threads=4)
    assert np.array_equal(g.cdf.cpu().numpy(), cg.cdf)
    nodes = np.arange(700)
    a = sampling.walk_sample(g, nodes, 20, 100, 3, rng="philox", seed=5, use_guide=True)
    b = sampling.walk_sample(g, nodes, 20, 100, 3, rng="philox", seed=5, use_guide=False)
    c = sampling.walk_sample(g, nodes, 20, 100, 3, rng="philox", seed=5, use_guide=True, use_packed=False)
    d = sampling.walk_sample(g, nodes, 20, 100, 3, rng="philox", seed=5, use_guide=True, use_buckets=False)
    e = sampling.walk_sample(g, nodes, 20, 100, 3, rng="philox", seed=5, use_packed=False, use_buckets=False)
    assert g.buckets is not None and g.bucket_bytes == 64    # a: LDS-staged start rows + bucket records
    gh = DeviceGraph(torch.from_numpy(ei), torch.from_numpy(ew))                     # the default: 32-byte half records
    assert gh.bucket_bytes == 32 and gh.buckets.numel() == 32 * gh.E
    f = sampling.walk_sample(gh, nodes, 20, 100, 3, rng="philox", seed=5)
    ids, counts, nv, _, _, _ = co.walk_sample(cg, nodes, 20, 3, 100, philox=(5, 0), threads=8)
    for x in (a, b, c, d, e, f):
        assert np.array_equal(x.ids.cpu().numpy(), ids) and np.array_equal(x.counts.cpu().numpy(), counts)
        assert np.array_equal(x.nvalid.cpu().numpy(), nv)


def test_numpy_stream_mode_is_shard_invariant():
    """rng='numpy' under item sharding: every 'rank' draws the whole batch's uniforms and uses the global
    stream offsets of its own slice -> rows identical to the unsharded call, same final np.random state."""
    from pinsage_hip.shard import HipOps
    from utils.random_walk import RandomWalkSampler
    ei, ew = bipartite_graph(3000, 2000, 150000, 17, "half")
    # a few isolated items inside the catalogue (no ratings): they consume no uniforms
    keep = ~np.isin(ei[1][: ei.shape[1] // 2], [5, 1500, 2999])
    keep2 = np.concatenate([keep, keep])
    ei, ew = ei[:, keep2], ew[keep2]
    s = RandomWalkSampler(torch.from_numpy(ei), torch.from_numpy(ew), walk_length=2, num_walks=100)
    M = 3000
    dev = s.graph.device
    np.random.seed(77)
    full = s.sample_batch(torch.arange(M, device=dev), 10)
    tail = np.random.random_sample()
    assert int((full.nvalid == 0).sum()) == 3
    ops = HipOps()
    for world in (2, 3):
        chunk = (M + world - 1) // world
        for r in range(world):
            lo, hi = r * chunk, min((r + 1) * chunk, M)
            np.random.seed(77)
            part = ops.sample(s, torch.arange(lo, hi, device=dev), 10, shard=(M, lo))
            assert torch.equal(part.ids, full.ids[lo:hi]) and torch.equal(part.counts, full.counts[lo:hi])
            assert np.random.random_sample() == tail


def test_error_paths_raise_loudly():
    """Unsupported shapes / bad arguments come back as exceptions, never as silent fallbacks."""
    from pinsage_hip import native, sampling, dense
    from utils.random_walk import RandomWalkSampler
    from utils.nearest_neighbors import LSHIndex
    ei, ew = bipartite_graph(50, 40, 600, 2, "half")
    s = RandomWalkSampler(torch.from_numpy(ei), torch.from_numpy(ew), walk_length=41, num_walks=100, rng="philox")
    with pytest.raises(native.NativeError, match="unsupported"):
        s.sample_batch([0, 1], 5)                                   # W * L = 4100 > 4096 positions per wave
    with pytest.raises(ValueError):
        sampling.walk_sample(s.graph, [0], 5, rng="xorshift")
    with pytest.raises(ValueError):
        RandomWalkSampler(torch.zeros((3, 4), dtype=torch.int64))   # edge_index must be [2, E]
    with pytest.raises(native.NativeError):
        dense.linear(torch.zeros(4, 8), torch.zeros(3, 8))          # host tensors are not device pointers
    with pytest.raises(ValueError):
        dense.linear(torch.zeros(4, 8, device="cuda"), torch.zeros(3, 7, device="cuda"))
    idx = LSHIndex(16, 64, 16)
    idx.build(torch.randn(10, 16))
    with pytest.raises(AssertionError):
        idx.search(torch.randn(2, 15), 3)                           # wrong dimensionality
    d, i = idx.search(torch.randn(2, 16), 100)                      # k > 64 and > ntotal: served (exact path), padded like faiss
    assert d.shape == (2, 100) and np.all(i[:, 10:] == -1) and np.all(i[:, :10] >= 0)
    # C-ABI argument checks of the round-4 entry points (status codes, nothing launched)
    nv = native
    L = nv.lib()
    z = torch.zeros(64, dtype=torch.int32, device="cuda")
    assert L.ps_dest_info_build(nv.ptr(z), nv.ptr(None), nv.i64(8), nv.i64(4), nv.ptr(z), nv.stream()) == nv.PS_EINVAL       # no node records
    assert L.ps_dest_info_build(nv.ptr(z), nv.ptr(z), nv.i64(-1), nv.i64(4), nv.ptr(z), nv.stream()) == nv.PS_EINVAL
    assert L.ps_dest_info_build(nv.ptr(z), nv.ptr(z), nv.i64(0), nv.i64(4), nv.ptr(None), nv.stream()) == nv.PS_OK              # no edges: nothing to do
    q = torch.zeros((64, 64), dtype=torch.uint8, device="cuda")
    o32, o64 = torch.zeros((64, 4), dtype=torch.int32, device="cuda"), torch.zeros((64, 4), dtype=torch.int64, device="cuda")
    big = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    assert L.ps_hamming_topk_mfma_codes(nv.ptr(None), nv.i64(64), nv.ptr(big), nv.i64(4096), nv.i32(64), nv.i32(4), nv.i64(0), nv.ptr(o32),
                                        nv.ptr(o64), nv.ptr(big), nv.C.c_size_t(1 << 20), nv.stream()) == nv.PS_EINVAL            # no queries
    assert L.ps_hamming_topk_mfma_codes(nv.ptr(q), nv.i64(8), nv.ptr(big), nv.i64(4096), nv.i32(64), nv.i32(4), nv.i64(0), nv.ptr(o32),
                                        nv.ptr(o64), nv.ptr(big), nv.C.c_size_t(1 << 20), nv.stream()) == nv.PS_EUNSUPPORTED       # fewer than 64 queries
    assert L.ps_hamming_topk_mfma_codes(nv.ptr(q), nv.i64(64), nv.ptr(big), nv.i64(4096), nv.i32(64), nv.i32(4), nv.i64(0), nv.ptr(o32),
                                        nv.ptr(o64), nv.ptr(big), nv.C.c_size_t(16), nv.stream()) == nv.PS_EWORKSPACE               # workspace too small


def test_hard_negatives_match_reference_semantics():
    """pinsage_hip.negatives.sample_hard_negatives vs a literal restatement of data/negative_sampler.py:44-99 on
    the oracle's `_single_walk` (same global numpy stream): identical indices and final RNG state, for the
    reference's default rank window (always the random fallback: 100 walks visit < 2000 nodes) and a narrow one."""
    from oracle import c_oracle as co
    from pinsage_hip.negatives import sample_hard_negatives
    from utils.random_walk import RandomWalkSampler
    ei, ew = bipartite_graph(300, 200, 9000, 23, "half")
    M = 300
    cg = co.Graph(ei, ew)
    s = RandomWalkSampler(torch.from_numpy(ei), torch.from_numpy(ew), walk_length=2, num_walks=7)
    queries = torch.tensor([5, 17, 299, 0, 123])

    def reference(min_rank, max_rank, num_hard):
        out = []
        for idx in queries.numpy():
            visited = {}
            for _ in range(100):
                u = np.random.random_sample(2)                         # two taken steps per walk on this graph
                walk, _ = co.single_walk(cg, int(idx), 2, u, 0)
                for node in walk[1:]:
                    visited[node] = visited.get(node, 0) + 1
            ranked = sorted(visited.items(), key=lambda x: x[1], reverse=True)
            cand = [item for item, _ in ranked[min_rank:max_rank] if item in range(M)]
            if not cand:
                smp = np.random.choice(list(range(M)), size=num_hard, replace=False)
            else:
                smp = np.random.choice(cand, size=min(num_hard, len(cand)), replace=False)
                if len(smp) < num_hard:
                    add = np.random.choice([i for i in range(M) if i not in smp], size=num_hard - len(smp), replace=False)
                    smp = np.concatenate([smp, add])
            out.append(smp)
        return np.asarray(out)

    for (lo, hi, nh) in ((2000, 5000, 5), (3, 40, 4), (60, 64, 6)):
        np.random.seed(31)
        ref = reference(lo, hi, nh)
        tail = np.random.random_sample()
        np.random.seed(31)
        got = sample_hard_negatives(s, M, queries, num_hard_samples=nh, max_rank=hi, min_rank=lo)
        assert got.shape == (5, nh) and got.dtype == torch.int64
        assert np.array_equal(got.numpy(), ref), (lo, hi)
        assert np.random.random_sample() == tail


def test_bucket_records_match_their_definition():
    """ps_bucket_build: record lo+j = the five CDF entries from guide[lo+j] on and their destinations (past the row
    end: 2.0 / the row's last destination) -- checked field by field against cdf / col / guide."""
    from pinsage_hip.graph import DeviceGraph
    ei, ew = bipartite_graph(60, 40, 900, 3, "half")
    g = DeviceGraph(torch.from_numpy(ei), torch.from_numpy(ew), buckets=True)
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    cdf, guide = g.cdf.cpu().numpy(), g.guide.cpu().numpy()
    raw = g.buckets.cpu().numpy().reshape(g.E, 64)
    c = raw[:, :32].copy().view(np.float64)                              # c0..c3
    k = raw[:, 32:48].copy().view(np.int32)                              # k0..k3
    c4 = raw[:, 48:56].copy().view(np.float64)[:, 0]
    k4 = raw[:, 56:60].copy().view(np.int32)[:, 0]
    cs = np.concatenate([c, c4[:, None]], axis=1)
    ks = np.concatenate([k, k4[:, None]], axis=1)
    for v in range(g.V):
        lo, hi = int(rowptr[v]), int(rowptr[v + 1])
        for e in range(lo, hi):
            first = lo + int(guide[e])
            for i in range(5):
                idx = first + i
                if idx < hi:
                    assert cs[e, i] == cdf[idx] and ks[e, i] == col[idx]
                else:
                    assert cs[e, i] == 2.0 and ks[e, i] == col[hi - 1]


def test_half_bucket_records_match_their_definition_and_the_full_records():
    """ps_bucket_build_half: record lo+j = [fp32 lower bounds of cdf[g] .. cdf[g+3] | col[g] .. col[g+3]] with g the bucket's guide
    position (past the row end: 2.0 / the row's last destination); every fp32 field is the nearest float at or below its fp64
    value.  Walks through half records == walks through full records == plain arrays, in both RNG modes and for fused layers,
    on rating weights and on weights that put many CDF entries into one bucket."""
    from pinsage_hip import sampling
    from pinsage_hip.graph import DeviceGraph
    rs = np.random.RandomState(8)
    for weights in ("half", "skew"):
        ei, ew = bipartite_graph(600, 500, 40000, 3, "half")
        if weights == "skew":
            n = ei.shape[1] // 2
            w = np.exp(rs.uniform(np.log(1e-3), np.log(1e4), size=n)).astype(np.float32)
            ew = np.concatenate([w, w])
        gf = DeviceGraph(torch.from_numpy(ei), torch.from_numpy(ew), buckets="full")
        g = DeviceGraph(torch.from_numpy(ei), torch.from_numpy(ew), buckets="half")
        rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
        cdf, guide = g.cdf.cpu().numpy(), g.guide.cpu().numpy()
        raw = g.buckets.cpu().numpy().reshape(g.E, 32)
        c = raw[:, :16].copy().view(np.float32)
        k = raw[:, 16:32].copy().view(np.int32)
        lo_of = np.repeat(rowptr[:-1], np.diff(rowptr))
        hi_of = np.repeat(rowptr[1:], np.diff(rowptr))
        first = lo_of + guide
        for i in range(4):
            idx = first + i
            inside = idx < hi_of
            want_c = np.where(inside, cdf[np.minimum(idx, g.E - 1)], 2.0)
            want_k = col[np.where(inside, idx, hi_of - 1)]
            assert np.array_equal(k[:, i], want_k)
            assert bool((c[:, i].astype(np.float64) <= want_c).all())
            assert bool((np.nextafter(c[:, i], np.float32(np.inf)).astype(np.float64) > want_c).all())
        nodes = np.arange(g.V)
        for rng in ("philox", "numpy"):
            np.random.seed(2)
            a = sampling.walk_sample(gf, nodes, 10, 100, 2, rng=rng, seed=4, call=1)
            np.random.seed(2)
            b = sampling.walk_sample(g, nodes, 10, 100, 2, rng=rng, seed=4, call=1)
            np.random.seed(2)
            c_ = sampling.walk_sample(g, nodes, 10, 100, 2, rng=rng, seed=4, call=1, use_guide=False)
            for x in (b, c_):
                assert torch.equal(x.ids, a.ids) and torch.equal(x.counts, a.counts) and torch.equal(x.nvalid, a.nvalid)
        two_f = sampling.walk_sample_layers(gf, range(600), 10, 2, 100, 2, rng="philox", seed=4, call=0)
        two_h = sampling.walk_sample_layers(g, range(600), 10, 2, 100, 2, rng="philox", seed=4, call=0)
        for x, y in zip(two_f, two_h):
            assert torch.equal(x.ids, y.ids) and torch.equal(x.counts, y.counts)


def test_destination_records_match_their_definition_and_change_no_walk():
    """ps_dest_info_build: dest_info[e] = nodeinfo[col[e]] = (row start, degree) of the edge's destination.  Walks that take the
    second step's row from these records (staged into LDS with the start row) == walks that gather the node record, in both RNG
    modes, single and fused layers, with full / half / no bucket records, L = 1 / 2 / 3, start rows too long to stage (hubs) and
    graphs with sinks (a destination of degree 0 ends the walk)."""
    from pinsage_hip import sampling
    from pinsage_hip.graph import DeviceGraph
    rs = np.random.RandomState(21)
    ei, ew = bipartite_graph(700, 300, 50000, 5, "half")
    hub = np.stack([np.full(900, 3), 700 + rs.randint(0, 300, size=900)])                # item-side ids start at 700: a 900+-edge row
    ei2 = np.concatenate([ei, hub, hub[::-1]], axis=1)
    ew2 = np.concatenate([ew, np.ones(1800, dtype=np.float32)])
    sink = np.concatenate([ei, np.stack([rs.randint(0, 700, size=300), 1000 + rs.randint(0, 40, size=300)])], axis=1)   # 40 sink nodes
    sw = np.concatenate([ew, np.ones(300, dtype=np.float32)])
    for (e, w, name) in ((ei2, ew2, "hub"), (sink, sw, "sink")):
        for form in ("half", "full", False):
            g0 = DeviceGraph(torch.from_numpy(e), torch.from_numpy(w), buckets=form, dest_info=False)
            g1 = DeviceGraph(torch.from_numpy(e), torch.from_numpy(w), buckets=form, dest_info=True)
            assert g0.dest_info is None and g1.dest_info is not None
            ni = g1.nodeinfo.cpu().numpy().reshape(-1, 2)
            assert np.array_equal(g1.dest_info.cpu().numpy().reshape(-1, 2), ni[g1.col.cpu().numpy()])
            assert g1.has_reachable_sink == (name == "sink")
            nodes = np.arange(g1.V)
            for rng in ("philox", "numpy"):
                for (W, L, T) in ((100, 2, 10), (64, 1, 5), (40, 3, 8)):
                    np.random.seed(3)
                    a = sampling.walk_sample(g0, nodes, T, W, L, rng=rng, seed=9, call=2)
                    sa = np.random.get_state()[1].copy()
                    np.random.seed(3)
                    b = sampling.walk_sample(g1, nodes, T, W, L, rng=rng, seed=9, call=2)
                    assert np.array_equal(sa, np.random.get_state()[1])
                    assert torch.equal(a.ids, b.ids) and torch.equal(a.counts, b.counts) and torch.equal(a.nvalid, b.nvalid), (name, form, rng, W, L)
            if name == "hub":
                for rng in ("philox", "numpy"):
                    np.random.seed(4)
                    two0 = sampling.walk_sample_layers(g0, range(700), 10, 3, 100, 2, rng=rng, seed=4, call=0)
                    np.random.seed(4)
                    two1 = sampling.walk_sample_layers(g1, range(700), 10, 3, 100, 2, rng=rng, seed=4, call=0)
                    for x, y in zip(two0, two1):
                        assert torch.equal(x.ids, y.ids) and torch.equal(x.counts, y.counts) and torch.equal(x.nvalid, y.nvalid)


def test_integration_md_ctypes_stub_reproduces_the_golden(golden):
    """The reference-side ctypes stub printed in INTEGRATION.md (plain C ABI, no accelerator tables) is executed as
    written and must return the reference's neighbours for a golden case."""
    import os, re
    from pinsage_hip import native as nv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = next(b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "def batch_sample" in b)
    code = code.replace('ctypes.CDLL("libpinsage_hip.so")', f'ctypes.CDLL({nv.lib()._name!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    g = golden
    name = "A" if "A" in G1 else sorted(G1)[0]
    np.random.seed(int(g[f"g1_{name}_npseed"]))
    pre = f"g1_{name}_0_"
    W, L, T = [int(v) for v in g[pre + "WLT"]]
    graph = ns["build_graph"](torch.from_numpy(g[f"g1_{name}_edge_index"]), torch.from_numpy(g[f"g1_{name}_edge_weights"]))
    ids, cnt, nvl = ns["batch_sample"](graph, g[pre + "nodes"].tolist(), T, W, L)
    ids, cnt, nvl = ids.cpu().numpy(), cnt.cpu().numpy(), nvl.cpu().numpy()
    assert np.array_equal(nvl, g[pre + "nvalid"])
    for i in range(ids.shape[0]):
        k = int(nvl[i])
        assert ids[i, :k].tolist() == g[pre + "ids"][i, :k].tolist()
        assert (cnt[i, :k] / cnt[i, :k].sum()).tolist() == g[pre + "weights"][i, :k].tolist()


@pytest.mark.parametrize("W,L,T,layers", [(100, 2, 10, 2), (100, 2, 50, 3), (10, 3, 5, 2), (130, 1, 7, 4), (20, 2, 5, 11)])   # 11 layers: two launches
def test_fused_layer_sampling_equals_separate_calls(W, L, T, layers, monkeypatch):
    """ps_walk_sample_layers (all GCN layers' samples of a node in one wave, or -- few start nodes -- one wave per (node, layer);
    model/pinsage.py:271-275 draws them as consecutive batch_sample_neighbors calls) vs `layers` separate launches, in both RNG modes: ids, counts, nvalid
    bit-identical, and in numpy mode the same final np.random state.  Isolated start nodes included.
    PS_MT_POISON=1: the item shard's ranged stream buffer is filled with 0xFFFFFFFF before the generator writes its runs, so
    a walk-kernel read outside the rank's runs cannot pass on stale same-seed words left in the allocator's block."""
    monkeypatch.setenv("PS_MT_POISON", "1")
    from pinsage_hip.shard import HipOps
    from utils.random_walk import RandomWalkSampler
    ei, ew = bipartite_graph(2500, 1800, 120000, 5, "half")
    keep = ~np.isin(ei[1][: ei.shape[1] // 2], [0, 77, 2499])
    keep2 = np.concatenate([keep, keep])
    ei, ew = ei[:, keep2], ew[keep2]
    M = 2500
    for rng in ("philox", "numpy"):
        a = RandomWalkSampler(torch.from_numpy(ei), torch.from_numpy(ew), walk_length=L, num_walks=W, rng=rng, seed=9)
        b = RandomWalkSampler.from_graph(a.graph, walk_length=L, num_walks=W, rng=rng, seed=9)
        np.random.seed(123)
        sep = [a.sample_batch(torch.arange(M, device=a.graph.device), T) for _ in range(layers)]
        tail = np.random.random_sample()
        for trial in range(2):                                   # the second pass answers the stream total from the cache
            # both forms of the launch: one wave per (node, layer) -- what few start nodes get by default -- and one wave per node
            monkeypatch.setenv("PS_WALK_SPLIT", str(trial))
            c = RandomWalkSampler.from_graph(a.graph, walk_length=L, num_walks=W, rng=rng, seed=9)
            np.random.seed(123)
            fused = c.sample_batches(range(M), T, layers)
            assert np.random.random_sample() == tail
            assert len(fused) == layers and c._calls == layers
            for f, s in zip(fused, sep):
                assert torch.equal(f.ids, s.ids) and torch.equal(f.counts, s.counts) and torch.equal(f.nvalid, s.nvalid)
            assert int((fused[0].nvalid == 0).sum()) == 3
        monkeypatch.delenv("PS_WALK_SPLIT")
        # tensor start nodes (not a range) and an item shard of the catalogue
        np.random.seed(123)
        fused = b.sample_batches(torch.arange(M), T, layers)
        assert all(torch.equal(f.ids, s.ids) for f, s in zip(fused, sep))
        lo, hi = 700, 1900
        np.random.seed(123)
        d = RandomWalkSampler.from_graph(a.graph, walk_length=L, num_walks=W, rng=rng, seed=9)   # Philox call counter at 0
        ops = HipOps()
        part = ops.sample_layers(d, lo, hi, T, layers, shard=(M, lo))
        ops.finish()                                             # the deferred np.random state hand-back (numpy mode)
        assert np.random.random_sample() == tail or rng == "philox"
        for f, s in zip(part, sep):
            assert torch.equal(f.ids, s.ids[lo:hi]) and torch.equal(f.counts, s.counts[lo:hi])


def test_hard_negatives_match_the_reference(golden2):
    """G7: NegativeSampler.sample_hard_negatives of the reference (data/negative_sampler.py:44-99) on the reference's
    sampler vs pinsage_hip.negatives on the HIP sampler, rng='numpy': the candidate branch (narrow rank window), the
    fill-up branch (fewer candidates than requested) and the fallback branch (default window) -- same indices, same
    np.random position afterwards."""
    from pinsage_hip.negatives import sample_hard_negatives
    from utils.random_walk import RandomWalkSampler
    g = golden2
    s = RandomWalkSampler(torch.from_numpy(g["g7_edge_index"]), torch.from_numpy(g["g7_edge_weights"]),
                          walk_length=2, num_walks=100)
    q = torch.from_numpy(g["g7_queries"])
    for tag in ("window", "short", "default"):
        nh, mx, mn = [int(v) for v in g[f"g7_{tag}_args"]]
        np.random.seed(31)
        out = sample_hard_negatives(s, int(g["g7_num_movies"]), q, num_hard_samples=nh, max_rank=mx, min_rank=mn)
        assert out.dtype == torch.int64 and np.array_equal(out.numpy(), g[f"g7_{tag}_out"]), tag
        assert np.random.random_sample() == float(g[f"g7_{tag}_tail"]), tag


def test_ppr_helpers_match_the_reference(golden2):
    """G11: compute_ppr_matrix / precompute_top_neighbors (utils/random_walk.py:144-229; dead code upstream but on the
    class surface): same (source, target) keys, scores to 1e-12, same top-neighbour ids and normalised weights."""
    from utils.random_walk import RandomWalkSampler
    g = golden2
    s = RandomWalkSampler(torch.from_numpy(g["g11_edge_index"]), torch.from_numpy(g["g11_edge_weights"]),
                          walk_length=2, num_walks=10)
    nodes = [int(v) for v in g["g11_nodes"]]
    ppr = s.compute_ppr_matrix(nodes, alpha=0.15, num_iterations=4)
    keys = sorted(ppr)
    assert np.array_equal(np.array(keys, dtype=np.int64), g["g11_ppr_keys"])
    np.testing.assert_allclose([ppr[k] for k in keys], g["g11_ppr_vals"], rtol=1e-12, atol=0)
    top = s.precompute_top_neighbors(nodes, num_neighbors=4)
    assert np.array_equal(np.array([top[n][0] for n in nodes]), g["g11_top_ids"])
    np.testing.assert_allclose(np.array([top[n][1] for n in nodes]), g["g11_top_w"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("W,L,T,rng", [(300, 4, 25, "philox"), (1000, 3, 50, "numpy"), (2048, 2, 10, "philox"), (137, 9, 7, "numpy")])
def test_more_than_1024_positions_per_start_node(W, L, T, rng):
    """num_walks * walk_length beyond 1024 (the reference takes any: utils/random_walk.py:98) up to 4096 positions per start
    node: 32 / 64 positions per lane, a larger count bitmap and hash table in (up to 160 KiB of) LDS.  ids / counts vs the C
    oracle, in both RNG modes, one and two fused layers."""
    from oracle import c_oracle as co
    from pinsage_hip import sampling
    from pinsage_hip.graph import DeviceGraph
    ei, ew = bipartite_graph(120, 90, 5000, 31, "half")
    g = DeviceGraph(torch.from_numpy(ei), torch.from_numpy(ew))
    cg = co.Graph(ei, ew)
    nodes = np.arange(0, 210, 3)
    if rng == "philox":
        got = sampling.walk_sample(g, nodes, T, W, L, rng="philox", seed=9, call=2)
        two = sampling.walk_sample_layers(g, nodes, T, 2, W, L, rng="philox", seed=9, call=2)
        ids, counts, nv, _, _, _ = co.walk_sample(cg, nodes, T, L, W, philox=(9, 2), threads=4)
        ids1, counts1, nv1, _, _, _ = co.walk_sample(cg, nodes, T, L, W, philox=(9, 3), threads=4)
        assert np.array_equal(two[1].ids.cpu().numpy(), ids1) and np.array_equal(two[1].counts.cpu().numpy(), counts1)
        assert torch.equal(two[0].ids, got.ids) and torch.equal(two[0].counts, got.counts)
    else:
        uoff, n = cg.uniform_offsets(nodes, W, L)
        u = np.random.RandomState(3).random_sample(n)
        got = sampling.walk_sample(g, nodes, T, W, L, rng="numpy", uniforms=torch.from_numpy(u).cuda())
        ids, counts, nv, _, _, _ = co.walk_sample(cg, nodes, T, L, W, uniforms=u)
    assert np.array_equal(got.ids.cpu().numpy(), ids) and np.array_equal(got.counts.cpu().numpy(), counts)
    assert np.array_equal(got.nvalid.cpu().numpy(), nv)
