"""The plain-C oracle (oracle/pinsage_oracle.c) against the reference's golden vectors and
against the numpy oracle on seeded random inputs.  CPU only."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import pinsage_oracle as orc
from conftest import bipartite_graph

G1 = {"A": 3, "B": 1, "C": 2, "D": 1, "E": 1}


def _cgraph(g, name):
    ew = g[f"{name}_edge_weights"] if f"{name}_edge_weights" in g.files else None
    return co.Graph(g[f"{name}_edge_index"], ew)


def test_np_sum_bit_exact():
    rs = np.random.RandomState(1)
    for n in [0, 1, 7, 8, 9, 127, 128, 129, 1000, 8192, 8193, 9001, 20000, 81237]:
        a = rs.random_sample(n) * 5 + 0.01
        assert co.np_sum(a) == float(a.sum()), n


def test_csr_cdf_match_numpy_oracle():
    for weights in ("half", "float", None):
        ei, ew = bipartite_graph(40, 30, 600, 3, weights)
        g = co.Graph(ei, ew)
        rowptr, col, w = orc.csr_from_edges(ei, ew)
        cdf = orc.cdf_from_csr(rowptr, w)
        assert np.array_equal(g.rowptr, rowptr) and np.array_equal(g.col, col)
        assert np.array_equal(g.cdf, cdf)                      # fp64 bit-exact


@pytest.mark.parametrize("name", sorted(G1))
def test_sampler_golden(golden, name):
    g = golden
    cg = _cgraph(g, f"g1_{name}")
    rs = np.random.RandomState(int(g[f"g1_{name}_npseed"]))
    for ci in range(G1[name]):
        pre = f"g1_{name}_{ci}_"
        W, L, T = [int(v) for v in g[pre + "WLT"]]
        nodes = g[pre + "nodes"]
        uoff, n = cg.uniform_offsets(nodes, W, L)
        u = rs.random_sample(n)
        for kw in (dict(), dict(uoff=uoff, threads=4)):        # sequential and offset/parallel forms
            ids, counts, nv, wts, used, _ = co.walk_sample(cg, nodes, T, L, W, uniforms=u, **kw)
            assert np.array_equal(ids, g[pre + "ids"])
            assert np.array_equal(nv, g[pre + "nvalid"])
            assert np.array_equal(wts, g[pre + "weights"])
        assert used == 0 or used == n
    assert rs.random_sample() == float(g[f"g1_{name}_tail"])


def test_sampler_sink_golden(golden):
    g = golden
    cg = _cgraph(g, "g1_S")
    u = np.random.RandomState(7).random_sample(300)
    ids, counts, nv, wts, used, _ = co.walk_sample(cg, [0, 1, 2, 3, 4], 4, 3, 20, uniforms=u)
    assert np.array_equal(ids, g["g1_S_ids"]) and np.array_equal(wts, g["g1_S_weights"])
    rs = np.random.RandomState(7)
    rs.random_sample(used)
    assert rs.random_sample() == float(g["g1_S_tail"])


def test_single_walk_golden(golden):
    g = golden
    cg = _cgraph(g, "g6")
    u = np.random.RandomState(11).random_sample(24)
    pos = 0
    for s, ref in zip(g["g6_starts"], g["g6_walks"]):
        walk, pos = co.single_walk(cg, int(s), 4, u, pos)
        assert walk == ref.tolist()


def test_philox_matches_numpy_oracle():
    ei, ew = bipartite_graph(50, 40, 700, 5, "half")
    cg = co.Graph(ei, ew)
    rowptr, col, w = orc.csr_from_edges(ei, ew)
    cdf = orc.cdf_from_csr(rowptr, w)
    nodes = np.arange(50)
    a = co.walk_sample(cg, nodes, 10, 2, 20, philox=(0x1234567890ABCDEF, 3), threads=3)
    b = orc.batch_sample_neighbors(rowptr, col, cdf, nodes, 10, 2, 20, philox=(0x1234567890ABCDEF, 3))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])


def test_pool_and_forward_golden(golden):
    from test_oracle_golden import _counts_from_weights
    g = golden
    counts = _counts_from_weights(g["g2_weights"], g["g2_nvalid"])
    for tag in ("items", "all"):
        out = co.importance_pool(g[f"g2_h_{tag}"], g["g2_ids"], counts, g["g2_nvalid"])
        np.testing.assert_allclose(out, g[f"g2_out_{tag}"], rtol=1e-5, atol=1e-6)
    params = {k[len("g3_param_"):]: g[k] for k in g.files if k.startswith("g3_param_")}
    layers = []
    for li in range(2):
        nv = g[f"g3_l{li}_nvalid"]
        layers.append((g[f"g3_l{li}_ids"], _counts_from_weights(g[f"g3_l{li}_weights"], nv), nv))
    np.testing.assert_allclose(co.pinsage_forward(params, g["g3_x"], layers), g["g3_e_pool"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(co.pinsage_forward(params, g["g3_x"], None), g["g3_e_mlp"], rtol=1e-5, atol=1e-6)


def test_lsh_encode_matches_numpy_fma_emulation():
    rs = np.random.RandomState(2)
    x = rs.standard_normal((300, 48)).astype(np.float32)
    A = orc.lsh_rotation_matrix(48, 128)
    codes, _ = orc.lsh_encode(x, A)
    assert np.array_equal(co.lsh_encode(x, A, threads=2), codes)


def test_hamming_and_dot_topk():
    rs = np.random.RandomState(3)
    codes = rs.randint(0, 256, size=(500, 8)).astype(np.uint8)
    codes[100] = codes[7]; codes[300] = codes[7]              # exact ties -> id order
    d0, i0 = orc.hamming_topk(codes[:20], codes, 11)
    d1, i1 = co.hamming_topk(codes[:20], codes, 11, threads=2)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    assert i1[7, :3].tolist() == [7, 100, 300]
    E = rs.standard_normal((400, 24)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    q = np.array([0, 5, 399])
    v, i = co.dot_topk(E, q, 11)
    for r, qi in enumerate(q):
        _, ref = orc.exact_topk(E, int(qi), 11)
        assert np.array_equal(i[r], ref)
