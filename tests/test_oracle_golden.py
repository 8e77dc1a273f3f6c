"""The CPU oracle against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Bit-exact for ids / counts / fp64 weights; fp32
tensors within 1e-6 (same torch-CPU ops, so in practice identical)."""
import numpy as np
import pytest

from oracle import pinsage_oracle as orc

G1 = {"A": 3, "B": 1, "C": 2, "D": 1, "E": 1}


def _graph(g, name):
    ei = g[f"{name}_edge_index"]
    ew = g[f"{name}_edge_weights"] if f"{name}_edge_weights" in g.files else None
    rowptr, col, w = orc.csr_from_edges(ei, ew)
    return rowptr, col, orc.cdf_from_csr(rowptr, w)


def test_pairwise_sum_matches_numpy():
    rs = np.random.RandomState(0)
    for n in [0, 1, 7, 8, 9, 15, 16, 127, 128, 129, 255, 256, 1000, 4097, 9001, 81237]:
        a = rs.random_sample(n) * 5 + 0.01
        assert orc.np_pairwise_sum(a) == float(a.sum()), n


@pytest.mark.parametrize("name", sorted(G1))
def test_sampler_bit_exact(golden, name):
    g = golden
    rowptr, col, cdf = _graph(g, f"g1_{name}")
    rs = np.random.RandomState(int(g[f"g1_{name}_npseed"]))
    for ci in range(G1[name]):
        pre = f"g1_{name}_{ci}_"
        W, L, T = [int(v) for v in g[pre + "WLT"]]
        nodes = g[pre + "nodes"]
        n = orc.count_uniforms(rowptr, nodes, L, W)
        u = rs.random_sample(n)
        ids, counts, nvalid, weights, used = orc.batch_sample_neighbors(
            rowptr, col, cdf, nodes, T, L, W, uniforms=u)
        assert used == n
        assert np.array_equal(ids, g[pre + "ids"])
        assert np.array_equal(nvalid, g[pre + "nvalid"])
        assert np.array_equal(weights, g[pre + "weights"])        # fp64 bit-exact
    assert rs.random_sample() == float(g[f"g1_{name}_tail"])      # same RNG position


def test_sampler_with_sink(golden):
    g = golden
    rowptr, col, cdf = _graph(g, "g1_S")
    rs = np.random.RandomState(7)
    u = rs.random_sample(5 * 20 * 3)
    ids, counts, nvalid, weights, used = orc.batch_sample_neighbors(
        rowptr, col, cdf, [0, 1, 2, 3, 4], 4, 3, 20, uniforms=u)
    assert np.array_equal(ids, g["g1_S_ids"])
    assert np.array_equal(weights, g["g1_S_weights"])
    assert np.array_equal(nvalid, g["g1_S_nvalid"])
    rs2 = np.random.RandomState(7)
    rs2.random_sample(used)
    assert rs2.random_sample() == float(g["g1_S_tail"])


def test_single_walk(golden):
    g = golden
    rowptr, col, cdf = _graph(g, "g6")
    rs = np.random.RandomState(11)
    src = orc.UniformSource(uniforms=rs.random_sample(6 * 4))
    walks = [orc.single_walk(rowptr, col, cdf, int(s), 4, src) for s in g["g6_starts"]]
    assert np.array_equal(np.array(walks), g["g6_walks"])


def _counts_from_weights(w, nv):
    """Recover integer visit counts from the reference's fp64 weights (count/total,
    total = sum of kept counts <= W*L): smallest total that makes all weights integral."""
    counts = np.zeros(w.shape, dtype=np.int32)
    for i in range(w.shape[0]):
        k = int(nv[i])
        if k == 0:
            continue
        for tot in range(1, 100000):
            c = w[i, :k] * tot
            if np.all(np.abs(c - np.round(c)) < 1e-9) and int(np.round(c).sum()) == tot:
                counts[i, :k] = np.round(c).astype(np.int32)
                break
        else:
            raise AssertionError("no integral total")
    return counts


def test_importance_pool(golden):
    g = golden
    ids, w, nv = g["g2_ids"], g["g2_weights"], g["g2_nvalid"]
    counts = _counts_from_weights(w, nv)
    for tag in ("items", "all"):
        out = orc.importance_pool(g[f"g2_h_{tag}"], ids, counts, nv)
        np.testing.assert_allclose(out, g[f"g2_out_{tag}"], rtol=1e-6, atol=1e-7)


def test_forward_branches(golden):
    g = golden
    params = {k[len("g3_param_"):]: g[k] for k in g.files if k.startswith("g3_param_")}
    layers = []
    for li in range(2):
        ids, w, nv = g[f"g3_l{li}_ids"], g[f"g3_l{li}_weights"], g[f"g3_l{li}_nvalid"]
        layers.append((ids, _counts_from_weights(w, nv), nv))
    e = orc.pinsage_forward(params, g["g3_x"], layers)
    np.testing.assert_allclose(e, g["g3_e_pool"], rtol=1e-5, atol=1e-6)
    e = orc.pinsage_forward(params, g["g3_x"], [layers[0]])      # tuple => shared lists
    np.testing.assert_allclose(e, g["g3_e_shared"], rtol=1e-5, atol=1e-6)
    e = orc.pinsage_forward(params, g["g3_x"], None)
    np.testing.assert_allclose(e, g["g3_e_mlp"], rtol=1e-5, atol=1e-6)


def test_get_embeddings_end_to_end(golden):
    """sampler (fresh draws per layer, model/pinsage.py:271-275) + pooled forward."""
    g = golden
    params = {k[len("g3_param_"):]: g[k] for k in g.files if k.startswith("g3_param_")}
    rowptr, col, cdf = _graph(g, "g3")
    rs = np.random.RandomState(5)
    layers = []
    for _ in range(2):
        n = orc.count_uniforms(rowptr, np.arange(30), 2, 100)
        ids, counts, nv, w, used = orc.batch_sample_neighbors(
            rowptr, col, cdf, np.arange(30), 10, 2, 100, uniforms=rs.random_sample(n))
        layers.append((ids, counts, nv))
    e = orc.pinsage_forward(params, g["g3_x"], layers)
    np.testing.assert_allclose(e, g["g3_e_get"], rtol=1e-5, atol=1e-6)


def test_aggregators(golden):
    g = golden
    ids, w, nv = g["g4_ids"], g["g4_weights"], g["g4_nvalid"]
    np.testing.assert_allclose(orc.mean_aggregate(g["g4_features"], ids, nv), g["g4_mean"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(orc.weighted_aggregate(g["g4_features"], ids, w, nv), g["g4_weighted"],
                               rtol=1e-6, atol=1e-7)


def test_exact_topk(golden):
    g = golden
    for qi, q in enumerate(g["g5_queries"]):
        _, idx = orc.exact_topk(g["g5_emb"], int(q), 11)
        assert np.array_equal(idx, g["g5_top11"][qi])
        _, idx = orc.exact_topk(g["g5_emb"], int(q), 5, exclude_query=False)
        assert np.array_equal(idx, g["g5_top5_incl"][qi])


def test_philox_known_answer():
    # Random123 known-answer vectors for Philox4x32-10
    assert orc.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert orc.philox4x32_10((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert orc.philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


def test_lsh_encode_and_hamming_properties():
    rs = np.random.RandomState(0)
    x = rs.standard_normal((50, 16)).astype(np.float32)
    A = orc.lsh_rotation_matrix(16, 32)
    codes, acc = orc.lsh_encode(x, A)
    assert codes.shape == (50, 4)
    ref = x.astype(np.float64) @ A.astype(np.float64).T
    np.testing.assert_allclose(acc, ref, rtol=1e-5, atol=1e-6)
    clear = np.abs(ref) > 1e-4
    bits = np.unpackbits(codes, axis=1, bitorder="little").astype(bool)
    assert np.array_equal(bits[clear], (ref >= 0)[clear])
    dist, ids = orc.hamming_topk(codes[:5], codes, 7)
    assert np.array_equal(ids[:, 0], np.arange(5)) and np.all(dist[:, 0] == 0)
    assert np.all(np.diff(dist, axis=1) >= 0)
    d2, i2 = orc.hamming_topk(codes[:2], codes[:3], 5)          # ntotal < k padding
    assert np.all(i2[:, 3:] == -1)
