"""The MFMA Hamming scan (fp4 sign planes, csrc/hamming_mfma.hip: ps_lsh_expand + ps_hamming_topk_mfma / ps_hamming_topk_mfma_codes) vs the C oracle's
restatement of faiss' hammings_knn_hc (reference utils/nearest_neighbors.py:47-68) and vs the popcount kernel:
(distance, id) lists must be bit-identical -- same distances, same ids, same tie order, same padding.

Edge cases the domain has: a ragged last tile (N % 32 != 0), nq % 32 != 0 and nq % 256 != 0, exact duplicates
(distance ties broken by id), a table of identical codes (EVERY distance ties: the lane-private candidate columns
overflow and are compacted over and over), clustered codes, k = 1 / 16 / 17 / 32 (both column capacities), every
served code size, id offsets (shards), 4 to 64 table slices per query block (both merge kernels)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _codes(rs, n, cs, kind):
    if kind == "random":
        return rs.randint(0, 256, size=(n, cs)).astype(np.uint8)
    if kind == "clustered":                                     # 40 centres, a few flipped bits each: many small distances
        cent = rs.randint(0, 256, size=(40, cs)).astype(np.uint8)
        c = cent[rs.randint(0, 40, size=n)].copy()
        flips = rs.randint(0, cs * 8, size=(n, 3))
        for j in range(3):
            c[np.arange(n), flips[:, j] >> 3] ^= (1 << (flips[:, j] & 7)).astype(np.uint8)
        return c
    if kind == "same":
        return np.tile(rs.randint(0, 256, size=(1, cs)).astype(np.uint8), (n, 1))
    raise ValueError(kind)


def test_sign_planes_layout():
    """planes[(tile*KS + s)*1024 + lane*16 ..]: the 32 nibbles (low nibble first) of bits 64 s + 32 (lane >> 5) + j of
    code 32 tile + (lane & 31), +1 = 0x2 / -1 = 0xA in fp4 e2m1; rows past the end are zero nibbles (they contribute
    nothing to a dot product)."""
    from pinsage_hip import dense
    rs = np.random.RandomState(0)
    n, cs = 77, 16
    codes = rs.randint(0, 256, size=(n, cs)).astype(np.uint8)
    pl = dense.lsh_expand(torch.from_numpy(codes).cuda()).cpu().numpy()
    KS, tiles = cs // 8, ((n + 31) // 32 + 3) // 4 * 4            # padded to whole ring entries (2 or 4 tiles)
    assert pl.size == tiles * KS * 1024
    nib = np.stack([pl & 15, pl >> 4], axis=-1).reshape(tiles, KS, 64, 32)          # low nibble first
    bits = np.unpackbits(codes, axis=1, bitorder="little")      # bit j of the code, LSB-first bytes (faiss)
    want = np.zeros((tiles * 32, cs * 8), dtype=np.uint8)
    want[:n] = np.where(bits == 1, 0x2, 0xA)
    want = want.reshape(tiles, 32, KS, 2, 32).transpose(0, 2, 3, 1, 4).reshape(tiles, KS, 64, 32)
    assert np.array_equal(nib, want)
    assert dense.lsh_expand(torch.zeros((8, 4), dtype=torch.uint8).cuda()) is None   # 32-bit codes: popcount scan only


@pytest.mark.parametrize("cs,kind,N,nq,k", [
    (64, "random", 4096 + 17, 300, 11),
    (64, "clustered", 6000, 257, 10),
    (32, "random", 5000, 64, 1),
    (32, "clustered", 4100, 500, 16),
    (16, "random", 8000, 333, 17),
    (8, "random", 4500, 100, 32),
    (4, "random", 4200, 96, 5),
    (64, "same", 4100, 70, 11),
    (32, "same", 5000, 64, 20),
    (64, "clustered", 40000, 100, 11),      # few queries, larger table: 39 slices -> the wave-per-query merge
    (16, "random", 70001, 64, 7),           # 64 slices
])
def test_mfma_scan_matches_oracle_and_popcount(cs, kind, N, nq, k):
    from oracle import c_oracle as co
    from pinsage_hip import dense
    rs = np.random.RandomState(cs * 1000 + N + k)
    codes = _codes(rs, N, cs, kind)
    codes[100] = codes[7]; codes[N - 1] = codes[7]; codes[N - 33] = codes[7]      # duplicates incl. the ragged tail
    q = codes[rs.permutation(N)[:nq]].copy()
    q[0] = codes[7]
    q[1:nq // 2] ^= rs.randint(0, 256, size=(nq // 2 - 1, cs)).astype(np.uint8) & rs.randint(0, 2, size=(nq // 2 - 1, cs)).astype(np.uint8)
    ct, qt = torch.from_numpy(codes).cuda(), torch.from_numpy(q).cuda()
    assert dense.hamming_mfma_supported(nq, N, cs, k) == (cs >= 8)       # 32-bit codes fall back to the popcount scan
    planes = dense.lsh_expand(ct)
    dm, im = dense.hamming_topk(qt, ct, k, planes=planes)
    dv, iv = dense.hamming_topk(qt, ct, k, use_mfma=False)
    assert torch.equal(dm, dv) and torch.equal(im, iv)
    rd, ri = co.hamming_topk(q, codes, k, threads=8)
    assert np.array_equal(im.cpu().numpy(), ri)
    assert np.array_equal(dm.cpu().numpy().astype(np.float32), rd)
    if kind != "same" and k >= 3:
        assert im[0, :3].tolist() == sorted([7, 100, N - 33, N - 1])[:3] and int(dm[0, 0]) == 0
    # shards: id offsets + merge reproduce the single-table answer
    cut = (N // 2) // 32 * 32 + 5
    if cut >= 4096 or N - cut >= 4096:
        parts = []
        for lo, hi in ((0, cut), (cut, N)):
            sub = ct[lo:hi].contiguous()
            parts.append(dense.hamming_topk(qt, sub, k, id_offset=lo, planes=dense.lsh_expand(sub)))
        dmm, imm = dense.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
        assert torch.equal(dmm, dm) and torch.equal(imm, im)


@pytest.mark.parametrize("cs,k", [(32, 11), (64, 11), (64, 5), (32, 12), (16, 11), (64, 20)])
def test_columns_that_fill_up_during_the_sweep(cs, k):
    """The lane-private candidate columns hold k + 16 keys.  Random data never fills one (a lane sees ~3 candidates per
    sweep), so the in-sweep compaction -- bisection for the k-th smallest distance, keep everything up to it, tighten the
    lane's threshold; the serial exact selection when ties fill the column -- needs data built for it: a table whose first
    fifth is far from the queries (loose bound from the sample) and whose rest holds, for a few queries, long runs of
    near-duplicates (a) at pairwise different distances, (b) all at the same distance (ties by id), (c) packed into single
    tiles (many passing rows of one lane in one tile: the walk path with a column that is already nearly full).  Every
    list must equal the popcount kernel's and the C oracle's."""
    from oracle import c_oracle as co
    from pinsage_hip import dense
    rs = np.random.RandomState(cs * 100 + k)
    N, nq, nbits = 16384, 96, cs * 8
    codes = rs.randint(0, 256, size=(N, cs)).astype(np.uint8)
    q = rs.randint(0, 256, size=(nq, cs)).astype(np.uint8)

    def flipped(base, nflip):
        c = base.copy()
        for b in rs.choice(nbits, size=nflip, replace=False):
            c[b >> 3] ^= np.uint8(1 << (b & 7))
        return c

    first = N // 5 + 64                                           # beyond the bound pass's sample
    # (a) query 0: 120 near-duplicates at distances 1..60 (two each), all in rows = 5 mod 32 (one lane's rows) of the later tiles
    rows = first + 5 + 32 * np.arange(120)
    for j, r in enumerate(rows):
        codes[r] = flipped(q[0], 1 + j // 2)
    # (b) query 1: 200 copies at distance exactly 3 (ties: ids decide), rows = 9 mod 32
    rows = first + 9 + 32 * np.arange(200)
    for r in rows:
        codes[r] = flipped(q[1], 3)
    # (c) query 2: whole tiles of near-duplicates (32 consecutive rows, distances 1..8), six tiles apart, 12 of them
    for t in range(12):
        r0 = (first // 32 + 3 + 6 * t) * 32
        for j in range(32):
            codes[r0 + j] = flipped(q[2], 1 + (j % 8))
    # (d) query 3: 40 exact duplicates scattered over every lane
    for r in rs.choice(np.arange(first, N), size=40, replace=False):
        codes[r] = q[3]
    ct, qt = torch.from_numpy(codes).cuda(), torch.from_numpy(q).cuda()
    assert dense.hamming_mfma_supported(nq, N, cs, k)
    dm, im = dense.hamming_topk(qt, ct, k, planes=dense.lsh_expand(ct))
    dv, iv = dense.hamming_topk(qt, ct, k, use_mfma=False)
    assert torch.equal(dm, dv) and torch.equal(im, iv)
    rd, ri = co.hamming_topk(q, codes, k, threads=8)
    assert np.array_equal(im.cpu().numpy(), ri) and np.array_equal(dm.cpu().numpy().astype(np.float32), rd)
    assert int(dm[3, 0]) == 0 and int(dm[1, k - 1]) == 3 and int(dm[0, 0]) == 1


def test_mfma_unsupported_shapes_fall_back():
    """k > 32, small tables and few queries are served by the popcount kernel; the wrapper must say so, not fail"""
    from pinsage_hip import dense
    assert not dense.hamming_mfma_supported(1000, 10000, 64, 33)
    assert not dense.hamming_mfma_supported(1000, 3000, 64, 11)
    assert not dense.hamming_mfma_supported(10, 10000, 64, 11)
    assert not dense.hamming_mfma_supported(1000, 10000, 128, 11)
    rs = np.random.RandomState(3)
    codes = torch.from_numpy(rs.randint(0, 256, size=(5000, 32)).astype(np.uint8)).cuda()
    planes = dense.lsh_expand(codes)
    d1, i1 = dense.hamming_topk(codes[:100], codes, 50, planes=planes)            # k = 50 -> popcount kernel
    d2, i2 = dense.hamming_topk(codes[:100], codes, 50)
    assert torch.equal(d1, d2) and torch.equal(i1, i2)


@pytest.mark.parametrize("nbits,d,k", [(512, 256, 11), (256, 128, 11), (512, 256, 10)])
def test_mfma_scan_full_catalogue_equals_popcount(nbits, d, k):
    """BASELINE configs 2 / 3 sizes: 10 000 queries x 59 047 items through LSHIndex; the MFMA path (default) and the
    popcount kernel return the same bits; self is the nearest item at distance 0; distances ascend; ids are the
    (distance, id) order."""
    from utils.nearest_neighbors import LSHIndex
    g = torch.Generator().manual_seed(nbits + k)
    M, nq = 59047, 10000
    cent = torch.randn(500, d, generator=g)
    emb = torch.nn.functional.normalize(cent[torch.randint(0, 500, (M,), generator=g)] +
                                        0.5 * torch.randn(M, d, generator=g), dim=1).cuda()
    idx = LSHIndex(d, nbits, 16)
    idx.build(emb)
    assert idx.index.planes is not None
    q = emb[torch.randperm(M, generator=g)[:nq].cuda()]
    dm, im = idx.search_device(q, k)
    idx.index.use_mfma = False
    dv, iv = idx.search_device(q, k)
    assert torch.equal(dm, dv) and torch.equal(im, iv)
    assert int((dm[:, 0] != 0).sum()) == 0
    assert bool((dm[:, 1:] >= dm[:, :-1]).all())
    tie = dm[:, 1:] == dm[:, :-1]
    assert bool((im[:, 1:][tie] > im[:, :-1][tie]).all())


def test_mfma_scan_random_shapes_equal_popcount():
    """seeded shape fuzz: the MFMA path and the popcount kernel must agree bit for bit on every served shape -- query counts
    around the 32 / 256 tile boundaries, tables around the 32 / 64-item tile and slice boundaries, every k up to 32, all
    code sizes, heavy ties (few distinct codes) and id offsets"""
    from pinsage_hip import dense
    rs = np.random.RandomState(2024)
    shapes = [(64, 4096, 8, 1), (65, 4097, 64, 32), (255, 4159, 32, 16), (257, 8191, 16, 17), (1000, 12345, 64, 11)]
    for _ in range(9):
        shapes.append((int(rs.randint(64, 1500)), int(rs.randint(4096, 30000)), int(rs.choice([8, 16, 32, 64])),
                       int(rs.randint(1, 33))))
    for nq, N, cs, k in shapes:
        few = rs.rand() < 0.3
        if few:                                                  # ~200 distinct codes: long runs of equal distances
            base = rs.randint(0, 256, size=(200, cs)).astype(np.uint8)
            codes = base[rs.randint(0, 200, size=N)]
        else:
            codes = rs.randint(0, 256, size=(N, cs)).astype(np.uint8)
        q = codes[rs.randint(0, N, size=nq)].copy()
        q[::3] ^= rs.randint(0, 256, size=q[::3].shape).astype(np.uint8) & 1
        ct, qt = torch.from_numpy(codes).cuda(), torch.from_numpy(q).cuda()
        off = int(rs.randint(0, 2 ** 33))
        assert dense.hamming_mfma_supported(nq, N, cs, k)
        dm, im = dense.hamming_topk(qt, ct, k, id_offset=off, planes=dense.lsh_expand(ct))
        dv, iv = dense.hamming_topk(qt, ct, k, id_offset=off, use_mfma=False)
        assert torch.equal(dm, dv) and torch.equal(im, iv), (nq, N, cs, k, few)


@pytest.mark.parametrize("k", [65, 100, 500])
def test_lsh_search_any_k(k):
    """faiss.IndexLSH.search takes any k; beyond the scans' 64 the request runs as an exact L2 search over +-1 vectors
    (squared distance = 4 x Hamming): same (distance, id) order as the C oracle, padding (INT32_MAX, -1) past the table."""
    from oracle import c_oracle as co
    from pinsage_hip import dense
    rs = np.random.RandomState(k)
    N, cs, nq = 400, 32, 19
    codes = _codes(rs, N, cs, "clustered")
    codes[50] = codes[3]; codes[399] = codes[3]
    q = codes[rs.permutation(N)[:nq]].copy()
    q[0] = codes[3]
    d, i = dense.hamming_topk(torch.from_numpy(q).cuda(), torch.from_numpy(codes).cuda(), k, id_offset=1000)
    d, i = d.cpu().numpy(), i.cpu().numpy()
    kk = min(k, N)
    rd, ri = co.hamming_topk(q, codes, kk, threads=4)
    assert np.array_equal(i[:, :kk], ri + 1000) and np.array_equal(d[:, :kk].astype(np.float32), rd)
    if k > N:
        assert np.all(i[:, N:] == -1) and np.all(d[:, N:] == 0x7fffffff)


@pytest.mark.parametrize("cs,N,nq,k", [(64, 9000, 300, 11), (32, 9000, 77 + 64, 10), (64, 5000, 100, 20), (16, 6000, 130, 9)])
def test_query_planes_given_or_built_by_the_scan(cs, N, nq, k, monkeypatch):
    """ps_hamming_topk_mfma (query sign planes made by ps_lsh_expand) and ps_hamming_topk_mfma_codes (packed query codes: the
    pipelined kernels' workgroups expand their own 32 queries in registers -- 512- / 256-bit codes with k <= 12 -- and the launcher
    expands into the workspace for every other kernel: k > 12, 128-bit codes, PS_HAMMING_PIPE=0) return the same lists, ragged
    last query tile included."""
    from pinsage_hip import dense, native as nv
    rs = np.random.RandomState(cs + N + k)
    codes = torch.from_numpy(_codes(rs, N, cs, "clustered")).cuda()
    q = codes[torch.from_numpy(rs.permutation(N)[:nq]).cuda()].contiguous()
    planes = dense.lsh_expand(codes)
    qplanes = dense.lsh_expand(q)
    wsb = int(nv.lib().ps_hamming_topk_mfma_workspace_bytes(nv.i64(nq), nv.i64(N), nv.i32(cs), nv.i32(k)))
    assert wsb > 0
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    d0 = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    i0 = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    nv.call("ps_hamming_topk_mfma", nv.ptr(qplanes), nv.i64(nq), nv.ptr(planes), nv.i64(N), nv.i32(cs), nv.i32(k), nv.i64(5),
            nv.ptr(d0), nv.ptr(i0), nv.ptr(ws), nv.C.c_size_t(wsb), nv.stream())
    for pipe in ("1", "0"):
        monkeypatch.setenv("PS_HAMMING_PIPE", pipe)
        d1, i1 = dense.hamming_topk(q, codes, k, id_offset=5, planes=planes)
        assert torch.equal(d0, d1) and torch.equal(i0, i1), pipe
