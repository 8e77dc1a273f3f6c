"""The production backend under a real process group: 2 and 3 ranks, started as child processes that
share cuda:0 and talk over gloo, run ShardedPinSage with HipOps -- item-sharded two-layer sampling in BOTH RNG modes (numpy
mode: every rank generates only the MT19937 runs of its own start nodes, buffers poisoned outside them), the per-layer
all-gather of hidden rows (and the replicated-feature layer-0 recompute), LSH build and BOTH search decompositions
(query-sharded over the all-gathered code table; code shards + gathered candidate records + ps_topk_merge_strided), with an
uneven last shard -- and every rank's embeddings, codes and (distance, id) lists must EQUAL the unsharded answer computed
here in the parent.  Functional coverage of pinsage_hip/shard.py:77-105, 356-404 with real collectives; RCCL over xGMI
itself only runs in the driver's multi-GPU bench."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "helpers"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


@pytest.fixture(scope="module")
def unsharded():
    import dist_worker as w
    from pinsage_hip.shard import ShardedPinSage
    from utils.random_walk import RandomWalkSampler
    dev = torch.device("cuda", 0)
    M, graph, params, x, A = w.problem(dev)
    ref = {}
    for tag, rng, replicate, xrep in w.CASES:
        pipe, emb, codes, d, i, tail = w.run_case(ShardedPinSage, RandomWalkSampler, graph, params, x, A, M, rng, replicate, xrep)
        ref[tag] = (emb.cpu().numpy(), codes.cpu().numpy(), d.cpu().numpy(), i.cpu().numpy(), tail)
    # the two decompositions and the two feature paths are the same computation at world 1
    assert np.array_equal(ref["numpy_qshard_xrep"][0], ref["numpy_cshard"][0])
    assert np.array_equal(ref["philox_qshard"][3], ref["philox_cshard_xrep"][3])
    del graph
    torch.cuda.empty_cache()
    return M, ref


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pipeline_under_a_process_group_equals_unsharded(unsharded, world, tmp_path):
    import dist_worker as w
    M, ref = unsharded
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "helpers", "dist_worker.py"), str(r), str(world), port, str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=420)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"ok {r}" in out, out[-3000:]
    got = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    nq_local = w.NQ // world
    for tag, rng, replicate, xrep in w.CASES:
        emb, codes, d, i, tail = ref[tag]
        covered = 0
        for r in range(world):
            lo, hi = [int(v) for v in got[r][tag + "_range"]]
            assert lo == covered and hi > lo
            covered = hi
            assert np.array_equal(got[r][tag + "_emb"], emb[lo:hi]), (tag, r, "embeddings")
            assert np.array_equal(got[r][tag + "_codes"], codes[lo:hi]), (tag, r, "codes")
            if rng == "numpy":
                assert float(got[r][tag + "_tail"]) == tail, (tag, r, "np.random state after the pass")
        assert covered == M and M % world != 0                   # uneven last shard
        # queries: rank r contributes the first nq_local rows of ITS shard; every rank holds all answers, rank-major
        from pinsage_hip.shard import shard_range
        rows = np.concatenate([np.arange(shard_range(M, r, world)[0], shard_range(M, r, world)[0] + nq_local) for r in range(world)])
        # the unsharded reference answered queries 0 .. NQ - 1 of the catalogue; recompute the expectation for THESE rows
        exp_d, exp_i = _search_rows(ref, tag, rows)
        for r in range(world):
            assert np.array_equal(got[r][tag + "_i"], exp_i), (tag, r, "top-k ids")
            assert np.array_equal(got[r][tag + "_d"], exp_d), (tag, r, "top-k distances")


_search_cache = {}


def _search_rows(ref, tag, rows):
    """unsharded top-k of the given catalogue rows' embeddings over the whole code table (HIP, one process)"""
    key = (tag, rows.tobytes())
    if key not in _search_cache:
        import dist_worker as w
        from pinsage_hip import dense
        dev = torch.device("cuda", 0)
        emb, codes = ref[tag][0], ref[tag][1]
        c = torch.from_numpy(codes).to(dev)
        d, i = dense.hamming_topk(c[torch.from_numpy(rows).to(dev)].contiguous(), c, w.K, planes=dense.lsh_expand(c))
        _search_cache[key] = (d.cpu().numpy(), i.cpu().numpy())
    return _search_cache[key]
