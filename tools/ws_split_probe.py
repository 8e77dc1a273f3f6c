#!/usr/bin/env python3
"""ps_walk_sample_layers (two layers) over the first B items of SYN-25M: one wave per node (all rounds) against one wave per
(node, round) -- PS_WALK_SPLIT=0 / 1, interleaved in one process.  usage: ws_split_probe.py [B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd")); sys.path.insert(0, ROOT)
import torch
from pinsage_hip import synth, sampling
from pinsage_hip.graph import DeviceGraph
dev = torch.device("cuda")
U, M, R = synth.ML25M["num_users"], synth.ML25M["num_items"], synth.ML25M["num_ratings"]
ei, ew = synth.bipartite_ratings(U, M, R, seed=20240601, device=dev)
g = DeviceGraph(ei, ew, device=dev); del ei, ew


def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for B in [int(x) for x in sys.argv[1:]] or [3000, 5120, 7381, 10240, 14762, 29524, 59047]:
    best = {"0": 1e9, "1": 1e9}
    ref = None
    for _ in range(3):
        for v in ("0", "1"):
            os.environ["PS_WALK_SPLIT"] = v
            best[v] = min(best[v], timed(lambda: sampling.walk_sample_layers(g, range(0, B), 10, 2, 100, 2, rng="philox", seed=1, call=0)))
            out = sampling.walk_sample_layers(g, range(0, B), 10, 2, 100, 2, rng="philox", seed=1, call=0)
            if ref is None:
                ref = out
            else:
                assert all(torch.equal(x.ids, y.ids) and torch.equal(x.counts, y.counts) and torch.equal(x.nvalid, y.nvalid) for x, y in zip(ref, out))
    print(f"B = {B:6d}: one wave per node {best['0'] * 1e3:7.1f} us, one wave per (node, layer) {best['1'] * 1e3:7.1f} us", flush=True)
