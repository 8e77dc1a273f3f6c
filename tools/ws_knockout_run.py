#!/usr/bin/env python3
"""ps_walk_sample_layers (both layers, SYN-25M) time for the library named by PS_HIP_LIB, Philox and numpy-stream mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from pinsage_hip import synth, sampling, dense
from pinsage_hip.graph import DeviceGraph
dev = torch.device("cuda")
U, M, R = synth.ML25M["num_users"], synth.ML25M["num_items"], synth.ML25M["num_ratings"]
ei, ew = synth.bipartite_ratings(U, M, R, seed=20240601, device=dev)
g = DeviceGraph(ei, ew, device=dev); del ei, ew
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
ph = timed(lambda: sampling.walk_sample_layers(g, range(0, M), 10, 2, 100, 2, rng="philox", seed=1, call=0))
# numpy-stream mode with the uniforms already generated: the walk kernel alone
np.random.seed(1)
raw = dense.mt19937_random_sample(2 * M * 200, dev, advance=False)      # doubles: PS_RNG_STREAM
torch.cuda.synchronize()
try:
    st = timed(lambda: sampling.walk_sample_layers(g, range(0, M), 10, 2, 100, 2, rng="numpy", uniforms=raw))
except Exception as ex:
    st = float("nan"); print("stream mode probe failed:", ex)
print(f"{os.environ.get('PS_HIP_LIB', 'base').split('/')[-1]}: philox {ph:.4f} ms, uniform stream {st:.4f} ms", flush=True)
