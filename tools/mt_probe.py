#!/usr/bin/env python3
"""Device MT19937 (numpy legacy stream) generation rate at the bench size (59047 x 100 x 2 doubles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd"))
import numpy as np, torch
from pinsage_hip import dense
dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 59047 * 200
np.random.seed(0)
st0 = np.random.get_state()
out = dense.mt19937_random_sample(n, dev)
ref = np.random.RandomState(0).random_sample(n)
print("matches numpy:", np.array_equal(out.cpu().numpy(), ref))
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); dense.mt19937_random_sample(n, dev); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(f"n={n}: min {min(ts):.3f} ms -> {n / min(ts) / 1e6:.2f} G doubles/s")
# the walk sampler's form: raw state words, the numpy state handed back asynchronously (what a bench step pays)
ts = []
for _ in range(8):
    np.random.seed(0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); dense.mt19937_random_sample(n, dev, raw=True, advance="defer"); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
dense.finish_rng_state()
print(f"raw stream n={n}: min {min(ts):.3f} ms, median {sorted(ts)[len(ts) // 2]:.3f} ms (event pair around host enqueue + device work)")
# a rank's share of the stream (two layers' runs of 1 / P of the uniforms each): back-to-back calls, no state hand-back
if len(sys.argv) > 2:
    P = int(sys.argv[2])
    h = n // 2
    for rank in (0, P // 2, P - 1):
        runs = [(r * h + rank * (h // P), r * h + (rank + 1) * (h // P)) for r in range(2)]
        for rg in (runs, None):
            for _ in range(3):
                dense.mt19937_random_sample(n, dev, raw=True, advance=False, ranges=rg)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                dense.mt19937_random_sample(n, dev, raw=True, advance=False, ranges=rg)
            b.record(); torch.cuda.synchronize()
            print(f"P={P} rank {rank} {'ranged' if rg else 'whole '}: {a.elapsed_time(b) / 20:.4f} ms per call (20 calls back to back)")
