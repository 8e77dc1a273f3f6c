#!/usr/bin/env python3
"""Per-kernel timing of the hot path on a SYN-25M-shaped graph (HIP events on the launch stream).
Usage: python tools/microbench.py [--scale 1.0] [--T 10] [--reps 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def timed(fn, reps):
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--T", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from pinsage_hip import synth, sampling
    from pinsage_hip.graph import DeviceGraph
    dev = torch.device("cuda")
    U, M, R = [int(v * a.scale) for v in (synth.ML25M["num_users"], synth.ML25M["num_items"], synth.ML25M["num_ratings"])]
    t0 = time.time()
    ei, ew = synth.bipartite_ratings(U, M, R, device=dev)
    torch.cuda.synchronize()
    print(f"synth graph U={U} M={M} R={R}: {time.time()-t0:.2f}s")
    t0 = time.time()
    g = DeviceGraph(ei, ew)
    torch.cuda.synchronize()
    print(f"DeviceGraph build: {time.time()-t0:.3f}s  V={g.V} E={g.E} maxdeg={g.max_degree} bytes={g.nbytes()/1e6:.1f}MB")
    nodes = torch.arange(M, device=dev)
    W, L = 100, 2
    for rng in ("philox", "numpy"):
        if rng == "numpy":
            n = M * W * L
            u = torch.rand(n, dtype=torch.float64, device=dev)
            fn = lambda: sampling.walk_sample(g, nodes, a.T, W, L, rng="numpy", uniforms=u)
        else:
            fn = lambda: sampling.walk_sample(g, nodes, a.T, W, L, rng="philox", seed=42)
        fn()
        mn, med = timed(fn, a.reps)
        print(f"walk_sample[{rng}] T={a.T}: min {mn:.3f} ms  med {med:.3f} ms  -> {M/mn*1e3/1e6:.2f} M nodes/s, "
              f"{M*W*L/mn*1e3/1e9:.2f} G steps/s")
    b = sampling.walk_sample(g, nodes, a.T, W, L, rng="philox", seed=42)
    x = torch.randn(M, 256, device=dev)
    fn = lambda: sampling.importance_pool(x, b)
    fn()
    mn, med = timed(fn, a.reps)
    nv = b.nvalid.sum().item()
    print(f"importance_pool H=256 T={a.T}: min {mn:.3f} ms (sum nvalid={nv}, ~{(nv*1024+M*1024)/mn/1e6:.1f} GB/s incl. user-id drops)")


if __name__ == "__main__":
    main()
