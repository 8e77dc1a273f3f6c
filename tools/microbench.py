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
    from pinsage_hip import dense
    for (K, N, K2, relu, l2) in ((128, 256, 0, True, False), (256, 256, 0, False, False), (256, 256, 256, True, True),
                                 (256, 128, 0, False, True), (256, 256, 0, False, True)):
        xx = torch.randn(M, K, device=dev)
        Wt = torch.randn(N, K + K2, device=dev) / 16
        bb = torch.randn(N, device=dev)
        x2 = torch.randn(M, K2, device=dev) if K2 else None
        fn = lambda: dense.linear(xx, Wt[:, :K], bb, x2=x2, W2=Wt[:, K:] if K2 else None, relu=relu, l2norm=l2)
        fn()
        mn, med = timed(fn, a.reps)
        fl = 2.0 * M * N * (K + K2)
        print(f"linear M={M} K={K}+{K2} N={N} relu={relu} l2={l2}: min {mn:.3f} ms -> {fl/mn/1e9:.1f} TFLOP/s")
    for d, nbits in ((128, 256), (256, 512)):
        emb = torch.nn.functional.normalize(torch.randn(M, d, device=dev), dim=1)
        A = torch.randn(nbits, d, device=dev)
        fn = lambda: dense.lsh_encode(emb, A)
        codes = fn()
        mn, med = timed(fn, a.reps)
        print(f"lsh_encode N={M} d={d} nbits={nbits}: min {mn:.3f} ms -> {2.0*M*d*nbits/mn/1e9:.1f} TFLOP/s")
        for nq in (M, 10000):
            fn = lambda: dense.hamming_topk(codes[:nq], codes, 11)
            fn()
            mn, med = timed(fn, a.reps)
            print(f"hamming_topk nq={nq} N={M} nbits={nbits} k=11: min {mn:.3f} ms -> {nq/mn*1e3/1e6:.2f} M queries/s, "
                  f"{nq*M*nbits/8/mn/1e9:.1f} TB/s logical code bytes")
    emb = torch.nn.functional.normalize(torch.randn(M, 128, device=dev), dim=1)
    qi = torch.arange(0, 4096, device=dev)
    fn = lambda: dense.dot_topk(emb, qi, 11)
    fn()
    mn, med = timed(fn, a.reps)
    print(f"dot_topk nq=4096 N={M} D=128 k=11: min {mn:.3f} ms -> {4096/mn*1e3/1e6:.3f} M queries/s")
    n = M * W * L
    np.random.seed(0)
    fn = lambda: dense.mt19937_random_sample(n, dev)
    fn()
    mn, med = timed(fn, 3)
    print(f"mt19937_random_sample n={n}: min {mn:.3f} ms -> {n/mn/1e6:.2f} G doubles/s")
    # end-to-end get_embeddings (2 layers): sampler (philox) + pooled forward
    from utils.random_walk import RandomWalkSampler
    from model.pinsage import PinSage
    smp = RandomWalkSampler.__new__(RandomWalkSampler)
    smp.graph, smp.walk_length, smp.num_walks, smp.rng, smp.seed, smp._calls = g, L, W, "philox", 42, 0
    model = PinSage(128, 256, 128, 2).to(dev).eval()
    feats = torch.randn(M, 128, device=dev)
    with torch.no_grad():
        fn = lambda: model.get_embeddings(feats, smp, a.T)
        fn()
        mn, med = timed(fn, a.reps)
    print(f"get_embeddings(philox) M={M} T={a.T} d=128: min {mn:.3f} ms med {med:.3f} -> {M/mn*1e3/1e6:.2f} M items/s")


if __name__ == "__main__":
    main()
