#!/usr/bin/env python3
"""ps_linear / ps_lsh_encode at the bench shapes (M = 59047): per-shape time and fp32-MFMA TFLOP/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd"))
import torch
from pinsage_hip import dense

M = int(sys.argv[1]) if len(sys.argv) > 1 else 59047
dev = torch.device("cuda")


def timed(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tot = 0.0
for (K, N, K2, relu, l2) in ((128, 256, 0, True, False), (256, 256, 256, True, True), (256, 256, 256, True, True), (256, 256, 0, False, True)):
    xx = torch.randn(M, K, device=dev); Wt = torch.randn(N, K + K2, device=dev) / 16; bb = torch.randn(N, device=dev)
    x2 = torch.randn(M, K2, device=dev) if K2 else None
    W1, W2 = Wt[:, :K].contiguous(), (Wt[:, K:].contiguous() if K2 else None)
    ms = timed(lambda: dense.linear(xx, W1, bb, x2=x2, W2=W2, relu=relu, l2norm=l2)); tot += ms
    print(f"linear K={K}+{K2} N={N}: {ms:.4f} ms  {2.0*M*N*(K+K2)/ms/1e9:.1f} TFLOP/s")
emb = torch.randn(M, 256, device=dev); A = torch.randn(512, 256, device=dev)
ms = timed(lambda: dense.lsh_encode(emb, A)); tot += ms
print(f"lsh_encode d=256 nbits=512: {ms:.4f} ms  {2.0*M*256*512/ms/1e9:.1f} TFLOP/s")
print(f"total {tot:.4f} ms")
