"""Times the Hamming k-NN scan (int8-MFMA path vs popcount kernel) at a BASELINE-config shape.
usage: python tools/bench_hamming.py [nbits] [nq] [N] [k] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "movie-recommendation-engine_amd")]
import torch

from pinsage_hip import dense

nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 59047
k = int(sys.argv[4]) if len(sys.argv) > 4 else 11
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
g = torch.Generator().manual_seed(0)
codes = torch.randint(0, 256, (N, nbits // 8), generator=g, dtype=torch.uint8).cuda()
if os.environ.get("PS_BENCH_CODES") == "zeros":            # data-dependent power / clock probe (probe libraries only)
    codes.zero_()
q = codes[torch.randperm(N, generator=g)[:nq].cuda()].contiguous()
planes = dense.lsh_expand(codes)


def run(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:      # settle the clock
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


ms_m = run(lambda: dense.hamming_topk(q, codes, k, planes=planes), reps)
ms_v = run(lambda: dense.hamming_topk(q, codes, k, use_mfma=False), reps)
ops = 2.0 * nq * N * nbits
print(f"nbits={nbits} nq={nq} N={N} k={k}: mfma path {ms_m:.4f} ms ({ops / ms_m / 1e9:.1f} Tops/s, incl. query expand + bound + merge), "
      f"popcount {ms_v:.4f} ms; logical code bytes {nq * N * nbits / 8 / ms_m / 1e9:.1f} TB/s")
