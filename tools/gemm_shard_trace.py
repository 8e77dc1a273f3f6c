#!/usr/bin/env python3
"""Where a workgroup of gemm_shard_kernel spends its life (library built with -DPS_GEMM_DEBUG=64, tools/gemm_knockout.sh 64):
per wave the cycles waiting for the stage + barrier, from the barrier to the step's first fragments (which includes requesting
the next stage), in the MFMA streams, and in the epilogue.  usage: gemm_shard_trace.py [M] [depth]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PS_HIP_LIB", os.path.join(ROOT, "tools/ubench/_dbg/libps_gm64.so"))
os.environ["PS_GEMM_SHARD"] = sys.argv[2] if len(sys.argv) > 2 else "2"
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd"))
import torch
from pinsage_hip import dense

raw = ctypes.CDLL(os.environ["PS_HIP_LIB"])
M, K, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 7381), 256, 256
dev = torch.device("cuda")
x = torch.randn(M, K, device=dev); x2 = torch.randn(M, K, device=dev)
W = dense.stage_weight(torch.randn(N, K, device=dev) / 16); W2 = dense.stage_weight(torch.randn(N, K, device=dev) / 16); b = torch.randn(N, device=dev)
for _ in range(3):
    dense.linear(x, W, b, x2=x2, W2=W2, relu=True, l2norm=True)
torch.cuda.synchronize()
raw.ps_debug_gemm_trace(None, 1)
dense.linear(x, W, b, x2=x2, W2=W2, relu=True, l2norm=True)
torch.cuda.synchronize()
buf = np.zeros(4096 * 4 * 48, dtype=np.uint64)
assert raw.ps_debug_gemm_trace(buf.ctypes.data_as(ctypes.c_void_p), 0) == 0
t = buf.reshape(4096, 4, 48)[: (M + 31) // 32].astype(np.int64)
start, wait, first, stream, epi0, end = t[..., 0], t[..., 1], t[..., 2], t[..., 3], t[..., 40], t[..., 41]
hw, xcc = t[..., 46], t[..., 47]
cu = ((hw >> 8) & 0xff) | (xcc << 8)
steps = 2 * K // 32
print(f"M {M}, depth {os.environ['PS_GEMM_SHARD']}: {t.shape[0]} workgroups on {len(np.unique(cu[:, 0]))} CUs; "
      f"workgroups per CU: {dict(zip(*np.unique(np.unique(cu[:, 0], return_counts=True)[1], return_counts=True)))}")
life = end - start
before = epi0 - start - wait - first - stream
print(f"wave life median {np.median(life):.0f} cycles (max {life.max()}); before the loop {np.median(before):.0f}; per K step: stage wait + barrier "
      f"{np.median(wait) / steps:.0f}, barrier -> first fragments {np.median(first) / steps:.0f}, MFMA stream {np.median(stream) / steps:.0f} "
      f"(32 MFMAs); epilogue {np.median(end - epi0):.0f}")
for x in np.unique(xcc[:, 0]):
    sel = xcc[:, 0] == x
    print(f"  XCD {x}: {sel.sum()} workgroups, first start {start[sel].min() - start.min()}, last end {end[sel].max() - start[sel].min()}")
