#!/usr/bin/env python3
"""Timeline of gemm_f32_kernel from a library built with -DPS_GEMM_DEBUG=64 (tools/gemm_knockout.sh 64): every wave stamps
s_memtime at the start, around the two barriers of every K step, before the epilogue and at the end, with HW_ID / XCC_ID.
Prints: blocks per CU, how the two co-resident blocks' K steps are phased against each other, and where a block's life goes."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PS_HIP_LIB", os.path.join(ROOT, "tools/ubench/_dbg/libps_gm64.so"))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd"))
import torch
from pinsage_hip import dense, native

lib = native.lib() if hasattr(native, "lib") else ctypes.CDLL(os.environ["PS_HIP_LIB"])
raw = ctypes.CDLL(os.environ["PS_HIP_LIB"])
M, K, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 59047), 256, 256    # 7381 + PS_GEMM_PERSIST=0: the 32 x 256 shard tile
dev = torch.device("cuda")
x = torch.randn(M, K, device=dev); x2 = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) / 16; W2 = torch.randn(N, K, device=dev) / 16; b = torch.randn(N, device=dev)
for _ in range(3):
    dense.linear(x, W, b, x2=x2, W2=W2, relu=True, l2norm=True)
torch.cuda.synchronize()
raw.ps_debug_gemm_trace(None, 1)
dense.linear(x, W, b, x2=x2, W2=W2, relu=True, l2norm=True)
torch.cuda.synchronize()
buf = np.zeros(4096 * 4 * 48, dtype=np.uint64)
assert raw.ps_debug_gemm_trace(buf.ctypes.data_as(ctypes.c_void_p), 0) == 0
t = buf.reshape(4096, 4, 48)[: (M + 63) // 64].astype(np.int64)
hw, xcc = t[:, :, 46], t[:, :, 47]
cu = ((hw >> 8) & 0xff) | (xcc << 8)                      # cu_id, sh_id, se_id + XCD
simd, slot = (hw >> 4) & 3, hw & 15
t0 = np.zeros_like(t[:, :, 0])
for c in np.unique(cu[:, 0]):                             # s_memtime is per XCD: times are relative to the CU's first block
    t0[cu[:, 0] == c] = t[cu[:, 0] == c][:, :, 0].min()
start, epi, end = t[:, :, 0] - t0, t[:, :, 40] - t0, t[:, :, 41] - t0
print(f"blocks {t.shape[0]}, distinct CUs {len(np.unique(cu[:, 0]))}; kernel span {end.max()} ticks")
print("waves of a block sit on SIMDs:", np.unique(simd, axis=0)[:6].tolist(), " slots seen:", np.unique(slot).tolist())
same_cu = (cu == cu[:, :1]).all()
print("all four waves of a block on one CU:", bool(same_cu))
# per block (wave 0): durations
ends = np.concatenate([t[:, 0, 4:34:2], t[:, 0, 40:41]], axis=1)
mf = ends - t[:, 0, 3:35:2]                     # MFMA stream of steps 0..15: stamp(2+2(s+1)) [last: 40] - stamp(3+2s)
st = t[:, 0, 3:35:2] - t[:, 0, 2:34:2]          # barrier, stash, barrier of step s
life = end[:, 0] - start[:, 0]
print(f"block life median {np.median(life):.0f} ticks; prologue {np.median(t[:,0,2]-t[:,0,0]):.0f}; "
      f"MFMA streams {np.median(mf.sum(1)):.0f} ({np.median(mf):.0f} per step); barrier+stash+barrier {np.median(st[:,1:].sum(1)):.0f} "
      f"({np.median(st[:,1:]):.0f} per step); epilogue {np.median(end[:,0]-epi[:,0]):.0f}")
e = t[:, 0, [40, 42, 43, 44, 45, 41]]
print("epilogue pieces (median ticks): bias/relu %d, sums of squares + LDS %d, barrier %d, sqrt + divisions %d, stores %d"
      % tuple(np.median(np.diff(e, axis=1), axis=0)))
first = start[:, 0] < np.median(life) * 0.5
print(f"first-round blocks {first.sum()}: MFMA/step {np.median(mf[first]):.0f}, stash/step {np.median(st[first][:,1:]):.0f}; "
      f"later blocks {(~first).sum()}: MFMA/step {np.median(mf[~first]):.0f}, stash/step {np.median(st[~first][:,1:]):.0f}")
# phase of co-resident blocks: for every CU, blocks alive at the same time; offset of their step starts modulo the step period
offs = []
for c in np.unique(cu[:, 0]):
    ids = np.where(cu[:, 0] == c)[0]
    for i in ids:
        for j in ids:
            if j <= i:
                continue
            lo, hi = max(start[i, 0], start[j, 0]), min(end[i, 0], end[j, 0])
            if hi - lo < 0.5 * min(life[i], life[j]):
                continue
            si, sj = t[i, 0, 3:35:2] - t0[i, 0], t[j, 0, 3:35:2] - t0[j, 0]
            per = np.median(np.diff(si))
            for a in si[(si > lo) & (si < hi)]:
                d = np.abs(sj - a).min() / per
                offs.append(min(d, 1 - d) if d < 1 else 0.5)
offs = np.array(offs)
ov = []
for c in np.unique(cu[:, 0]):                              # share of every block's epilogue that overlaps a co-resident epilogue
    ids = np.where(cu[:, 0] == c)[0]
    for i in ids:
        o = 0
        for j in ids:
            if j != i:
                o += max(0, min(end[i, 0], end[j, 0]) - max(epi[i, 0], epi[j, 0]))
        ov.append(o / max(1, end[i, 0] - epi[i, 0]))
print(f"epilogue overlap with the CU's other resident: mean {np.mean(ov):.2f}")
print(f"co-resident pairs: phase offset of K-step starts (0 = lockstep, 0.5 = alternating): median {np.median(offs):.2f}, "
      f"quartiles {np.percentile(offs, 25):.2f} / {np.percentile(offs, 75):.2f}, n = {len(offs)}")
per_cu = np.array([np.sum(cu[:, 0] == c) for c in np.unique(cu[:, 0])])
print("blocks per CU histogram:", dict(zip(*np.unique(per_cu, return_counts=True))))
np.save(os.path.join(ROOT, "gpurun_out", "gemm_trace.npy"), t)
