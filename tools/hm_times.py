"""Per-workgroup start / end cycle stamps of the pipelined collect pass (library built with -DPS_HM_DEBUG=512 by tools/hm_probe.sh 512):
python tools/hm_times.py [nbits]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PS_HIP_LIB"] = os.path.join(ROOT, "tools", "ubench", "_dbg", "libps_dbg512.so")
sys.path[:0] = [ROOT, os.path.join(ROOT, "movie-recommendation-engine_amd")]
import numpy as np, torch
from pinsage_hip import dense, native
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nq, N, k = 10000, 59047, 11
g = torch.Generator().manual_seed(0)
codes = torch.randint(0, 256, (N, nbits // 8), generator=g, dtype=torch.uint8).cuda()
q = codes[torch.randperm(N, generator=g)[:nq].cuda()].contiguous()
planes = dense.lsh_expand(codes)
L = native.lib()
for _ in range(20):
    dense.hamming_topk(q, codes, k, planes=planes)
torch.cuda.synchronize()
L.ps_debug_hm_times_clear()                     # the stamps of ONE launch: clear, then one more launch
dense.hamming_topk(q, codes, k, planes=planes)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16384)()
L.ps_debug_hm_times(buf)
t = np.array(buf[:], dtype=np.uint64).reshape(4096, 4)
t = t[t[:, 1] > 0]
start, end = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
hw = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
app = ((t[:, 2] >> np.uint64(32)) & np.uint64(0xffffff)).astype(np.int64)
comp = (t[:, 2] >> np.uint64(56)).astype(np.int64)
xcc = (t[:, 3] & np.uint64(0xf)).astype(np.int64)
slc = ((t[:, 3] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
qb = ((t[:, 3] >> np.uint64(48)) & np.uint64(0xffff)).astype(np.int64)
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7          # HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9)
dur = end - start
print(f"{len(t)} workgroups; duration cycles min {dur.min()} median {int(np.median(dur))} p90 {int(np.percentile(dur, 90))} max {dur.max()}")
for x in sorted(set(xcc)):
    m = xcc == x
    s0 = start[m].min()
    rel_s, rel_e = start[m] - s0, end[m] - s0
    cuid = se[m] * 16 + cu[m]
    per_cu = {}
    for c, a, b, sl, q in zip(cuid, rel_s, rel_e, slc[m], qb[m]):
        per_cu.setdefault(int(c), []).append((int(a), int(b), int(sl), int(q)))
    late = sum(1 for v in per_cu.values() for (a, b, _, _) in v if a > 20000)
    print(f"xcc {x}: {m.sum()} workgroups on {len(per_cu)} CUs, span {rel_e.max()} cycles, started late (> 20 K cycles): {late}; "
          f"workgroups per CU: {sorted(len(v) for v in per_cu.values())}")
    if x == sorted(set(xcc))[0]:
        for c in sorted(per_cu)[:12]:
            print("   cu", c, sorted(per_cu[c]))
order = np.argsort(-dur)[:24]
print("longest workgroups: (duration, xcc, se, cu, slice, qb, wave-0 appends, wave-0 in-sweep compactions)")
for i in order:
    print("  ", int(dur[i]), int(xcc[i]), int(se[i]), int(cu[i]), int(slc[i]), int(qb[i]), int(app[i]), int(comp[i]))
print("mean duration by slice:", [int(dur[slc == s_].mean()) for s_ in sorted(set(slc))])
print("mean duration by query block (first 40):", [int(dur[qb == q_].mean()) for q_ in sorted(set(qb))][:40])
print("wave-0 appends: median", int(np.median(app)), "max", int(app.max()), "; workgroups with in-sweep compactions in wave 0:", int((comp > 0).sum()))
