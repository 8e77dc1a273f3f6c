"""Event counts of the MFMA Hamming scan's collect pass (library built with -DPS_HM_DEBUG=32 by tools/hm_probe.sh 32):
python tools/hm_counts.py [nbits] [nq] [N] [k]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PS_HIP_LIB"] = os.path.join(ROOT, "tools", "ubench", "_dbg", "libps_dbg32.so")
sys.path[:0] = [ROOT, os.path.join(ROOT, "movie-recommendation-engine_amd")]
import torch
from pinsage_hip import dense, native
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 59047
k = int(sys.argv[4]) if len(sys.argv) > 4 else 11
g = torch.Generator().manual_seed(0)
codes = torch.randint(0, 256, (N, nbits // 8), generator=g, dtype=torch.uint8).cuda()
q = codes[torch.randperm(N, generator=g)[:nq].cuda()].contiguous()
planes = dense.lsh_expand(codes)
L = native.lib()
buf = (ctypes.c_ulonglong * 8)()
L.ps_debug_hm_counts(buf, 1)
dense.hamming_topk(q, codes, k, planes=planes)
torch.cuda.synchronize()
L.ps_debug_hm_counts(buf, 0)
names = ["tile epilogues (wave)", "slow-path entries (wave)", "group entries (wave)", "row ballots with a hit (wave)", "appended candidates (lane)", "compactions (wave)"]
for n, v in zip(names, buf):
    print(f"{n:34s} {v:12d}  ({v / max(buf[0], 1):.4f} per tile epilogue)")
print(f"candidates per query: {buf[4] / nq:.1f}")
