"""WeakANDIndex (inverted file) vs flat L2 at the reference's defaults; run under rocprofv3 --kernel-trace --stats for the kernel split.
python tools/ivf_probe.py [nq]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "movie-recommendation-engine_amd")]
import torch
from utils.nearest_neighbors import WeakANDIndex, _DeviceFlatL2
g = torch.Generator().manual_seed(3)
M, D, k = 59047, 128, 11
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cent = torch.randn(300, D, generator=g)
emb = torch.nn.functional.normalize(cent[torch.randint(0, 300, (M,), generator=g)] + 0.35 * torch.randn(M, D, generator=g), dim=1).cuda()
idx = WeakANDIndex(D); idx.build(emb); idx.index.nprobe = 20
q = emb[torch.randperm(M, generator=g)[:nq].cuda()].contiguous()
flat = _DeviceFlatL2(D); flat.add(emb)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"ivf {timed(lambda: idx.index.search_device(q, k)):.3f} ms   flat {timed(lambda: flat.search_device(q, k)):.3f} ms")
ivf = idx.index
lists = ivf._coarse(q, 20)
print(f"coarse {timed(lambda: ivf._coarse(q, 20)):.3f} ms")
xs, lp, ids, mx = ivf._inverted_lists()
qo = torch.sort(lists[:, 0], stable=True).indices
from pinsage_hip import dense
pr = lists[qo].to(torch.int32); qs = q[qo]
print(f"ivf_topk alone {timed(lambda: dense.ivf_topk(xs, lp, ids, qs, pr, k, max_list=mx)):.3f} ms")
# fraction of 64 x 128 blocks multiplied
import numpy as np
lp_h = lp.cpu().numpy(); pr_h = pr.cpu().numpy()
ntiles = (M + 127) // 128
tl = np.searchsorted(lp_h, np.arange(ntiles) * 128, side="right") - 1
tl1 = np.searchsorted(lp_h, np.minimum(np.arange(ntiles) * 128 + 127, M - 1), side="right") - 1
on = 0
for rt in range(0, nq, 64):
    s = np.zeros(101, dtype=bool); s[np.unique(pr_h[rt:rt + 64])] = True
    cs = np.concatenate([[0], np.cumsum(s)])
    on += int(((cs[tl1 + 1] - cs[tl]) > 0).sum())
print(f"blocks multiplied: {on} of {((nq + 63) // 64) * ntiles} = {on / (((nq + 63) // 64) * ntiles):.3f}; distinct lists per 64-query tile: "
      f"{np.mean([np.unique(pr_h[rt:rt + 64]).size for rt in range(0, nq, 64)]):.1f}")
