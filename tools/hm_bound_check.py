"""How loose is the 20 %-sample bound per query?  python tools/hm_bound_check.py [nbits]  -- candidates (items within the k-th smallest
distance of the first fifth of the table) per query, on the data of tools/bench_hamming.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "movie-recommendation-engine_amd")]
import numpy as np, torch
nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nq, N, k = 10000, 59047, 11
g = torch.Generator().manual_seed(0)
codes = torch.randint(0, 256, (N, nbits // 8), generator=g, dtype=torch.uint8).cuda()
q = codes[torch.randperm(N, generator=g)[:nq].cuda()].contiguous()
pop = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int16, device="cuda")
sample = ((N + 31) // 32 // 5) * 32
cands = []
for b in range(0, nq, 250):
    x = (q[b:b + 250, None, :] ^ codes[None, :, :]).long()
    d = pop[x].sum(-1)                                            # [250, N]
    thr = torch.kthvalue(d[:, :sample], k, dim=1).values
    cands.append((d <= thr[:, None]).sum(1))
c = torch.cat(cands).cpu().numpy()
print(f"candidates per query with the exact k-th distance of the first {sample} items as the bound: mean {c.mean():.1f} median {np.median(c):.0f} "
      f"p99 {np.percentile(c, 99):.0f} max {c.max()}; queries above 200: {(c > 200).sum()}, above 400: {(c > 400).sum()}")
blk = c.reshape(-1, 250)
top = np.argsort(-c)[:10]
print("largest:", [(int(i), int(c[i]), "query block %d" % (i // 256)) for i in top])
