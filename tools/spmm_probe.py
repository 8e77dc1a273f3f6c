"""GraphConv edge-branch aggregation at ML-25M shape: ps_spmm_csr vs torch index_add_ (both on the GPU)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "movie-recommendation-engine_amd"))
import torch
from pinsage_hip import graph as G, synth

H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = dict(synth.ML25M)
ei, ew = synth.bipartite_ratings(device="cuda", **cfg)
V = cfg["num_users"] + cfg["num_items"]
E = ei.size(1)
x = torch.randn(V, H, device="cuda")
t0 = time.time(); tc = G.TargetCSR(ei, V); torch.cuda.synchronize(); print(f"TargetCSR build {time.time()-t0:.3f}s  V={V} E={E}")
val = ew.float()[tc.perm].contiguous()


def timed(f, n=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


ms = timed(lambda: G.spmm_csr(tc, x, val))
print(f"ps_spmm_csr   {ms:8.3f} ms  gather {E*H*4/ms/1e6:8.1f} GB/s logical, {E/ms/1e6:.2f} Gedge/s")
ms2 = timed(lambda: torch.zeros_like(x).index_add_(0, ei[1], x[ei[0]] * ew.view(-1, 1)), n=2)
print(f"torch idx_add {ms2:8.3f} ms  ({ms2/ms:.1f}x)")
a = G.spmm_csr(tc, x, val); b = torch.zeros_like(x).index_add_(0, ei[1], x[ei[0]] * ew.view(-1, 1))
print("max rel diff", ((a - b).abs().max() / b.abs().max()).item())
