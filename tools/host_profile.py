#!/usr/bin/env python3
"""cProfile of the host launch path of one bench step (where do the CPU-side microseconds go?):
python tools/host_profile.py [scale] [philox|numpy] [world]  (world > 1: rank 0's shard through the stand-in communicator)"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from pinsage_hip import synth
from pinsage_hip.graph import DeviceGraph
from pinsage_hip.shard import ShardedPinSage
from utils.random_walk import RandomWalkSampler
from utils.nearest_neighbors import lsh_rotation_matrix
from model.pinsage import PinSage
dev = torch.device("cuda")
sc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
rng = sys.argv[2] if len(sys.argv) > 2 else "philox"
world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
U, M, R = [int(v * sc) for v in (162541, 59047, 25000095)]
ei, ew = synth.bipartite_ratings(U, M, R, device=dev)
g = DeviceGraph(ei, ew)
smp = RandomWalkSampler.from_graph(g, 2, 100, rng=rng, seed=42)
model = PinSage(128, 256, 256, 2).to(dev).eval()
params = {k: v.detach().float().contiguous() for k, v in model.state_dict().items()}
A = torch.from_numpy(lsh_rotation_matrix(256, 512)).to(dev)
pipe = ShardedPinSage(params, 2, smp, M, standin=(0, world) if world > 1 else None)
x_all = torch.randn(M, 128, device=dev)
x = x_all[pipe.lo:pipe.hi].contiguous()
def step():
    if rng == "numpy":
        np.random.seed(42)
    emb = pipe.embed(x, 10, x_full=x_all if world > 1 else None)
    pipe.build_index(emb, A)
    return pipe.search(emb[:1000], 11)
with torch.no_grad():
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): step()
    t_launch = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"host launch time per step: {t_launch/50*1e3:.3f} ms (GPU work small at scale {sc})")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): step()
    pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
