#!/usr/bin/env python3
"""Per-rank compute of the item-sharded step WITHOUT communication: one GPU plays rank 0 of `world` ranks (gathers are
replaced by local buffers of the gathered shape).  Shows how far the kernels themselves scale down with the shard size."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd")); sys.path.insert(0, ROOT)
import torch
from pinsage_hip import synth, shard as S, native as nv
from pinsage_hip.graph import DeviceGraph
from utils.random_walk import RandomWalkSampler
from utils.nearest_neighbors import lsh_rotation_matrix
from model.pinsage import PinSage

ap = argparse.ArgumentParser(); ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--graph", action="store_true", help="also time the step replayed from ONE captured hipGraph")
ap.add_argument("--rng", default="philox", choices=["philox", "numpy"])
ap.add_argument("--rank", type=int, default=0, help="which rank's shard this GPU plays (numpy mode: its slice of the RNG stream)")
a = ap.parse_args()
dev = torch.device("cuda")
U, M, R = synth.ML25M["num_users"], synth.ML25M["num_items"], synth.ML25M["num_ratings"]
ei, ew = synth.bipartite_ratings(U, M, R, device=dev)
g = DeviceGraph(ei, ew); del ei, ew
smp = RandomWalkSampler.from_graph(g, 2, 100, rng=a.rng, seed=42)
model = PinSage(128, 256, 256, 2).to(dev).eval()
P = {k: v.detach().float().contiguous() for k, v in model.state_dict().items()}
x_full = torch.randn(M, 128, device=dev)
A = torch.from_numpy(lsh_rotation_matrix(256, 512)).to(dev)
for world in [int(w) for w in a.worlds.split(",")]:
    rank = min(a.rank, world - 1)
    pipe = S.ShardedPinSage(P, 2, smp, M, standin=(rank, world))      # gathers = copies into buffers of the gathered shape
    x_loc = x_full[pipe.lo:pipe.hi].contiguous()
    nq_local = 10000 // world

    def step():
        if a.rng == "numpy":
            import numpy as np
            np.random.seed(42)
        emb = pipe.embed(x_loc, 10, x_full=x_full if world > 1 else None)
        pipe.build_index(emb, A)
        return pipe.search(emb[:nq_local], 11)

    for _ in range(20):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        step()
    host_ms = (time.perf_counter() - t0) / 50 * 1e3          # time to ENQUEUE a step (python + ctypes + allocator)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
    tm = nv.KernelTimer(); nv.set_timer(tm)
    for _ in range(10):
        step()
    ks = tm.summary(); nv.set_timer(None)
    gms = None
    if a.graph:
        with torch.no_grad():
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                smp._calls = 0; step()
                with torch.cuda.graph(gr, stream=side):
                    smp._calls = 0; step()
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(20):
                gr.replay()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50):
                gr.replay()
            torch.cuda.synchronize(); gms = (time.perf_counter() - t0) / 50 * 1e3
    print(f"world {world}: {ms:.3f} ms per step (compute only, rank {rank} of {world}, rng {a.rng}; host enqueue {host_ms:.3f} ms"
          + (f"; replayed from one hipGraph {gms:.3f} ms" if gms is not None else "") + "); "
          + ", ".join(f"{k[3:]} {v['ms'] / 10:.3f}" for k, v in ks.items()), flush=True)
