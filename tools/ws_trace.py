#!/usr/bin/env python3
"""Per-wave timeline of the fused walk sampler (library built with -DPS_WS_DEBUG=16): median cycles per phase."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PS_HIP_LIB", os.path.join(ROOT, "tools/ubench/_dbg/libps_ws16.so"))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from pinsage_hip import synth, sampling
from pinsage_hip.graph import DeviceGraph
dev = torch.device("cuda")
rng = "numpy" if "numpy" in sys.argv[1:] else "philox"
big = len(sys.argv) > 1 and sys.argv[1] == "5"            # BASELINE config 5's graph (2 x 10^9 edges), the first 2 M items of rank 0's shard
U, M, R = (10_000_000, 100_000_000, 1_000_000_000) if big else (synth.ML25M["num_users"], synth.ML25M["num_items"], synth.ML25M["num_ratings"])
ei, ew = synth.bipartite_ratings(U, M, R, seed=20240601, device=dev)
g = DeviceGraph(ei, ew, device=dev); del ei, ew
print(f"graph V={g.V} E={g.E} bucket records {g.bucket_bytes} B, destination records {'on' if g.dest_info is not None else 'off'}")
for _ in range(3):
    sampling.walk_sample_layers(g, range(0, 2_000_000 if big else M), 10, 2, 100, 2, rng=rng, seed=1, call=0)
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ["PS_HIP_LIB"])
buf = np.zeros(16384 * 12, dtype=np.uint64)
assert raw.ps_debug_ws_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(16384, 12).astype(np.int64)
t = t[t[:, 0] > 0]
names = ["row bounds + stage start row", "round 0 step 0 (first 128 walks)", "round 0 step 1", "round 1 step 0", "round 1 step 1",
         "(rest of the walk loops)", "count r0", "select r0", "count r1", "select r1"]
d = np.diff(t[:, :11], axis=1)
print(f"rng {rng}: {len(t)} waves; life median {np.median(t[:, 10] - t[:, 0]):.0f} cycles")
for n, col in zip(names, d.T):
    print(f"  {n:36s} median {np.median(col):7.0f}  p90 {np.percentile(col, 90):7.0f}")
