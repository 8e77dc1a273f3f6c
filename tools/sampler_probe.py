#!/usr/bin/env python3
"""Sampler-only timing at several graph scales (is ps_walk_sample latency- or traffic-bound?)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scales", default="1.0,0.3,0.1,0.03")
    ap.add_argument("--T", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--hamming", action="store_true")
    a = ap.parse_args()
    from pinsage_hip import synth, sampling, dense
    from pinsage_hip.graph import DeviceGraph
    dev = torch.device("cuda")
    for sc in [float(v) for v in a.scales.split(",")]:
        U, M, R = [int(v * sc) for v in (162541, 59047, 25000095)]
        ei, ew = synth.bipartite_ratings(U, M, R, device=dev)
        g = DeviceGraph(ei, ew); del ei, ew
        nodes = torch.arange(M, device=dev)
        for L, pk, bk in ((1, True, True), (2, True, True), (2, True, False), (2, False, False)):
            fn = lambda: sampling.walk_sample(g, nodes, a.T, 100, L, rng="philox", seed=42, use_packed=pk, use_buckets=bk)
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(a.reps):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
            print(f"scale {sc}: M={M} E={g.E} graph {g.nbytes()/1e6:.0f} MB  L={L} packed={pk} buckets={bk}: {min(ts):.3f} ms -> "
                  f"{min(ts)*1e6/(M*100*L):.2f} ns/step, {M*100*L/min(ts)/1e6:.2f} G steps/s", flush=True)
        del g
    if a.hamming:
        M = 59047
        for nbits, nq in ((512, 10000), (512, 59047), (256, 10000), (256, 59047), (512, 64), (512, 1)):
            codes = torch.randint(0, 256, (M, nbits // 8), dtype=torch.uint8, device=dev)
            fn = lambda: dense.hamming_topk(codes[:nq], codes, 11)
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(a.reps):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
            print(f"hamming nbits={nbits} nq={nq}: {min(ts):.3f} ms -> {nq/min(ts)/1e3:.2f} M q/s", flush=True)

if __name__ == "__main__":
    main()
