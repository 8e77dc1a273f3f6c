#!/usr/bin/env python3
"""Does the shader clock hold under a sustained fp32-MFMA GEMM loop?  Samples rocm-smi while ps_linear runs."""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd"))
import torch
from pinsage_hip import dense

dev = torch.device("cuda")
M = 59047
x = torch.randn(M, 256, device=dev); x2 = torch.randn(M, 256, device=dev)
W = torch.randn(256, 256, device=dev) / 16; W2 = torch.randn(256, 256, device=dev) / 16; b = torch.randn(256, device=dev)
stop = False
samples = []


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            samples.append([l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "mclk" in l])
        except Exception as e:                                  # noqa
            samples.append([repr(e)])
        time.sleep(0.3)


print("idle:", subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout[-900:])
th = threading.Thread(target=poll); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(200):
        dense.linear(x, W, b, x2=x2, W2=W2, relu=True, l2norm=True)
    torch.cuda.synchronize(); n += 200
el = time.time() - t0
stop = True; th.join()
print(f"{n} launches in {el:.2f}s -> {el/n*1e3:.4f} ms each, {2.0*M*256*512*n/el/1e12:.1f} TFLOP/s")
for s in samples[:12]:
    print(s)
