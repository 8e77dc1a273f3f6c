#!/usr/bin/env python3
"""Known-byte-count access patterns for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md,
HBM: wide coalesced reads are reported at exactly half, "other access widths are uncalibrated: calibrate on a known byte
count in your own access pattern").  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE in a second
pass) and feed the CSVs to tools/pmc_calibrate_report.py.  Patterns (tables far beyond the 256 MiB Infinity Cache):
  copy   : y.copy_(x), 4 GiB fp32 each way (16 B per lane streaming)
  gather8: 2^27 uniformly random 8-byte reads from a 16 GiB fp64 table  (the sampler's CDF probes / node records)
  row64  : 2^26 uniformly random 64-byte rows from a 16 GiB table        (the sampler's bucket records / packed half blocks)
  row1k  : 2^22 uniformly random 1 KiB rows from a 16 GiB table           (importance pooling's hidden rows)
Each pattern prints its exact useful byte counts; the index streams are separate tensors whose bytes are listed too."""
import json, sys
import torch
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
out = {}
x = torch.empty(1 << 30, dtype=torch.float32, device=dev).normal_(generator=g)
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
out["copy"] = {"launches": 3, "read_bytes": x.numel() * 4, "write_bytes": x.numel() * 4}
del y
tab = x.view(torch.float64)[: (1 << 29)]                       # 4 GiB view is too small for a cache-free gather: use 16 GiB
del x
tab = torch.empty(1 << 31, dtype=torch.float64, device=dev).normal_(generator=g)      # 16 GiB
N8 = 1 << 27
idx = torch.randint(0, tab.numel(), (N8,), device=dev, generator=g)
for _ in range(3):
    r = torch.index_select(tab, 0, idx)
out["gather8"] = {"launches": 3, "gathers": N8, "useful_read_bytes": N8 * 8, "index_bytes": N8 * 8, "write_bytes": N8 * 8}
del r
t64 = tab.view(-1, 8)
N64 = 1 << 26
idx = torch.randint(0, t64.size(0), (N64,), device=dev, generator=g)
for _ in range(3):
    r = torch.index_select(t64, 0, idx)
out["row64"] = {"launches": 3, "gathers": N64, "useful_read_bytes": N64 * 64, "index_bytes": N64 * 8, "write_bytes": N64 * 64}
del r
t1k = tab.view(-1, 128)
N1k = 1 << 22
idx = torch.randint(0, t1k.size(0), (N1k,), device=dev, generator=g)
for _ in range(3):
    r = torch.index_select(t1k, 0, idx)
out["row1k"] = {"launches": 3, "gathers": N1k, "useful_read_bytes": N1k * 1024, "index_bytes": N1k * 8, "write_bytes": N1k * 1024}
torch.cuda.synchronize()
print(json.dumps(out))
