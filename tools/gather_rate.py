#!/usr/bin/env python3
"""Random 64-byte row gathers (the walk sampler's bucket-record access): how many per second does this GPU deliver, by table size?
torch.index_select over a [rows, 16] fp32 table with uniformly random row ids (output written to HBM: 64 B read + 64 B written +
8 B index per gather).  python tools/gather_rate.py"""
import torch
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
n = 1 << 26
for gb in (0.0625, 0.25, 1.6, 4.8, 16.0):
    rows = int(gb * (1 << 30)) // 64
    table = torch.empty((rows, 16), dtype=torch.float32, device=dev).normal_(generator=g)
    idx = torch.randint(0, rows, (n,), device=dev, generator=g)
    out = torch.empty((n, 16), dtype=torch.float32, device=dev)
    for _ in range(2):
        torch.index_select(table, 0, idx, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        torch.index_select(table, 0, idx, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"table {gb:7.3f} GB: {n / ms / 1e6:7.2f} G gathers/s = {n * 64 / ms / 1e9:6.2f} TB/s of 64-byte rows read ({ms:.3f} ms per 2^26)")
    del table, idx, out
