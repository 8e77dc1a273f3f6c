"""Two processes on one GPU.  The jump product of the device MT19937 generator keeps operand loads in flight by inline asm with
counted waits; a wait that did not own the last requests' registers let late loads land on epilogue registers -- invisible while the
process had the GPU to itself, a memory fault every few steps as soon as a second process' kernels shared the CUs (found by
rehearsing bench.py with two ranks on one GPU).  This test runs the generator (whole stream and a rank's runs) from two
processes at once and holds every call to the first call's words."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path[:0] = [{root!r}, os.path.join({root!r}, "movie-recommendation-engine_amd")]
import numpy as np, torch
from pinsage_hip import dense
n = 23_618_800
h = n // 2
runs = [(r * h + {rank} * (h // 2), r * h + ({rank} + 1) * (h // 2)) for r in range(2)]
ref = None
for it in range(40):
    np.random.seed(42)
    rg = runs if it % 2 else None
    w = dense.mt19937_random_sample(n, "cuda", raw=True, ranges=rg)
    tail = np.random.random_sample()
    if ref is None:
        ref, ref_tail = w.clone(), tail
    assert tail == ref_tail, it
    if rg is None:
        assert torch.equal(w[: 2 * n], ref[: 2 * n]), it
    else:
        for lo, hi in rg:
            assert torch.equal(w[2 * lo: 2 * hi], ref[2 * lo: 2 * hi]), (it, lo, hi)
torch.cuda.synchronize()
print("ok", {rank})
"""


@pytest.mark.timeout(300)
def test_generator_from_two_processes_at_once():
    procs = [subprocess.Popen([sys.executable, "-c", WORKER.format(root=ROOT, rank=r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=280)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"ok {r}" in out and "Memory access fault" not in out, out[-2000:]
