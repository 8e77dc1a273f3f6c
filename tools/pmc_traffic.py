#!/usr/bin/env python3
"""HBM traffic per C-ABI call from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per launch as reported).
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.txt> <out.json> <steps> [label]"""
import csv, glob, json, os, sys
from collections import defaultdict

fetch_dir, write_dir, out_txt, out_json, steps = sys.argv[1:6]
label = sys.argv[6] if len(sys.argv) > 6 else out_txt
steps = int(steps)


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


F, Wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
names = sorted(set(F) | set(Wr), key=lambda k: -(F[k][0] + Wr[k][0]))
lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `bench.py --steps %d --warmup 1 --no-cpu-baseline`" % steps,
         "values: KB per launch (mean over launches), as reported (FETCH_SIZE = TCC_EA0_RDREQ x 64 B; no wide-stream correction applied)", ""]
for k in names:
    own = any(t in k for t in ("walk_", "gemm_f32", "hamming", "importance_pool", "topk_merge", "slice_merge", "bound_select", "lsh_expand", "cdf_", "guide_", "pack_", "bucket_", "gather_kernel", "mt_", "spmm"))
    if not own:
        continue
    nf, nw = F[k][1] or 1, Wr[k][1] or 1
    lines.append(f"{k[:48]:48s} launches={F[k][1]:3d} FETCH_SIZE={F[k][0] / nf:12.1f} KB  WRITE_SIZE={Wr[k][0] / nw:12.1f} KB")
open(out_txt, "w").write("\n".join(lines) + "\n")


def _gemm_epi(k):
    import re
    m = re.match(r"gemm_f32_p?kernel<\s*\d+,\s*\d+,\s*\d+,\s*\d+,\s*\d+,\s*(\d+)", k)
    return int(m.group(1)) if m else -1


def _sampler_kernel(k):
    if not k.startswith("walk_sample_kernel"):
        return False
    stream = any(n.startswith("walk_sample_kernel") and n.rstrip().endswith("true>") for n in names)
    want_stream = stream and "philox" not in label
    return k.rstrip().endswith("true>") == want_stream if stream else True


def per_launch(pred):
    """kernels that one call launches together (scan + merge): per-launch means add up"""
    tot = 0.0
    for k in names:
        if pred(k):
            tot += (F[k][0] / (F[k][1] or 1) + Wr[k][0] / (Wr[k][1] or 1)) * 1024.0
    return tot


def per_call(pred):
    """kernel variants of which a call launches ONE (tile shapes of the GEMM): total bytes / total launches"""
    b = sum(F[k][0] + Wr[k][0] for k in names if pred(k)) * 1024.0
    n = sum(max(F[k][1], Wr[k][1]) for k in names if pred(k))
    return b / max(n, 1)


out = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps {steps} --warmup 1 --no-cpu-baseline ({label})",
       "units": "bytes per launch = (FETCH_SIZE + WRITE_SIZE) KB * 1024, as reported.  Calibration (profiles/r03_config5/pmc_calibration.txt): narrow gathers (8-byte / 64-byte random reads: the sampler) count exactly one 64-byte sector per request = real fabric traffic; wide coalesced streams count at 0.50 and random 1 KiB rows at 0.28 of their bytes (GEMM operands, pooling rows: multiply before comparing with a byte count)",
       # walk_sample_kernel<NP, STREAM>: the headline (numpy-stream) step launches the STREAM = true kernel; the `other_rng_mode` leg of
       # the same bench run launches the Philox one (listed in the .txt, not part of this figure)
       "ps_walk_sample": per_launch(_sampler_kernel),
       "ps_walk_sample_layers": per_launch(_sampler_kernel),
       "ps_hamming_topk_mfma_codes": per_launch(lambda k: k.startswith("hamming_mfma_kernel") or k.startswith("hamming_pipe_kernel") or k.startswith("bound_select_kernel")
                                          or k.startswith("slice_merge_kernel")),
       # every mt_* kernel of the one-round generator runs once per call (begin, planes, jump products, reduce, finish, chunks)
       "ps_mt19937_raw_stream": per_launch(lambda k: k.startswith("mt_") and not k.startswith("mt_raw_to_double")),
       "ps_mt19937_random_sample": per_launch(lambda k: k.startswith("mt_")),
       "ps_importance_pool": per_launch(lambda k: k.startswith("importance_pool_kernel") or k.startswith("importance_pool4_kernel")),
       # template arguments <WM, WN, TM, TN, BK, EPI[, FAST]>: EPI 0 = ps_linear, 1 = ps_lsh_encode (one-tile and persistent kernels)
       "ps_linear": per_call(lambda k: _gemm_epi(k) == 0),
       "ps_lsh_encode": per_call(lambda k: _gemm_epi(k) == 1),
       "ps_hamming_topk": per_launch(lambda k: k.startswith("hamming_scan_kernel") or k.startswith("topk_merge_kernel"))}
json.dump(out, open(out_json, "w"), indent=1)
print(open(out_txt).read())
