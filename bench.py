#!/usr/bin/env python3
"""bench.py -- the hot path's headline benchmark on MI355X.

One "step" = one pass of the PinSage hot path over the whole synthetic ML-25M-shaped catalogue:
  embed   : per GCN layer, random-walk neighbour sampling (W=100, L=2, top-T) of every item, then the
            importance-pooled forward (fp32)                          -> [M, d] unit-norm embeddings
  index   : LSH encode of all M embeddings (nbits = 2d)               -> [M, nbits/8] codes
  query   : top-K (K=11) Hamming search of `--queries` item embeddings over all M codes
with the graph (CSR + fp64 CDF), features and weights resident in HBM.  `value` = M items / step time.
N > 1: the item catalogue is sharded by id range over N ranks (RCCL all-gather of hidden rows per layer
and of the [nq, K] candidates), total work fixed -> "scaling": "strong".

Launch: python bench.py --gpus 1 --steps K --warmup W
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.  At N = 1 the line carries `cpu_baseline` (the C oracle + torch CPU on the host cores, the
whole step) and `parity_check`: that CPU run's outputs compared with one GPU step from the same RNG state -- sampled ids /
counts of both layers in both RNG modes, the np.random state afterwards, LSH codes and top-k lists bit-exact, embeddings
within 1e-5; a mismatch exits with status 1 after printing the line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "movie-recommendation-engine_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK = 8.0e12          # B/s, spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK = 157.3e12   # FLOP/s, v_mfma_f32_32x32x2_f32
# v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 operands: 4 x the bf16 rate per clock (MI355X_MICROARCH.md, Matrix cores) =
# 131 072 ops per 32 cycles per SIMD; 1024 SIMDs at 2.4 GHz = 10.07 POP/s dense (tools/ubench/mfma_fp4_probe.hip measures
# 8.4 POP/s for a bare register-resident loop on random signs)
MFMA_FP4_PEAK = 1024 * 131072 / 32 * 2.4e9
# popcount scan (fallback path): SURVEY 8(d)'s bound, wave64 VALU at 2 clocks per instruction (SIMD-32): one v_xor + one
# v_bcnt per 32-bit word and 64 pairs = 256 B of logical code bytes per 4 clocks per SIMD
VALU_POPCNT_PEAK = 1024 * 256 / 4 * 2.4e9

# Indexed 1 KiB-row gathers served on-die (MI355X_MICROARCH.md, "Indexed rows"): rows every workgroup shares (an XCD's L2)
# 16.8-18.8 TB/s chip-wide, uniformly random rows of a 38 MB Infinity-Cache-resident table 8.6 TB/s.  The pooled hidden rows
# (M x H x 4 = 60 MB at ML-25M) are a popularity-skewed mix of both, so the upper figure is the bound when the table fits.
# what a kernel that issues nothing but MFMAs reaches on this part (profiles/r01_mfma_rate_ubench.txt: 141-147 TFLOP/s fp32;
# profiles/r02_mfma_fp4_vs_i8_probe.txt: 16.0 ns per fp4 32x32x64 per SIMD = 8.39 POP/s): reported beside the nominal peaks
MFMA_F32_MEASURED = 145.0e12
MFMA_FP4_MEASURED = 8.39e15
RANDOM_SECTOR_PEAK = 55.4e9     # random 64-byte sectors per second beyond the L2, measured (tools/ubench/gather_rate.hip): 3.5 TB/s
MALL_BYTES = 256 << 20
ONDIE_GATHER_PEAK = 18.8e12

# BASELINE.json configs that fit one GPU (configs[1], configs[2]); the default run is the configuration the metric is
# quoted on (d = 256, T = 10, 512-bit codes).  5 = configs[4] (100 M items / 10^9 ratings, d = 256, T = 10, 8 GPUs) as ONE of
# its eight ranks sees it: the whole graph (2 x 10^9 directed edges) replicated in HBM, one item shard of 12.5 M, Philox
# uniforms; the gathers of the other seven ranks' rows are stood in for by local buffers of the gathered shape.
PRESETS = {2: dict(dim=128, T=10, lsh_bits=256), 3: dict(dim=256, T=50, lsh_bits=512),
           5: dict(dim=256, T=10, lsh_bits=512, rng="philox")}
CONFIG5 = dict(num_users=10_000_000, num_items=100_000_000, num_ratings=1_000_000_000, ranks=8)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="default 50 (config 5: 5)")
    ap.add_argument("--warmup", type=int, default=-1, help="default 5 (config 5: 1)")
    ap.add_argument("--dim", type=int, default=256, help="embedding dim d (BASELINE metric: d=256)")
    ap.add_argument("--T", type=int, default=10, help="neighbours kept per node (north_star: T=10)")
    ap.add_argument("--lsh-bits", type=int, default=0, help="default 2*dim (256-bit @128, 512-bit @256)")
    ap.add_argument("--queries", type=int, default=10000)
    ap.add_argument("--k", type=int, default=11, help="num_recommendations + 1 (inference.py:110)")
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of ML-25M (tests)")
    ap.add_argument("--rng", default="numpy", choices=["philox", "numpy"],
                    help="numpy = the reference's global np.random MT19937 stream (bit-exact ids; the mode the goldens pin); "
                         "philox = counter-based, shard-count invariant, no stream generation")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 5],
                    help="BASELINE.json config preset: 2 = d128/T10/256-bit, 3 = d256/T50/512-bit (the rocprof roofline run), "
                         "5 = one rank's shard of the 100 M-item / 10^9-rating stress graph (HBM-bound sampler)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="start items timed on the CPU (0 = the whole catalogue)")
    a = ap.parse_args()
    if a.config:
        for k, v in PRESETS[a.config].items():
            setattr(a, k, v)
    if a.steps <= 0:
        a.steps = 5 if a.config == 5 else 50
    if a.warmup < 0:
        a.warmup = 1 if a.config == 5 else 5
    return a


def ceil_log2p1(d):
    return torch.ceil(torch.log2((d + 1).double())).long()


def sampler_algorithmic_bytes(graph, sampler, nodes, T, W, L, call, stream_mode, chunk=1 << 20):
    """per taken step from v: 8 B (the node's row bounds: two 32-bit offsets -- the kernel reads ONE 8-byte
    (row start, degree) record, csrc/walk_sample.hip load_row; edge ids are uint32, E < 2^32 is enforced by ps_csr_build)
    + 8 B * ceil(log2(deg v + 1)) (fp64 CDF probes of the binary search) + 4 B (col) [+ 8 B uniform in stream mode];
    per start node T * 8 B + 4 B of output.  The walks are replayed exactly with ps_walk_paths (same Philox counters),
    `chunk` start nodes at a time."""
    from pinsage_hip import sampling
    deg = graph.rowptr[1:] - graph.rowptr[:-1]
    per_step_fixed = 8 + 4 + (8 if stream_mode else 0)
    steps, nbytes = 0, 0
    for c0 in range(0, nodes.numel(), chunk):
        nd = nodes[c0:c0 + chunk]
        d0 = deg[nd]
        act = d0 > 0
        steps += int(act.sum().item()) * W
        nbytes += int(((ceil_log2p1(d0) * 8 + per_step_fixed) * act).sum().item()) * W
        if L > 1:
            starts = nd.repeat_interleave(W)
            paths = sampling.walk_paths(graph, starts, L, rng="philox", seed=sampler.seed, call=call, walk_mod=W)
            for st in range(1, L):
                src = paths[:, st - 1].long()
                ok = src >= 0
                d = deg[src.clamp(min=0)] * ok
                took = d > 0
                steps += int(took.sum().item())
                nbytes += int(((ceil_log2p1(d) * 8 + per_step_fixed) * took).sum().item())
            del starts, paths
    nbytes += nodes.numel() * (T * 8 + 4)
    return nbytes, steps


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # PS_BENCH_BACKEND=gloo + PS_BENCH_SHARE_GPU=1: rehearsal of the N > 1 path on a one-GPU box (all ranks on
    # cuda:0, collectives staged through the host); the driver's runs use RCCL, one GPU per rank.
    backend = os.environ.get("PS_BENCH_BACKEND", "nccl")
    if os.environ.get("PS_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from pinsage_hip import synth, dense, sampling
    from pinsage_hip import native as nv
    from pinsage_hip.graph import DeviceGraph
    from pinsage_hip.shard import ShardedPinSage
    from utils.random_walk import RandomWalkSampler
    from utils.nearest_neighbors import lsh_rotation_matrix
    from model.pinsage import PinSage

    big = a.config == 5
    if big:
        assert world == 1, "--config 5 plays ONE rank of the 8-rank job on one GPU (the driver's N > 1 runs use the default config)"
        src = {k: max(8, int(v * a.scale)) for k, v in CONFIG5.items() if k != "ranks"}
    else:
        src = {k: max(8, int(v * a.scale)) for k, v in synth.ML25M.items()}
    U, M, R = src["num_users"], src["num_items"], src["num_ratings"]
    F_IN, HID, D, LAYERS, W, L, T = 128, 256, a.dim, 2, 100, 2, a.T
    nbits = a.lsh_bits or 2 * D
    ei, ew = synth.bipartite_ratings(U, M, R, seed=20240601, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    graph = DeviceGraph(ei, ew, device=dev)        # CSR + CDF + guide + packed blocks + bucket records (first call: incl. allocations)
    del ei, ew
    torch.cuda.synchronize()
    t_graph = time.time() - t0
    if big:
        # 66 GB of graph + 64 GB of half bucket records + the 102 GB stand-in for the gathered hidden rows: while the job steps, the
        # plain col / cdf / guide arrays (32 GB) and the sorted weights (16 GB) are dropped -- the walk kernel reads the same values
        # from the packed blocks -- and restored (bit for bit) for the accounting and the CPU oracle afterwards
        graph.compact()
        torch.cuda.empty_cache()
    sampler = RandomWalkSampler.from_graph(graph, walk_length=L, num_walks=W, rng=a.rng, seed=42)
    torch.manual_seed(2)
    model = PinSage(F_IN, HID, D, LAYERS).to(dev).eval()
    params = {k: v.detach().float().contiguous() for k, v in model.state_dict().items()}
    A = torch.from_numpy(lsh_rotation_matrix(D, nbits)).to(dev)
    sim_ranks = CONFIG5["ranks"] if big else 1
    pipe = ShardedPinSage(params, LAYERS, sampler, M, standin=(0, sim_ranks) if big else None)
    if big:
        x_loc = torch.randn(pipe.hi - pipe.lo, F_IN, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        x_rep = None                               # hidden rows are gathered (a replicated feature table would be 51 GB)
    else:
        gen = torch.Generator(device="cpu").manual_seed(1)
        x_full = torch.randn(M, F_IN, generator=gen)
        x_loc = x_full[pipe.lo:pipe.hi].to(dev).contiguous()
        x_rep = x_full.to(dev).contiguous() if world > 1 else None      # replicated features (30 MB): lets every rank
        del x_full                                                       # recompute layer-0 rows instead of gathering them
    ranks_total = world * sim_ranks
    nq_local = max(1, a.queries // ranks_total)
    nq = nq_local * ranks_total
    items_per_step = (pipe.hi - pipe.lo) if big else M              # config 5: this rank's shard is what a step processes

    ev = lambda: torch.cuda.Event(enable_timing=True)
    phase_ms = {"embed": 0.0, "index": 0.0, "query": 0.0}

    def step(record=None):
        if a.rng == "numpy":
            np.random.seed(42)
        e0, e1, e2, e3 = ev(), ev(), ev(), ev()
        e0.record()
        emb = pipe.embed(x_loc, T, x_full=x_rep)
        e1.record()
        pipe.build_index(emb, A)
        e2.record()
        d, i = pipe.search(emb[:nq_local], a.k)
        e3.record()
        if record is not None:
            record.append((e0, e1, e2, e3))
        return emb, d, i

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    with torch.no_grad():
        # settle: the first ~100 ms after idle run at a lower clock (measured: 5.0 ms/step after 3 steps
        # vs 3.6 ms/step after 30); run untimed steps for >= 0.5 s, then the W warm-up steps proper
        t_settle = time.perf_counter()
        while True:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            # every step contains collectives, so all ranks must run the same number of settle steps: rank 0
            # decides, everybody follows
            go_on = torch.tensor([1 if time.perf_counter() - t_settle < 0.5 else 0], dtype=torch.int32,
                                 device=dev if backend == "nccl" else "cpu")
            if world > 1:
                dist.broadcast(go_on, src=0)
            if int(go_on.item()) == 0:
                break
        for _ in range(a.warmup):
            step()
        # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides ----
        recs_clean = []                     # the four phase events of every timed step (they are recorded in every step anyway)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            emb, d_out, i_out = step(recs_clean)
        sync_all()
        elapsed = time.perf_counter() - t0
        # ---- instrumented region: the same K steps again with one HIP-event pair around every kernel
        # launch (on the launch stream) and around the three phases.  Timing events are barrier packets on
        # ROCm (~46 per step), so this region is slower than the clean one and is NOT used for `value`.
        timer = nv.KernelTimer()
        recs = []
        pipe.overlap_sampling = False       # per-kernel durations are measured with the kernels running alone
        overlap_proj = pipe.overlap_input_proj
        pipe.overlap_input_proj = False
        sync_all()
        nv.set_timer(timer)
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step(recs)
        sync_all()
        elapsed_instr = time.perf_counter() - t1
        nv.set_timer(None)
        pipe.overlap_sampling = False
        pipe.overlap_input_proj = overlap_proj
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # phases: from the clean region (the per-kernel event pairs of the instrumented one stretch the host-bound parts)
    for (e0, e1, e2, e3) in recs_clean:
        phase_ms["embed"] += e0.elapsed_time(e1)
        phase_ms["index"] += e1.elapsed_time(e2)
        phase_ms["query"] += e2.elapsed_time(e3)
    ms_per_step = elapsed / a.steps * 1e3
    value = items_per_step * a.steps / elapsed
    ksum = timer.summary()

    gpu_ref = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        gpu_ref = parity_gpu_step(a, graph, pipe, sampler, step, M, T, W, L, LAYERS, big)
    if big:
        # the step's buffers (102 GB stand-in, activations) go, the plain graph arrays come back for the accounting / the oracle
        emb = d_out = i_out = None
        pipe.comm._bufs.clear()
        torch.cuda.empty_cache()
        graph.expand()
    out = None
    if rank == 0:
        # ---------------- roofline of every kernel (rank 0's shard), algorithmic bytes / flops ---------
        nodes = torch.arange(pipe.lo, pipe.hi, dtype=torch.int64, device=dev)
        n_loc = nodes.numel()
        stream_mode = a.rng == "numpy"
        sb0, steps0 = sampler_algorithmic_bytes(graph, sampler, nodes, T, W, L, 0, stream_mode)
        sb1, steps1 = sampler_algorithmic_bytes(graph, sampler, nodes, T, W, L, 1, stream_mode)
        samp_bytes = (sb0 + sb1) / 2.0
        b0 = sampling.walk_sample(graph, nodes, T, W, L, rng="philox", seed=42, call=0)
        valid = ((b0.ids >= 0) & (b0.ids <= M - 1)).sum().item()
        del b0
        pool_bytes = valid * HID * 4 + n_loc * HID * 4 + n_loc * T * 8
        # where the gathered rows live decides the bound: a hidden-row table that fits the Infinity Cache (60 MB at ML-25M)
        # is served on-die (L2 hits for popular rows, MALL for the rest); config 5's gathered table (102 GB) is HBM
        pool_table = pipe.world * pipe.chunk * HID * 4 if pipe.world > 1 else M * HID * 4
        pool_bound, pool_peak = ("mall", ONDIE_GATHER_PEAK) if pool_table <= MALL_BYTES else ("hbm", HBM_PEAK)
        # executed flops: lin_self is composed into lin_update once per forward (2 * H^3 + 2 * H^2 per layer),
        # so per item: input_proj + layers * (h W'^T + h_neigh Wu2^T) + output_proj
        lin_flops_step = 2.0 * n_loc * (F_IN * HID + LAYERS * (2 * HID * HID) + HID * D) + LAYERS * (2.0 * HID ** 3 + 2.0 * HID * HID)
        enc_flops = 2.0 * D * nbits                              # per encoded row
        kern = {}

        def add(name, bound, per_launch_work, unit_peak):
            if name not in ksum:
                return
            k = ksum[name]
            ach = per_launch_work / (k["avg_ms"] * 1e-3)
            kern[name] = {"launches_per_step": k["launches"] / a.steps, "avg_ms": round(k["avg_ms"], 4),
                          "ms_per_step": round(k["ms"] / a.steps, 4), "bound": bound, "achieved": ach,
                          "peak": unit_peak, "frac": ach / unit_peak}

        # the fused launch samples LAYERS rounds per start node: per-launch work = the sum over the layers
        add("ps_walk_sample_layers", "hbm", sb0 + sb1, HBM_PEAK)
        add("ps_walk_sample", "hbm", samp_bytes, HBM_PEAK)
        add("ps_importance_pool", pool_bound, pool_bytes, pool_peak)
        if "ps_linear" in ksum:
            add("ps_linear", "mfma", lin_flops_step / (ksum["ps_linear"]["launches"] / a.steps), MFMA_F32_PEAK)
        if "ps_lsh_encode" in ksum:
            rows = (n_loc + nq_local) / 2.0                      # two launches per step: index + queries
            add("ps_lsh_encode", "mfma", enc_flops * rows, MFMA_F32_PEAK)
        # Hamming scan as an exact +-1 contraction on fp4 MFMA: 2 * nq * N * nbits sign operations (dot = nbits - 2 *
        # hamming); the C-ABI call covers the bound pass (1/5 of the table again), the collect pass and the slice merge
        add("ps_hamming_topk_mfma_codes", "mfma", 2.0 * nq * n_loc * nbits, MFMA_FP4_PEAK)
        # popcount fallback (shapes the MFMA path does not serve): VALU bound on the logical code bytes
        add("ps_hamming_topk", "valu", float(nq) * n_loc * (nbits // 8), VALU_POPCNT_PEAK)
        for mt_call in ("ps_mt19937_raw_stream", "ps_mt19937_random_sample"):     # numpy-stream mode: 8 B written per uniform
            if mt_call in ksum:
                add(mt_call, "hbm", 8.0 * (steps0 + steps1), HBM_PEAK)
        # counter traffic per C-ABI call (FETCH_SIZE + WRITE_SIZE of separate --pmc passes over this same command, committed
        # under profiles/; as reported by rocprofv3 -- see the file's "units" for the gfx950 caveats): shown next to the
        # algorithmic rate of every call as bytes per launch and GB/s over the launch time measured in THIS run
        tfile = "pmc_traffic_config5.json" if big else "pmc_traffic_latest.json"
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", tfile)))
        except Exception:
            traffic = {}
        for n, k in kern.items():
            if k.get("bound") == "mfma":
                meas = MFMA_FP4_MEASURED if n == "ps_hamming_topk_mfma_codes" else MFMA_F32_MEASURED
                k["frac_of_measured_mfma_rate"] = k["achieved"] / meas
            if isinstance(traffic.get(n), (int, float)) and traffic[n] > 0:
                k["counter_bytes_per_launch"] = traffic[n]
                k["counter_GBps"] = traffic[n] / (k["avg_ms"] * 1e-3) / 1e9
                if n in ("ps_walk_sample_layers", "ps_walk_sample"):
                    # The sampler's gathers are one 64-byte sector each (calibrated: FETCH_SIZE counts exactly the sectors that leave
                    # the L2), and the memory system beyond the L2 delivers RANDOM sectors at 55 G/s whatever the table size, loads
                    # in flight or dependency (tools/ubench/gather_rate.hip, profiles/r03_gather_rate_ubench.txt; L2-resident:
                    # 267 G/s): the bound this kernel is actually up against, next to the contract's algorithmic-bytes-over-HBM figure
                    k["sectors_beyond_l2_per_launch"] = traffic[n] / 64.0
                    k["random_sector_rate_Gps"] = traffic[n] / 64.0 / (k["avg_ms"] * 1e-3) / 1e9
                    k["random_sector_peak_Gps"] = RANDOM_SECTOR_PEAK / 1e9
                    k["frac_of_random_sector_peak"] = k["random_sector_rate_Gps"] * 1e9 / RANDOM_SECTOR_PEAK
        # `roofline`: north_star's roofline target is the sampler (">= 40 % HBM-roofline on the sampler"), which is also the longest
        # single launch of the step; the C-ABI call with the largest time per step -- ps_linear's FOUR launches together come to
        # about the same -- is named beside it with its own fraction (`dominant_kernel`, `dominant_frac`), so nothing hides behind
        # the choice
        dom = max(kern, key=lambda n: kern[n]["ms_per_step"])
        dominant_call = dom
        smp_call = "ps_walk_sample_layers" if "ps_walk_sample_layers" in kern else "ps_walk_sample"
        if smp_call in kern:
            dom = smp_call
        kd = kern[dom]
        div = 1e9 if kd["bound"] in ("hbm", "valu", "mall") else 1e12
        unit = {"hbm": "GB/s", "valu": "GB/s", "mall": "GB/s", "mfma": "TFLOP/s"}[kd["bound"]]
        if dom == "ps_hamming_topk_mfma_codes":
            unit = "TOP/s"
        roofline = {"kernel": dom, "dominant_kernel": dominant_call,
                    "dominant_kernel_ms_per_step": kern[dominant_call]["ms_per_step"], "dominant_bound": kern[dominant_call]["bound"],
                    "dominant_frac": kern[dominant_call]["frac"], "kernel_ms_per_step": kd["ms_per_step"],
                    "bound": kd["bound"], "achieved": kd["achieved"] / div, "peak": kd["peak"] / div,
                    "unit": unit, "frac": kd["frac"],
                    "traffic": traffic.get(dom), "traffic_source": traffic.get("source") if dom in traffic else None,
                    "avg_launch_ms": kd["avg_ms"], "algorithmic_per_launch": kd["achieved"] * kd["avg_ms"] * 1e-3,
                    "note": {"ps_hamming_topk_mfma_codes": "exact +-1 contraction on fp4 MFMA; peak = dense fp4 (4 x bf16 per clock at 2.4 GHz); "
                                                     "the call includes the bound pass over 1/5 of the table, counted as overhead",
                             "ps_walk_sample_layers": "algorithmic bytes per SURVEY 8(d) (8 B row bounds + 8 B x ceil(log2(deg+1)) CDF "
                                                      "probes + 4 B col [+ 8 B uniform] per taken step, T x 8 + 4 B out per start "
                                                      "node), both layers of a start node in one launch"}.get(dom)}
        if roofline["bound"] == "mall":
            roofline["bound"] = "hbm"          # the contract's vocabulary; the on-die bound is named in kernels[*]
        # device kernels behind each C-ABI call (the rows of profiles/*/kernel_stats.csv the timings agree with)
        symbols = {"ps_walk_sample_layers": ["walk_sample_kernel<4, STREAM>"], "ps_walk_sample": ["walk_sample_kernel<4, STREAM>"],
                   "ps_importance_pool": ["importance_pool4_kernel<PAGES> (four output rows per wave; T <= 64)"],
                   "ps_linear": ["gemm_f32_pkernel<2,2,1,2,32,0> (input_proj: no row norm, 64x128 tiles, persistent)",
                                 "gemm_f32_kernel<1,4,2,2,32,0,true> (layers + output_proj: fused L2 norm, 64x256 tiles)"],
                   "ps_lsh_encode": ["gemm_f32_pkernel<2,2,1,2,32,1>"],
                   "ps_hamming_topk_mfma_codes": ["hamming_pipe_kernel<KS, 0> (bound)", "bound_select_kernel",
                                            "hamming_pipe_kernel<KS, 1> (collect)", "slice_merge_kernel"],
                   "ps_hamming_topk": ["hamming_scan_kernel<16,4>", "topk_rank_merge_kernel"],
                   "ps_mt19937_raw_stream": ["mt_begin", "mt_planes", "mt_jump_mfma", "mt_jump_reduce", "mt_jump_finish", "mt_chunk (both directions)"],
                   "ps_mt19937_random_sample": ["mt_* (jump-ahead windows + chunk generators + mt_raw_to_double)"]}
        for n, k in kern.items():
            k["device_kernels"] = symbols.get(n, [])
        if "ps_importance_pool" in kern:
            kern["ps_importance_pool"]["note"] = (
                f"algorithmic bytes = gathered rows + output (SURVEY 8d); the gathered table is {pool_table / 1e6:.0f} MB: "
                + ("it stays in L2 / Infinity Cache, so the bound is the on-die indexed-row gather rate of MI355X_MICROARCH.md "
                   "(16.8-18.8 TB/s for L2-shared rows, 8.6 TB/s for uniformly random MALL rows; the upper figure is used), not HBM"
                   if pool_bound == "mall" else "far beyond the Infinity Cache: random 1 KiB rows from HBM"))
        for n, k in kern.items():
            if k["frac"] > 1.0 + 1e-9:     # a fraction above 1 means the bound is the wrong one for this shape: flag it, keep the line
                k["bound_suspect"] = True
                print(f"bench.py: {n} runs at {k['frac']:.2f} of its '{k['bound']}' bound -- wrong bound for this shape", file=sys.stderr)
            div = 1e12 if k["bound"] == "mfma" else 1e9
            k["achieved"] = k["achieved"] / div
            k["peak"] = k["peak"] / div
            k["unit"] = "TFLOP/s" if k["bound"] == "mfma" else "GB/s"
        if "ps_hamming_topk_mfma_codes" in kern:
            kern["ps_hamming_topk_mfma_codes"]["unit"] = "TOP/s"

        out = {
            "metric": "item embeddings/sec + top-K ANN queries/sec, ML-25M d=256, 1/2/4/8 GPU",
            "value": value, "unit": "items/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64 cdf / f32 features / fp4 sign planes (exact integer Hamming)", "data": "synthetic",
            "config": {"workload": (f"SYN-1B (BASELINE config 5: U={U} M={M} R={R} = {2 * R} directed edges, replicated) as rank 0 of "
                                    f"{sim_ranks} sees it: item shard [{pipe.lo}, {pipe.hi}), gathers of the other ranks' rows stood in "
                                    f"for by local buffers of the gathered shape (no xGMI time); value = shard items / step; "
                                    if big else f"SYN-25M (ML-25M-shaped: U={U} M={M} R={R}), ")
                                   + f"F=128 H=256 d={D}, 2 GCN layers, "
                                   f"W=100 L=2 T={T}, LSH {nbits}-bit, {nq} queries K={a.k}, rng={a.rng}"
                                   + (" (the reference's np.random MT19937 stream generated on device: neighbour ids bit-exact with "
                                      "the reference CPU path)" if a.rng == "numpy" else " (counter-based, shard invariant)")
                                   + (f", BASELINE config {a.config}" if a.config else ""),
                       "global_items": M, "items_per_step": items_per_step, "queries": nq,
                       "parallelism": f"item-shard x{world}" if not big else f"one rank of item-shard x{sim_ranks}"},
            "embeddings_per_s": items_per_step / (phase_ms["embed"] / a.steps * 1e-3),
            "index_items_per_s": items_per_step / (phase_ms["index"] / a.steps * 1e-3),
            "queries_per_s": nq / (phase_ms["query"] / a.steps * 1e-3),
            "phase_ms": {k: round(v / a.steps, 4) for k, v in phase_ms.items()},
            "ms_per_step_instrumented": elapsed_instr / a.steps * 1e3,
            "graph_build_s": round(t_graph, 3),
            "roofline": roofline, "kernels": kern,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"], out["parity_check"] = cpu_baseline(a, graph, pipe, gpu_ref, params, x_loc, A, M, T, W, L,
                                                                    LAYERS, HID, D, nbits, nq, big)
            out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if world == 1 and not big:            # config 5: 5 x 10^9 stream uniforms per shard pass -- Philox only
            out["other_rng_mode"] = other_mode_probe(a, graph, params, LAYERS, M, x_loc, A, T, W, L, nq_local)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None and not out.get("parity_check", {}).get("ok", True):
        print(f"bench.py: the GPU step does not match the oracle: {out['parity_check']}", file=sys.stderr)
        sys.exit(1)


def other_mode_probe(a, graph, params, LAYERS, M, x_loc, A, T, W, L, nq_local):
    """The same step in the RNG mode that is NOT the headline (philox when the run is rng=numpy and vice versa),
    timed over a few steps after a short warm-up: both modes are reported by every run."""
    from pinsage_hip.shard import ShardedPinSage
    from utils.random_walk import RandomWalkSampler
    other = "philox" if a.rng == "numpy" else "numpy"
    smp = RandomWalkSampler.from_graph(graph, L, W, rng=other, seed=42)
    pipe = ShardedPinSage(params, LAYERS, smp, M)

    def step():
        if other == "numpy":
            np.random.seed(42)
        emb = pipe.embed(x_loc, T)
        pipe.build_index(emb, A)
        return pipe.search(emb[:nq_local], a.k)

    with torch.no_grad():
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        n = max(5, min(20, a.steps))
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    return {"rng": other, "value": M / dt, "unit": "items/s", "ms_per_step": dt * 1e3,
            "note": "same step (embed + index + query) with the other RNG mode; philox = counter-based uniforms computed in "
                    "the walk kernel, numpy = the reference's global MT19937 stream generated on the device"}


def cpu_forward(co, P, x, h_table, layers, threads):
    """Pooled forward on the host: the C oracle's pooling + torch CPU dense layers (what the reference runs,
    model/pinsage.py:202-249).  h_table=None: the pooling gathers the REAL hidden rows of the previous layer (x holds every
    item); otherwise rows of the given stand-in table (a sample of start items of a catalogue too large to embed here)."""
    F = torch.nn.functional
    with torch.no_grad():
        h = torch.relu(F.linear(x, P["input_proj.weight"], P["input_proj.bias"]))
        for i in range(len(layers)):
            src = h.numpy() if h_table is None else h_table
            hn = torch.from_numpy(co.importance_pool(src, layers[i][0], layers[i][1], layers[i][2], threads=threads))
            hs = F.linear(h, P[f"convs.{i}.lin_self.weight"], P[f"convs.{i}.lin_self.bias"])
            h = torch.relu(F.linear(torch.cat([hs, hn], 1), P[f"convs.{i}.lin_update.weight"], P[f"convs.{i}.lin_update.bias"]))
            h = F.normalize(h, dim=1)
        return F.normalize(F.linear(h, P["output_proj.weight"], P["output_proj.bias"]), dim=1)


def oracle_samples(co, cg, nodes, T, W, L, layers, rng, seed, threads, call0=0):
    """`layers` consecutive batch_sample_neighbors calls on the C oracle.  rng='numpy': the uniforms are the legacy MT19937
    stream of np.random.seed(seed) consumed in the reference's order (utils/random_walk.py:79), drawn here with a private
    RandomState; returns the RandomState too (its next draw is what np.random must return after the GPU's pass)."""
    out, rs = [], None
    if rng == "numpy":
        rs = np.random.RandomState(seed)
        uoff, n = cg.uniform_offsets(nodes, W, L)
    for r in range(layers):
        if rng == "numpy":
            ids, counts, nv, _, _, _ = co.walk_sample(cg, nodes, T, L, W, uniforms=rs.random_sample(n), uoff=uoff, threads=threads)
        else:
            ids, counts, nv, _, _, _ = co.walk_sample(cg, nodes, T, L, W, philox=(seed, call0 + r), threads=threads)
        out.append((ids, counts, nv))
    return out, rs


def cpu_sample_nodes(a, pipe, M, big):
    """the start items the CPU leg runs: the whole catalogue when it fits, otherwise a uniform sample of the shard"""
    n_items = pipe.hi - pipe.lo if big else M                 # what one step embeds
    if big and a.cpu_sample <= 0:
        a.cpu_sample = 65536
    S = n_items if a.cpu_sample <= 0 else min(a.cpu_sample, n_items)
    whole = (S == n_items) and not big
    rs = np.random.RandomState(0)
    nodes = np.sort(rs.choice(n_items, size=S, replace=False)) + (pipe.lo if big else 0)
    return n_items, S, whole, nodes


def parity_gpu_step(a, graph, pipe, sampler, step, M, T, W, L, LAYERS, big):
    """The GPU step the oracle is compared with: one more step from a known RNG state (np.random.seed(42) / Philox call 0),
    outside every timed region, and the sampler's output for the same state."""
    from pinsage_hip import sampling
    n_items, S, whole, nodes = cpu_sample_nodes(a, pipe, M, big)
    with torch.no_grad():
        sampler._calls = 0
        emb_g, d_g, i_g = step()
        tail_g = np.random.random_sample() if a.rng == "numpy" else None
        codes_g = pipe.codes
        sampler._calls = 0
        if a.rng == "numpy":
            np.random.seed(42)
        if whole:
            got = sampler.sample_batches(range(pipe.lo, pipe.hi), T, LAYERS)
        else:                                                  # config 5 / --cpu-sample: the sampled start nodes only (Philox)
            nd = torch.from_numpy(nodes).to(emb_g.device)
            got = [sampling.walk_sample(graph, nd, T, W, L, rng="philox", seed=42, call=c) for c in range(LAYERS)]
        torch.cuda.synchronize()
    return {"emb": emb_g, "d": d_g, "i": i_g, "tail": tail_g, "codes": codes_g, "got": got}


def cpu_baseline(a, graph, pipe, gpu_ref, params, x_loc, A, M, T, W, L, LAYERS, HID, D, nbits, nq, big=False):
    """The CPU oracle (C port of the reference algorithm, pinned by the goldens) timed on this box's host cores on the
    same workload -- the whole step when the catalogue fits (SYN-25M: every start item, both layers, the real pooled forward
    over the real hidden rows, all queries), otherwise a bounded sample -- with torch CPU for the dense layers (what the
    reference runs).  The oracle's outputs are then COMPARED with the GPU's (`parity_check`): one more GPU step is run from a
    known RNG state (np.random.seed(42) / Philox call 0) and its sampled neighbour ids / visit counts, embeddings, LSH codes
    and top-k (distance, id) lists are held to the oracle's -- bit-exact for ids / counts / codes / top-k, 1e-5 for the
    fp32 embeddings (north_star's tolerances).
    Config 5: the sample is 65 536 start items of the shard and 256 queries over the shard's codes (the CSR + CDF, 25 GB,
    are copied to the host for the oracle), scaled to the shard; embeddings cannot be compared there (the gathered rows of
    the other seven ranks are stand-ins)."""
    from oracle import c_oracle as co
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(threads, co.max_threads()))
    torch.set_num_threads(threads)
    dev = x_loc.device
    cg = co.Graph.from_arrays(graph.rowptr.cpu().numpy(), graph.col.cpu().numpy(), graph.cdf.cpu().numpy())
    n_items, S, whole, nodes = cpu_sample_nodes(a, pipe, M, big)
    emb_g, d_g, i_g, tail_g, codes_g, got = (gpu_ref[k] for k in ("emb", "d", "i", "tail", "codes", "got"))
    # ---- timed: sampler ----
    rng_cpu = a.rng if whole else "philox"
    t0 = time.perf_counter()
    layers, rs42 = oracle_samples(co, cg, nodes, T, W, L, LAYERS, rng_cpu, 42, threads)
    t_sample = time.perf_counter() - t0
    # ---- timed: pooled forward ----
    P = {k: v.cpu() for k, v in params.items()}
    if whole:
        xs, h_table = x_loc.cpu(), None
    else:
        xs = torch.randn(S, 128)
        # hidden rows of every item for the pooling gather; config 5: 102 GB of zero pages that are never written (reads of
        # untouched anonymous memory share the kernel's zero page), so the gather's address stream is the real one
        h_table = np.zeros((M, HID), dtype=np.float32) if big else torch.randn(M, HID).numpy()
    t0 = time.perf_counter()
    e = cpu_forward(co, P, xs, h_table, layers, threads)
    t_dense = time.perf_counter() - t0
    Ah = A.cpu()
    t0 = time.perf_counter()
    bits = (e @ Ah.t()) >= 0
    np.packbits(bits.numpy(), axis=1, bitorder="little")
    t_enc = time.perf_counter() - t0
    codes_all = codes_g.cpu().numpy()                         # the index built by the GPU pass (held to the oracle's encode below)
    Sq = nq if a.cpu_sample <= 0 else min(256 if big else 2048, nq)
    t0 = time.perf_counter()
    d_o, i_o = co.hamming_topk(codes_all[:Sq], codes_all, a.k, id_offset=pipe.lo, threads=threads)
    t_q = time.perf_counter() - t0
    per_item = (t_sample + t_dense + t_enc) / S
    step_s = per_item * n_items + t_q / Sq * nq
    Ncodes = codes_all.shape[0]
    # ---- parity: the GPU step above against the oracle's outputs (untimed) ----
    par = {"rng": a.rng if whole else "philox", "start_items": int(S), "layers": LAYERS}
    ok_ids = True
    for r in range(LAYERS):
        ok_ids &= np.array_equal(got[r].ids.cpu().numpy().astype(np.int64), layers[r][0])
        ok_ids &= np.array_equal(got[r].counts.cpu().numpy(), layers[r][1])
        ok_ids &= np.array_equal(got[r].nvalid.cpu().numpy(), layers[r][2])
    par["sampler_ids"] = bool(ok_ids)
    if rs42 is not None:
        par["np_random_state_after"] = bool(tail_g == rs42.random_sample())
    if whole:
        eg, ec = emb_g.cpu().numpy(), e.numpy()
        par["embeddings_max_rel"] = float(np.max(np.linalg.norm(eg - ec, axis=1) / np.linalg.norm(ec, axis=1)))
        par["embeddings_max_abs"] = float(np.max(np.abs(eg - ec)))
        par["embeddings"] = bool(np.allclose(eg, ec, rtol=1e-5, atol=2e-6))
        # the other RNG mode: sampler only
        other = "philox" if a.rng == "numpy" else "numpy"
        from utils.random_walk import RandomWalkSampler
        smp_o = RandomWalkSampler.from_graph(graph, L, W, rng=other, seed=42)
        np.random.seed(42)
        got_o = smp_o.sample_batches(range(pipe.lo, pipe.hi), T, LAYERS)
        lay_o, _ = oracle_samples(co, cg, nodes, T, W, L, LAYERS, other, 42, threads)
        par["sampler_ids_" + other] = bool(all(
            np.array_equal(got_o[r].ids.cpu().numpy().astype(np.int64), lay_o[r][0])
            and np.array_equal(got_o[r].counts.cpu().numpy(), lay_o[r][1])
            and np.array_equal(got_o[r].nvalid.cpu().numpy(), lay_o[r][2]) for r in range(LAYERS)))
    # codes: the oracle's k-ordered fmaf projection of the GPU's embeddings (rows of the sample when the catalogue is sampled)
    rows = np.arange(n_items) if whole else (nodes - pipe.lo)
    rows = rows[: min(rows.size, 65536)] if not whole else rows
    enc = co.lsh_encode(emb_g[torch.from_numpy(rows).to(dev)].cpu().numpy(), Ah.numpy(), threads=threads)
    par["codes"] = bool(np.array_equal(enc, codes_all[rows]))
    par["codes_rows"] = int(rows.size)
    dg, ig = d_g[:Sq].cpu().numpy(), i_g[:Sq].cpu().numpy()
    par["topk_ids"] = bool(np.array_equal(ig, i_o))
    par["topk_dist"] = bool(np.array_equal(dg.astype(np.float32), d_o))
    par["topk_queries"] = int(Sq)
    par["ok"] = all(v for k, v in par.items() if isinstance(v, bool))
    # the same port on ONE core (SURVEY 8d asks for both), on a small slice: 256 start items, 16 queries
    S1, Q1 = min(256, S), min(16, Sq)
    torch.set_num_threads(1)
    # a slice cannot pool from its own rows: a stand-in table of the right shape
    h1 = np.random.RandomState(1).standard_normal((M, HID)).astype(np.float32) if whole else h_table
    t0 = time.perf_counter()
    l1, _ = oracle_samples(co, cg, nodes[:S1], T, W, L, LAYERS, "philox", 42, 1)
    e1 = cpu_forward(co, P, xs[:S1], h1, l1, 1)
    np.packbits(((e1 @ Ah.t()) >= 0).numpy(), axis=1, bitorder="little")
    t_item1 = (time.perf_counter() - t0) / S1
    t0 = time.perf_counter()
    co.hamming_topk(codes_all[:Q1], codes_all, a.k, threads=1)
    t_q1 = (time.perf_counter() - t0) / Q1
    torch.set_num_threads(threads)
    single = {"value": n_items / (t_item1 * n_items + t_q1 * nq), "unit": "items/s", "cores": 1,
              "sample": f"{S1} start items + {Q1} queries over all {Ncodes} codes, scaled to the full step"}
    base = {"value": n_items / step_s, "unit": "items/s", "cores": threads, "kind": "port", "single_thread": single,
            "sample": (f"the whole step: all {S} start items (sampler x{LAYERS} layers, rng={rng_cpu}, pooling over the real hidden "
                       f"rows, dense, LSH encode) + all {Sq} queries over {Ncodes} codes" if whole and Sq == nq else
                       f"{S} uniformly drawn start items (sampler x{LAYERS} layers, pooling, dense, LSH encode) + {Sq} queries "
                       f"over all {Ncodes} codes, scaled to the full step"),
            "seconds": {"sampler": round(t_sample, 3), "pool+dense": round(t_dense, 3), "encode": round(t_enc, 4),
                        "query": round(t_q, 3)},
            "embeddings_per_s": 1.0 / per_item, "queries_per_s": Sq / t_q}
    return base, par


if __name__ == "__main__":
    main()
