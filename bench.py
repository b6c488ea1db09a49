#!/usr/bin/env python3
"""Benchmark of the BN254 pairing hot path on MI355X (BASELINE.json metric: optimal-ate pairings/sec, bit-exact).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of `pairing_batch` over one batch of 2^16 synthetic (r*G1, s*G2) pairs per GPU, resident in HBM
(BASELINE.json configs[1]; with N GPUs every rank owns its own 2^16 pairs: weak scaling, no data-path collective).
Rank 0 prints ONE JSON line.  Extra objects:
  roofline     the dominant kernel's algorithmic 32x32->64 MACs per launch / its HIP-event-measured duration, against the
               measured v_mad_u64_u32 peak of the chip (this path is integer-VALU bound, not HBM or MFMA: SURVEY.md 8d);
               `traffic` is the rocprofv3 HBM byte count per launch from profiles/ (null if not collected)
  cpu_baseline the reference-faithful CPU port (oracle/) timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

BATCH = 1 << 16
# algorithmic work per pairing (DESIGN.md "Work per pairing"): Fq multiplication equivalents of the reference's formulas
# with 3-mult Fq2 products and add-only beta/xi, x 136 MAC32 (8x32-bit-limb Montgomery), split by kernel
MAC32_PER_PAIRING = 2.583e6
KERNEL_SHARE = {"miller": 9919 / 18686, "final_exp": 8767 / 18686}
PEAK_TMAC32 = 30.1            # measured v_mad_u64_u32 issue peak, profiles/r01_ubench_valu_rates.txt (1024 SIMDs x 0.459 G/s x 64)
ALGO_BYTES_PER_PAIRING = 672


def _barrier(dist, dev):
    """barrier + device sync on both sides of a timed region (RCCL: tied to this rank's GPU)"""
    import torch
    torch.cuda.synchronize(dev)
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[dev.index])
        else:
            dist.barrier()
        torch.cuda.synchronize(dev)


def _max_over_ranks(dist, dev, elapsed):
    import torch
    if not dist.is_initialized():
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_baseline(batch_p, batch_q):
    """reference-faithful CPU port on all host cores, bounded sample (~20 s of CPU work)"""
    sys.path.insert(0, str(ROOT / "oracle"))
    import bn_oracle
    bn_oracle.build()
    o = bn_oracle.Oracle()
    cores = bn_oracle.usable_cpus()          # respects a cgroup CPU quota (the GPU box grants 16 of its 256 hardware threads)
    t0 = time.perf_counter(); o.pairing_batch(batch_p[:8], batch_q[:8], nthreads=1); t1 = (time.perf_counter() - t0) / 8
    n = int(min(len(batch_p), max(cores, min(4096, 20.0 / t1))))   # ~20 s of single-thread work
    t0 = time.perf_counter(); o.pairing_batch(batch_p[:n], batch_q[:n], nthreads=cores); dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "pairings/s", "cores": cores, "kind": "port",
            "sample": f"{n} pairings of the same synthetic batch on {cores} threads ({dt:.2f} s wall); 1 thread: {t1 * 1e3:.2f} ms/pairing"}


def bench_g1mul(args, eng, dev, world, rank, local_rank):
    """side metric (not the headline): 2^20 normalized G1 scalar multiplications per GPU per step"""
    import numpy as np
    import torch
    import torch.distributed as dist
    from bn_amd import distributed as D
    n = 1 << 20
    base, _ = D.synthetic_points(eng, rank * (1 << 14), (rank + 1) * (1 << 14))
    P = base.repeat(n >> 14, 1).contiguous()
    k = torch.from_numpy(D.synthetic_scalars(0, n >> 4, 1).view(np.int64)).to(dev).repeat(16, 1).contiguous()
    for _ in range(args.warmup):
        eng.g1_mul(P, k)
    _barrier(dist, dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.g1_mul(P, k)
    _barrier(dist, dev)
    elapsed = _max_over_ranks(dist, dev, time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "BN254 G1 scalar multiplications/sec (normalized output, bit-exact vs ref)", "value": world * n * args.steps / elapsed,
                          "unit": "scalar muls/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
                          "data": "synthetic", "config": {"workload": "2^20 G1 scalar muls by random Fr per GPU per step (BASELINE.json configs[4])"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def bench_product(args, eng, dev, world, rank, local_rank):
    """side metric: BASELINE.json configs[3] - multi-pairing product of 2^15 pairs per GPU -> ONE Gt; the only workload with an
    exchange step: one RCCL all-gather of 384 B per rank, then world-1 Fq12 products and a single final exponentiation"""
    import torch
    import torch.distributed as dist
    from bn_amd import distributed as D
    n = 1 << 15
    P, Q = D.synthetic_points(eng, rank * n, (rank + 1) * n)
    for _ in range(args.warmup):
        gt = D.pairing_product_sharded(eng, P, Q)
    _barrier(dist, dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gt = D.pairing_product_sharded(eng, P, Q)
    _barrier(dist, dev)
    elapsed = _max_over_ranks(dist, dev, time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "BN254 pairs/sec folded into one multi-pairing product (bit-exact vs ref)", "value": world * n * args.steps / elapsed,
                          "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "data": "synthetic",
                          "config": {"workload": f"product of {n} pairs per GPU -> 1 Gt (BASELINE.json configs[3]); all_gather of 384 B per rank"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def bench_prepared(args, eng, dev, world, rank, local_rank):
    """side metric: prepared-G2 mode - 2^16 pairings of random P against ONE precomputed Q per GPU per step"""
    import torch
    from bn_amd import distributed as D
    n = args.batch
    P, Q = D.synthetic_points(eng, rank * n, (rank + 1) * n)
    coeffs = eng.empty(102, 24)
    eng.e.g2_precompute_dev(Q.data_ptr(), coeffs.data_ptr(), 1, eng._stream())
    out = eng.empty(n, 48)
    def step():
        eng.e.miller_prepared_dev(P.data_ptr(), coeffs.data_ptr(), True, out.data_ptr(), n, eng._stream())
        eng.e.final_exp_batch_dev(out.data_ptr(), out.data_ptr(), n, eng._stream())
    for _ in range(args.warmup):
        step()
    eng.e.profile(True); eng.e.profile_reset()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if rank == 0:
        st = {k: eng.e.kernel_stats(k) for k in ("miller_prepared", "final_exp")}
        print(json.dumps({"metric": "BN254 pairings/sec against one prepared G2 point (bit-exact vs ref)", "value": world * n * args.steps / elapsed,
                          "unit": "pairings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "higher_is_better": True, "data": "synthetic", "kernel_ms": {k: v[0] / max(v[1], 1) for k, v in st.items()},
                          "config": {"workload": f"{n} random P against one precomputed Q (102 x 192 B coefficients shared by all lanes)"}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="pairings per GPU per step")
    ap.add_argument("--mapping", type=int, default=None, help="0: one lane per pairing, 1: lane pair per pairing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["pairing", "g1mul", "prepared", "product"], default="pairing",
                    help="pairing: the headline metric (default); g1mul: BASELINE.json configs[4], 2^20 G1 scalar muls (side metric)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import bn_amd
    from bn_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: bn_amd has no CPU path")
    # Test hooks for a ONE-GPU box (never set by the driver): all ranks share cuda:0 and rendezvous over gloo, which exercises
    # the N>1 control flow (shards, barriers, MAX over ranks, rank-0 line) without RCCL's one-GPU-per-rank requirement.
    share_gpu = os.environ.get("BN254_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("BN254_BENCH_BACKEND", "nccl")
    gpu_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    eng = D.TorchEngine(bn_amd.Engine(gpu_index, mapping=args.mapping), dev)
    if args.workload == "g1mul":
        return bench_g1mul(args, eng, dev, world, rank, local_rank)
    if args.workload == "prepared":
        return bench_prepared(args, eng, dev, world, rank, local_rank)
    if args.workload == "product":
        return bench_product(args, eng, dev, world, rank, local_rank)
    n = args.batch
    lo = rank * n
    P, Q = D.synthetic_points(eng, lo, lo + n)              # untimed: inputs resident in HBM before the clock starts
    out = eng.empty(n, 48)

    def sync():
        _barrier(dist, dev)

    for _ in range(args.warmup):
        D.pairing_batch_sharded(eng, P, Q, out)
    eng.e.profile(True); eng.e.profile_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        D.pairing_batch_sharded(eng, P, Q, out)
    sync()
    elapsed = time.perf_counter() - t0
    eng.e.profile(False)
    elapsed = _max_over_ranks(dist, dev, elapsed)

    if rank == 0:
        value = world * n * args.steps / elapsed
        stats = {k: eng.e.kernel_stats(k) for k in ("miller", "final_exp")}
        dom = max(stats, key=lambda k: stats[k][0])
        ms, cnt = stats[dom]
        avg_s = ms * 1e-3 / max(cnt, 1)
        achieved = n * MAC32_PER_PAIRING * KERNEL_SHARE[dom] / avg_s / 1e12
        traffic = None
        tf = ROOT / "profiles" / "pmc_traffic.json"
        if tf.exists():
            traffic = json.loads(tf.read_text()).get(dom, {}).get("hbm_bytes_per_launch")
        line = {
            "metric": "BN254 optimal-ate pairings/sec (bit-exact vs ref)", "value": value, "unit": "pairings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{n} independent pairings per GPU per step, inputs r*G1 / s*G2 (Jacobian, z != 1) resident in HBM "
                                   "(BASELINE.json configs[1])", "pairings_per_gpu": n, "parallelism": f"dp{world} (sharded, no collective)",
                       "number_system": "exact integer: 9 x 29-bit limbs in u32, v_mad_u64_u32 accumulation, Montgomery radix 2^261",
                       "mapping": 1 if args.mapping is None else args.mapping},
            "roofline": {"bound": "valu-int32-mac (neither hbm nor mfma: SURVEY.md 8d)", "kernel": dom, "achieved": achieved,
                         "peak": PEAK_TMAC32, "unit": "TMAC32/s", "frac": achieved / PEAK_TMAC32, "traffic": traffic,
                         "hbm_GBps_of_8000": None if traffic is None else traffic / avg_s / 1e9,
                         "avg_launch_ms": avg_s * 1e3, "launches": cnt,
                         "algorithmic_hbm_bytes_per_launch": n * ALGO_BYTES_PER_PAIRING,
                         "kernel_ms": {k: v[0] / max(v[1], 1) for k, v in stats.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            Pn = P[:4096].cpu().numpy().view(np.uint64); Qn = Q[:4096].cpu().numpy().view(np.uint64)
            line["cpu_baseline"] = cpu_baseline(Pn, Qn)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
