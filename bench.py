#!/usr/bin/env python3
"""Benchmark of the BN254 pairing hot path on MI355X (BASELINE.json metric: optimal-ate pairings/sec, bit-exact).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE re-launches itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of `pairing_batch` over one batch of synthetic (r*G1, s*G2) pairs resident in HBM:
  N = 1   BASELINE.json configs[1]: 2^16 independent pairings on one MI355X;
  N > 1   BASELINE.json configs[2]: 2^20 independent pairings sharded over the N GPUs (2^20/N each, contiguous index ranges, no
          data-path collective) - total work fixed, so "scaling": "strong".
Rank 0 prints ONE JSON line.  The K steps are timed with the library's event profiling OFF; the per-kernel durations behind
`roofline` come from a separate short profiled pass right after.  Extra objects:
  roofline     the dominant kernel's algorithmic 32x32->64 MACs per step / its HIP-event-measured duration, against the
               v_mad_u64_u32 issue peak of the chip MEASURED IN THIS RUN (bn254_ubench_mac32 in the untimed prologue; this path is
               integer-VALU bound, not HBM or MFMA: SURVEY.md 8d); `traffic` is the rocprofv3 HBM byte count per launch taken
               from profiles/ (a separate --pmc run, not this run - `traffic_source` says which)
  host_api     PCIe-inclusive rate of the host-buffer entry point bn254_pairing_batch on pageable numpy buffers (never `value`)
  cpu_baseline the reference-faithful CPU port (oracle/) timed on this box's host cores on a bounded sample
  side         (N = 1, default workload) the other BASELINE configs that fit one GPU, each measured in a few steps after the
               headline: configs[4] 2^20 G1 scalar multiplications, configs[3] the 2^18-pair multi-pairing and its 2^15 per-GPU
               shard, and the latency of ONE pairing - never part of `value`
Side workloads (--workload g1mul | g2mul | gtpow | prepared | product) print the same kind of line for their own metric.
"""
import argparse
import json
import os
import pathlib
import socket
import subprocess
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

BATCH = 1 << 16               # configs[1]
TOTAL_MULTI = 1 << 20         # configs[2]: total over all GPUs when N > 1
PRODUCT_TOTAL = 1 << 18       # configs[3]
# algorithmic work per unit: Fq multiplication equivalents of the reference's formulas with 3-mult Fq2
# products, 2-mult Fq2 squarings and add-only beta/xi, x 136 MAC32 (8x32-bit-limb Montgomery)
MAC32_PER_FQMUL = 136
MAC32_PER_PAIRING = 2.583e6
KERNEL_SHARE = {"miller": 9919 / 18686, "final_exp": 8767 / 18686}
ALGO_BYTES_PER_PAIRING = 672
# Side kernels: Fq-product equivalents per unit of (a) the REFERENCE's chain (groups/mod.rs:228-311 double-and-add: 255 doublings +
# 127 additions on average; fields/mod.rs:35-46: 256 Fq12 squarings + 128 products) and (b) the chain the kernel actually EXECUTES
# (GLV + Booth windows for G1, windows for G2 and Gt::pow), counted by the host simulation of the device code (fe_mul + 1.5 x
# fe_mul2 per unit; tests/test_hostsim.py::test_executed_chain_lengths keeps profiles/executed_chain_lengths.json honest).
# `roofline.frac` of a side kernel is over (b) - what the hardware is asked to do - and `frac_vs_reference_chain` over (a); (a) can
# exceed the peak when the kernel's chain is shorter than the reference's, (b) cannot.
FQMUL_REF = {"g1_mul": 3817, "g2_mul": 9541, "gt_pow": 16128}
_CHAINS = json.loads((ROOT / "profiles" / "executed_chain_lengths.json").read_text())
FQMUL_OWN = _CHAINS["fq_products_per_unit"]
# multiply instructions ONE LANE issues per unit (exact, from the host simulation) x the lanes a unit occupies = the EXECUTED multiply-adds
# behind `roofline.frac_executed`
MAC_INSTR = _CHAINS["mac_instructions_per_lane_and_unit"]
LANES_PER_UNIT = {"g1_mul": 1}                                    # every other kernel of these lines runs a unit on a lane pair
# committed one-GPU bench lines, newest first: where an N > 1 run, which measures no side workloads itself, takes the one-GPU shard times of
# its scaling prediction from (expected_scaling) - the first one that carries all four configs[3] shard sizes
SCALING_SOURCES = sorted((ROOT / "profiles").glob("r[0-9][0-9]?_bench_line.json"), reverse=True)


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


def timed_steps(dist, dev, fn, steps, warmup):
    """W untimed warm-up steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; MAX over ranks"""
    # a barrier BEFORE the warm-up as well: the first collective of a process group takes tens of milliseconds (lazy set-up), the
    # GPU idles meanwhile, and after ~5 ms of idling the next few steps run up to 20 % slower (clock ramp: 8.7, 8.1, 7.6, 7.4 ms
    # instead of 7.2) - with the set-up paid here the barrier that opens the timed region costs 30 us and nothing ramps
    _barrier(dist, dev)
    for _ in range(warmup):
        fn()
    _barrier(dist, dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    _barrier(dist, dev)
    return _max_over_ranks(dist, dev, time.perf_counter() - t0)


def kernel_times(eng, dev, fn, names, steps):
    """a SEPARATE profiled pass (event pairs around every launch): {name: (total_ms, launches)} over `steps` steps"""
    import torch
    eng.e.profile(True); eng.e.profile_reset()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    eng.e.profile(False)
    st = {k: eng.e.kernel_stats(k) for k in names}
    return {k: v for k, v in st.items() if v[1]}


class PowerSampler:
    """Socket power and shader clock of ONE GPU, sampled from the amdgpu driver while kernels run: a thread reads the hwmon files of the
    device (power1_input in microwatts, freq1_input = sclk in Hz; found through the PCI address torch reports) every ~2 ms, and the
    ROCm-SMI energy accumulator (rsmi_dev_energy_count_get, 15.3 uJ resolution) is read before and after the leg.  Everything is optional:
    `why` says what was missing when a field comes back None."""

    def __init__(self, torch, dev):
        import glob
        self.why = None
        self.power_f = self.freq_f = self.cap_w = None
        self.rsmi = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            cands = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if not cands:                                        # containers may hide the PCI tree: a single amdgpu hwmon node is unambiguous
                cands = [h for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(h + "/power1_input")]
                if len(cands) != 1:
                    cands = []
            if cands:
                h = cands[0]
                self.power_f = h + "/power1_input" if os.path.exists(h + "/power1_input") else (h + "/power1_average" if os.path.exists(h + "/power1_average") else None)
                self.freq_f = h + "/freq1_input" if os.path.exists(h + "/freq1_input") else None
                if os.path.exists(h + "/power1_cap"):
                    self.cap_w = int(open(h + "/power1_cap").read()) / 1e6
            if not self.power_f:
                self.why = f"no readable hwmon power node for {bdf}"
        except Exception as e:                                   # noqa: BLE001 - measurement is best effort, the bench line must still appear
            self.why = f"hwmon lookup failed: {e!r}"
        try:
            import ctypes as C
            l = C.CDLL("/opt/rocm/lib/librocm_smi64.so")
            if l.rsmi_init(C.c_uint64(0)) == 0:
                n = C.c_uint32()
                l.rsmi_num_monitor_devices(C.byref(n))
                want = None
                pr = torch.cuda.get_device_properties(dev)
                want = (pr.pci_domain_id << 32) | (pr.pci_bus_id << 8) | (pr.pci_device_id << 3)
                for i in range(n.value):
                    b = C.c_uint64()
                    if l.rsmi_dev_pci_id_get(i, C.byref(b)) == 0 and (b.value & 0xffffffff0000fff8) == want or n.value == 1:
                        self.rsmi = (l, C, i)
                        break
        except Exception:                                        # noqa: BLE001
            self.rsmi = None

    def _energy_uj(self):
        if not self.rsmi:
            return None
        l, C, i = self.rsmi
        e = C.c_uint64(); res = C.c_float(); ts = C.c_uint64()
        if l.rsmi_dev_energy_count_get(i, C.byref(e), C.byref(res), C.byref(ts)) != 0:
            return None
        return e.value * float(res.value)

    def run(self, torch, dev, step, units_per_step, min_seconds=1.2, est_ms_per_step=None):
        """loops `step` for at least `min_seconds` (outside any timed region), sampling; returns the fields of the roofline object"""
        import threading
        if est_ms_per_step is None:
            est_ms_per_step = 10.0
        k = max(10, int(min_seconds * 1e3 / est_ms_per_step) + 1)
        samples = []
        stop = threading.Event()

        def rd(f):
            with open(f) as fh:
                return int(fh.read())

        def loop():
            while not stop.is_set():
                t = time.perf_counter()
                try:
                    samples.append((t, rd(self.power_f) / 1e6 if self.power_f else None, rd(self.freq_f) / 1e6 if self.freq_f else None))
                except Exception:                                # noqa: BLE001
                    pass
                time.sleep(0.002)
        torch.cuda.synchronize(dev)
        th = threading.Thread(target=loop, daemon=True)
        e0 = self._energy_uj()
        t0 = time.perf_counter()
        if self.power_f or self.freq_f:
            th.start()
        for _ in range(k):
            step()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        e1 = self._energy_uj()
        stop.set()
        if th.is_alive():
            th.join()
        # the first 0.2 s are the ramp (clocks and the power average settle), the tail after the last kernel is idle
        body = [x for x in samples if t0 + 0.2 <= x[0] <= t1 - 0.005]
        pw = [x[1] for x in body if x[1] is not None]; fq = [x[2] for x in body if x[2] is not None]
        out = {"power_W": sum(pw) / len(pw) if pw else None, "power_W_max": max(pw) if pw else None, "sclk_MHz": sum(fq) / len(fq) if fq else None,
               "power_cap_W": self.cap_w, "power_samples": len(pw),
               "power_leg": f"{k} more steps of this line's workload ({(t1 - t0):.2f} s, outside the timed region), hwmon power1_input / freq1_input every ~2 ms, first 0.2 s dropped",
               "power_source": (self.power_f or self.why)}
        if e0 is not None and e1 is not None and e1 > e0:
            out["energy_uJ_per_unit"] = (e1 - e0) / (k * units_per_step)
            out["energy_source"] = "rsmi_dev_energy_count_get before / after the leg (whole socket, idle share included)"
            out["power_W_from_energy_counter"] = (e1 - e0) / 1e6 / (t1 - t0)
        elif pw:
            out["energy_uJ_per_unit"] = (sum(pw) / len(pw)) * (t1 - t0) / (k * units_per_step) * 1e6
            out["energy_source"] = "mean sampled power x wall time of the leg"
        else:
            out["energy_uJ_per_unit"] = None
        out["units_per_s_during_leg"] = k * units_per_step / (t1 - t0)
        return out


_PEAK = {}


def measured_peak(eng):
    """same-run ceiling: T lane-MAC32/s of a pure v_mad_u64_u32 stream at 8 waves/SIMD on random 32-bit operands (the chip's issue peak:
    `roofline.peak`) and at the 2 waves/SIMD the pairing kernels run at (256 VGPRs each) on random 29-BIT operands - the engine's own
    limbs, on which the instruction is ~3 % faster (`peak_at_kernel_occupancy`: the like-for-like ceiling)"""
    if "v" not in _PEAK:
        best8 = max(eng.e.ubench_mac32(8, 1 << 14)[0] for _ in range(3))
        best2 = max(eng.e.ubench_mac32(2, 1 << 14, operand_bits=29)[0] for _ in range(3))
        _PEAK["v"] = (best8 / 1e3, best2 / 1e3)
    return _PEAK["v"]


_TRAFFIC = None


def traffic_of(kernel, units):
    """HBM traffic of `units` units through `kernel` (a name of bn254_kernel_stats), from the rocprofv3 --pmc passes summarised in
    profiles/pmc_traffic.json (tools/summarize_pmc_all.py: (2 x FETCH_SIZE + WRITE_SIZE) per launch / units of that launch - a SEPARATE
    run, never this one), beside the algorithmic bytes (the reference's structs in and out) and their ratio; None if never measured"""
    global _TRAFFIC
    if _TRAFFIC is None:
        tf = ROOT / "profiles" / "pmc_traffic.json"
        _TRAFFIC = json.loads(tf.read_text()) if tf.exists() else {}
    e = _TRAFFIC.get(kernel)
    if not e or "hbm_bytes_per_unit" not in e:
        return None
    return {"traffic": e["hbm_bytes_per_unit"] * units, "algorithmic_bytes": e["algorithmic_bytes_per_unit"] * units,
            "traffic_ratio": e["hbm_bytes_per_unit"] / e["algorithmic_bytes_per_unit"], "traffic_measured_at_units_per_launch": e["units_per_launch"],
            "traffic_source": e["source"] + " (separate rocprofv3 --pmc run, NOT this run; per unit x the units of this line's launches)"}


def roofline(eng, stats, unit_count, mac32_per_unit, shares=None, traffic_key=None, steps=1, ref_mac32_per_unit=None):
    """roofline object for the dominant kernel among `stats` = {name: (total_ms, launches)}.  Each kernel processed `unit_count` units
    per step (a step may be several sub-launches of one machine round: bn_sub_launch in csrc/bn254_hip.hip) over `steps` recorded
    steps: achieved = units x steps x algorithmic MAC32 per unit / total kernel time."""
    peak8, peak2 = measured_peak(eng)
    dom = max(stats, key=lambda k: stats[k][0])
    per = {}
    for k, (ms, cnt) in stats.items():
        share = shares.get(k, 1.0) if shares else 1.0
        ach = unit_count * steps * mac32_per_unit * share / (ms * 1e-3) / 1e12
        per[k] = {"avg_launch_ms": ms / cnt, "launches": cnt, "ms_per_step": ms / steps, "achieved": ach, "frac": ach / peak8}
        mk = {"final_exp_quad": None, "miller_quad": None, "pairing_wave": None, "final_exp_wave": None}.get(k, k)     # (other mappings execute other chains)
        if mk in MAC_INSTR:
            ex = unit_count * steps * MAC_INSTR[mk] * LANES_PER_UNIT.get(mk, 2) / (ms * 1e-3) / 1e12
            per[k].update({"executed_TMAC_per_s": ex, "frac_executed": ex / peak8, "frac_executed_of_occupancy_peak": ex / peak2})
        t = traffic_of(k, unit_count * steps / cnt)                       # EVERY kernel of the line carries its traffic story, per launch
        if t:
            per[k].update({"traffic": t["traffic"], "algorithmic_bytes": t["algorithmic_bytes"], "traffic_ratio": t["traffic_ratio"],
                           "hbm_GBps": t["traffic"] / (ms / cnt * 1e-3) / 1e9})
    td = traffic_of(dom, unit_count * steps / stats[dom][1])
    traffic, src = (td["traffic"], td["traffic_source"]) if td else (None, None)
    d = per[dom]
    out = {"bound": "valu-int32-mac (neither hbm nor mfma: SURVEY.md 8d); see power_W / power_cap_W / sclk_MHz of this object where the line carries them: the socket's power cap is what limits the multiplier",
           "kernel": dom, "achieved": d["achieved"], "peak": peak8,
           "unit": "TMAC32/s", "frac": d["frac"],
           "achieved_is": "Fq products of the chain x 136 MAC32 (an 8 x 32-bit-limb Montgomery product) / measured kernel time",
           "frac_executed": d.get("frac_executed"), "executed_TMAC_per_s": d.get("executed_TMAC_per_s"),
           "frac_executed_of_occupancy_peak": d.get("frac_executed_of_occupancy_peak"),
           "frac_executed_is": "multiply instructions the kernel EXECUTES (v_mad_u64_u32 / v_mad_i64_i32 / v_mul_lo / v_mul_hi per lane, counted by the host simulation of the device code: "
                               "profiles/executed_chain_lengths.json) x lanes / measured kernel time / peak - how full the multiplier is, where `frac` prices the reference's formulas",
           "peak_source": "bn254_ubench_mac32 in this run: pure v_mad_u64_u32 stream, 8 waves/SIMD",
           "peak_at_kernel_occupancy": peak2, "kernel_occupancy_waves_per_simd": 2, "frac_of_occupancy_peak": d["achieved"] / peak2,
           "peak_at_kernel_occupancy_is": "the same stream at 2 waves/SIMD on random 29-bit operands (the engine's limbs)",
           "traffic": traffic, "traffic_source": src, "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"], "kernels": per}
    sq = ROOT / "profiles" / "sq_counters.json"
    if sq.exists():                          # the SQ counters of a separate session: VALU instructions per wave, issue interval per SIMD
        sqd = json.loads(sq.read_text())
        for k in per:
            if k in sqd:
                per[k]["valu_issue"] = sqd[k]
        if dom in sqd:
            out["valu_issue"] = sqd[dom]
    if all("traffic" in v for v in per.values()):
        # the whole step: what all kernels of the line move per step against the algorithmic bytes of the step (inputs in, results out once)
        out["traffic_per_step_all_kernels"] = sum(v["traffic"] * v["launches"] / steps for v in per.values())
    if ref_mac32_per_unit:
        out["frac_vs_reference_chain"] = d["frac"] * ref_mac32_per_unit / mac32_per_unit
        out["frac_is"] = ("over the Fq products of the chain this kernel EXECUTES (profiles/executed_chain_lengths.json); frac_vs_reference_chain "
                          "counts the reference's longer double-and-add / square-and-multiply chain instead and may exceed 1")
    return out


def cpu_baseline(batch_p, batch_q):
    """reference-faithful CPU port on all host cores, bounded sample (~20 s of CPU work), and beside it the same schedule with the
    CPU's native 64 x 64 -> 128 multiply (`native128`): the reference builds every 64-bit MAC from 32-bit halves
    (src/arith.rs:441-478), which handicaps it on a CPU that has the wide multiplier - both figures are the same 36 938-product chain"""
    sys.path.insert(0, str(ROOT / "oracle"))
    import bn_oracle
    import numpy as np
    bn_oracle.build()
    cores = bn_oracle.usable_cpus()          # respects a cgroup CPU quota (the GPU box grants 16 of its 256 hardware threads)

    def run(o, share):
        t0 = time.perf_counter(); ref = o.pairing_batch(batch_p[:8], batch_q[:8], nthreads=1); t1 = (time.perf_counter() - t0) / 8
        n = int(min(len(batch_p), max(cores, min(4096, share / t1))))   # ~`share` seconds of single-thread work
        t0 = time.perf_counter(); o.pairing_batch(batch_p[:n], batch_q[:n], nthreads=cores); dt = time.perf_counter() - t0
        return n, dt, t1, ref
    n, dt, t1, ref = run(bn_oracle.Oracle(), 20.0)
    out = {"value": n / dt, "unit": "pairings/s", "cores": cores, "kind": "port", "one_thread_ms_per_pairing": t1 * 1e3,
           "sample": f"{n} pairings of the same synthetic batch on {cores} threads ({dt:.2f} s wall); 1 thread: {t1 * 1e3:.2f} ms/pairing"}
    n2, dt2, t2, ref2 = run(bn_oracle.Oracle(native128=True), 10.0)
    out["native128"] = {"value": n2 / dt2, "unit": "pairings/s", "cores": cores, "one_thread_ms_per_pairing": t2 * 1e3,
                        "same_bytes_as_port": bool(np.array_equal(ref, ref2)),
                        "what": "the same chain built with -DBNO_NATIVE128 (unsigned __int128 MACs instead of the reference's 32-bit halves): "
                                f"{n2} pairings on {cores} threads ({dt2:.2f} s wall)"}
    return out


def host_api_rate(gpu_index, Pn, Qn, reps=5):
    """what a binding of the reference's `pairing` gets: pageable host buffers in and out through bn254_pairing_batch (chunked,
    copies of one chunk overlapped with the kernels of the other); and the latency of ONE by-value pairing(p, q) (lib.rs:181-183)"""
    import bn_amd
    import numpy as np
    e = bn_amd.Engine(gpu_index)
    n = Pn.shape[0]
    out = np.zeros((n, 48), np.uint64)                       # the caller's result buffer, reused (pages already resident)
    e.pairing_batch(Pn, Qn, out)                             # allocates the device staging
    t0 = time.perf_counter()
    for _ in range(reps):
        e.pairing_batch(Pn, Qn, out)
    dt = (time.perf_counter() - t0) / reps
    e.pairing_batch(Pn[:1], Qn[:1], out[:1])
    t0 = time.perf_counter()
    for _ in range(20):
        e.pairing_batch(Pn[:1], Qn[:1], out[:1])
    one = (time.perf_counter() - t0) / 20
    e.close()
    return {"value": n / dt, "unit": "pairings/s", "ms_per_call": dt * 1e3, "single_pairing_ms": one * 1e3,
            "what": f"bn254_pairing_batch, {n} pairings per call, pageable numpy buffers: H2D {n * 288 / 1e6:.1f} MB + kernels + D2H {n * 384 / 1e6:.1f} MB; single_pairing_ms: the same call with n = 1"}


def _line(metric, unit, value, world, args, elapsed, scaling, workload, extra=None, cfg=None):
    d = {"metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
         "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
         "dtype": "u32", "data": "synthetic", "config": dict({"workload": workload}, **(cfg or {}))}
    d.update(extra or {})
    return d


# ------------------------------------------------------------------------------------------------------------ side workloads
def run_mul(eng, dev, dist, which, n, lo, steps, warmup):
    """n normalized scalar multiplications of DISTINCT random points by distinct random Fr (benches/api.rs:107-111)"""
    from bn_amd import distributed as D
    P, Q = D.synthetic_points(eng, lo, lo + n)                              # distinct points r_i*G, Jacobian z != 1
    pts = P if which == 1 else Q
    k = D.synthetic_scalars_device(eng, (1 << 24) + lo, (1 << 24) + lo + n, 1)    # distinct scalars, another index range
    fn = eng.g1_mul if which == 1 else eng.g2_mul
    name = "g1_mul" if which == 1 else "g2_mul"
    step = lambda: fn(pts, k)
    elapsed = timed_steps(dist, dev, step, steps, warmup)
    ks = min(steps, 5)
    st = kernel_times(eng, dev, step, (name,), ks)
    rf = roofline(eng, st, n, FQMUL_OWN[name] * MAC32_PER_FQMUL, steps=ks, ref_mac32_per_unit=FQMUL_REF[name] * MAC32_PER_FQMUL)
    return elapsed, rf


def bench_mul(args, eng, dev, world, rank, which):
    """side metrics: BASELINE.json configs[4] - 2^20 normalized G1 scalar multiplications on one GPU - and the same for G2 (2^18)"""
    import torch.distributed as dist
    n = (1 << 20) if which == 1 else (1 << 18)
    elapsed, rf = run_mul(eng, dev, dist, which, n, rank * n, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(_line(f"BN254 G{which} scalar multiplications/sec (normalized output, bit-exact vs ref)", "scalar muls/s",
                               world * n * args.steps / elapsed, world, args, elapsed, "weak",
                               f"{n} G{which} scalar muls of distinct random points by distinct random Fr per GPU per step"
                               + (" (BASELINE.json configs[4])" if which == 1 else ""), {"roofline": rf})), flush=True)


def bench_gtpow(args, eng, dev, world, rank):
    """side metric: Gt::pow (lib.rs:171) on 2^16 pairing values with distinct random exponents"""
    import torch.distributed as dist
    from bn_amd import distributed as D
    n = 1 << 16
    lo = rank * n
    P, Q = D.synthetic_points(eng, lo, lo + n)
    g = eng.pairing_batch(P, Q)
    k = D.synthetic_scalars_device(eng, (1 << 24) + lo, (1 << 24) + lo + n, 0)
    step = lambda: eng.gt_pow(g, k)
    elapsed = timed_steps(dist, dev, step, args.steps, args.warmup)
    if rank == 0:
        ks = min(args.steps, 5)
        st = kernel_times(eng, dev, step, ("gt_pow",), ks)
        rf = roofline(eng, st, n, FQMUL_OWN["gt_pow"] * MAC32_PER_FQMUL, steps=ks, ref_mac32_per_unit=FQMUL_REF["gt_pow"] * MAC32_PER_FQMUL)
        print(json.dumps(_line("BN254 Gt::pow/sec (bit-exact vs ref)", "pows/s", world * n * args.steps / elapsed, world, args, elapsed,
                               "weak", f"{n} Gt values ^ distinct random Fr per GPU per step", {"roofline": rf})), flush=True)


PRODUCT_KERNELS = ("miller", "miller_shared", "miller_wave", "miller_quad", "gt_product", "gt_tail", "final_exp_wave", "final_exp_quad", "final_exp")


def run_product(eng, dev, dist, P, Q, steps, warmup):
    from bn_amd import distributed as D
    step = lambda: D.pairing_product_sharded(eng, P, Q)
    elapsed = timed_steps(dist, dev, step, steps, warmup)
    ks = min(steps, 3)
    st = kernel_times(eng, dev, step, PRODUCT_KERNELS, ks)
    n = P.shape[0]
    tr = {}
    for k, (ms, cnt) in st.items():                     # the Miller kernels (units = pairs) carry the traffic of a product; tree and tail are < 2 %
        t = traffic_of(k, n) if k.startswith("miller") else None
        if t:
            tr[k] = {"traffic_per_step": t["traffic"], "algorithmic_bytes_per_step": t["algorithmic_bytes"], "traffic_ratio": t["traffic_ratio"],
                     "hbm_GBps": t["traffic"] / (ms / ks * 1e-3) / 1e9}
    return elapsed, {k: v[0] / ks for k, v in st.items()}, tr


def run_product_prepared(eng, dev, dist, P, Q, steps, warmup, one_q=False, power=False):
    """the multi-pairing product of the pairs (P[i], Q[i]) with every Q[i] prepared natively beforehand (not timed: a verifier's G2 points are fixed);
    one_q: every P against ONE prepared point (the table then comes out of the caches: what streaming the tables costs)"""
    n = P.shape[0]
    prep = eng.e.g2_prepare_dev(Q.data_ptr(), 1 if one_q else n, eng._stream())
    part = eng.empty(1, 48)

    def step():
        eng.e.miller_product_prepared_native_dev(P.data_ptr(), prep, n, part.data_ptr(), stream=eng._stream())
        eng.e.final_exp_batch_dev(part.data_ptr(), part.data_ptr(), 1, eng._stream())
    elapsed = timed_steps(dist, dev, step, steps, warmup)
    ks = min(steps, 3)
    st = kernel_times(eng, dev, step, ("miller_native", "miller_native_shared") + PRODUCT_KERNELS, ks)
    kms = {k: v[0] / ks for k, v in st.items()}
    if "miller_native_shared" in st:
        # the Miller kernel priced over the chain it EXECUTES per pair (four / two pairs per accumulator: profiles/executed_chain_lengths.json)
        m = 4 if n >= 4 * eng.e.get_option("round_pairs") else 2
        key = f"miller_native_shared{m}"
        rf = roofline(eng, {key: st["miller_native_shared"]}, n, FQMUL_OWN[key] * MAC32_PER_FQMUL, steps=ks, ref_mac32_per_unit=11952 * MAC32_PER_FQMUL)
        kk = rf["kernels"][key]
        kms["roofline"] = dict({k: rf[k] for k in ("kernel", "achieved", "peak", "frac", "frac_executed", "frac_executed_of_occupancy_peak", "frac_vs_reference_chain", "avg_launch_ms", "traffic", "traffic_source")},
                               **{k: kk.get(k) for k in ("algorithmic_bytes", "traffic_ratio", "hbm_GBps")},
                               pairs_per_accumulator=m, table_GBps=(0 if one_q else n * 33792 / (kk["ms_per_step"] * 1e-3) / 1e9))
    if power:
        pw = PowerSampler(eng.torch, dev).run(eng.torch, dev, step, n, min_seconds=1.2, est_ms_per_step=elapsed / steps * 1e3)
        kms["power"] = {k: pw.get(k) for k in ("power_W", "sclk_MHz", "power_cap_W", "energy_uJ_per_unit", "units_per_s_during_leg")}
    prep.close()
    return elapsed, kms


def bench_product_prepared(args, eng, dev, world, rank):
    """side metric: configs[3] with the G2 side prepared natively, --batch pairs (default 2^18) per GPU -> one Gt per GPU (no exchange timed)"""
    import torch.distributed as dist
    from bn_amd import distributed as D
    n = args.batch or PRODUCT_TOTAL
    P, Q = D.synthetic_points(eng, rank * n, (rank + 1) * n)
    elapsed, kms = run_product_prepared(eng, dev, dist, P, Q, args.steps, args.warmup, one_q=args.prepared_mode == "native", power=True)
    if rank == 0:
        print(json.dumps(_line("BN254 pairs/sec folded into one multi-pairing product over prepared G2 points (bit-exact vs ref)", "pairs/s", world * n * args.steps / elapsed,
                               world, args, elapsed, "weak", f"product of {n} pairs -> 1 Gt per GPU, " + ("ONE native table for all pairs" if args.prepared_mode == "native" else "one native table (33.8 KB) per pair"), {"kernel_ms_per_step": kms})), flush=True)


def bench_product(args, eng, dev, world, rank):
    """side metric: BASELINE.json configs[3] - multi-pairing product of 2^18 pairs -> ONE Gt, sharded 2^18/N per GPU; the only
    workload with an exchange step: one RCCL all-gather of 384 B per rank, then world-1 Fq12 products and a single final
    exponentiation (one wave-cooperative launch)"""
    import torch.distributed as dist
    from bn_amd import distributed as D
    lo, hi = D.shard_range(PRODUCT_TOTAL, rank, world)
    P, Q = D.synthetic_points(eng, lo, hi)
    elapsed, kms, ktr = run_product(eng, dev, dist, P, Q, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(_line("BN254 pairs/sec folded into one multi-pairing product (bit-exact vs ref)", "pairs/s",
                               PRODUCT_TOTAL * args.steps / elapsed, world, args, elapsed, "strong",
                               f"product of 2^18 pairs -> 1 Gt, {hi - lo} pairs per GPU (BASELINE.json configs[3]); all_gather of 384 B per rank",
                               {"kernel_ms_per_step": kms, "kernel_traffic": ktr, "expected_scaling": expected_scaling("product", world, scaling_inputs())}, {"process_group": (dist.get_backend() if dist.is_initialized() else None)})), flush=True)


def run_prepared(eng, dev, dist, P, Q, n, mode, steps, warmup):
    """n pairings of the P against ONE prepared Q (the first of Q).  mode "native": the device-native table of bn254_g2_prepare (88 lines of
    the engine's own schedule as ready multiplier operands, 33.8 KB per Q); "reference": the reference image of G2Precomp (bn_ell_coeffs,
    102 x 192 B: groups/mod.rs:472-483) converted by every lane at every line - kept for the reference's known answers.  Returns the wall
    time of `steps` steps, the roofline object of the Miller kernel and the per-kernel milliseconds."""
    out = eng.empty(n, 48)
    if mode in ("native", "native_per_q"):
        # native_per_q: one table PER PAIRING (n distinct Q, 33.8 KB each: 2.2 GB at 2^16) - the flavour in which the tables are real HBM traffic
        prep = eng.e.g2_prepare_dev(Q.data_ptr(), n if mode == "native_per_q" else 1, eng._stream())
        step = lambda: eng.e.pairing_prepared_native_dev(P.data_ptr(), prep, out.data_ptr(), n, stream=eng._stream())
        kname = "miller_native"
    else:
        coeffs = eng.empty(102, 24)
        eng.e.g2_precompute_dev(Q.data_ptr(), coeffs.data_ptr(), 1, eng._stream())

        def step():
            eng.e.miller_prepared_dev(P.data_ptr(), coeffs.data_ptr(), True, out.data_ptr(), n, eng._stream())
            eng.e.final_exp_batch_dev(out.data_ptr(), out.data_ptr(), n, eng._stream())
        kname = "miller_prepared"
    elapsed = timed_steps(dist, dev, step, steps, warmup)
    ks = min(steps, 5)
    st = kernel_times(eng, dev, step, (kname, "final_exp"), ks)
    # the Miller kernel priced over the chain it EXECUTES (host-simulation count, profiles/executed_chain_lengths.json) - the convention of
    # the side kernels; `frac_vs_reference_chain`: over the reference's miller_loop (groups/mod.rs:486-519: 11 952 multiplications as written)
    rf = roofline(eng, {kname: st[kname]}, n, FQMUL_OWN[kname] * MAC32_PER_FQMUL, steps=ks, ref_mac32_per_unit=11952 * MAC32_PER_FQMUL)
    if mode == "native_per_q":                      # the same kernel on another workload: its traffic entry is keyed separately (tools/summarize_pmc_all.py)
        t = traffic_of("miller_native_per_q", n)
        k = rf["kernels"][kname]
        for f in ("traffic", "algorithmic_bytes", "traffic_ratio", "hbm_GBps"):
            k.pop(f, None)
        rf["traffic"] = rf["traffic_source"] = None
        if t:
            k.update({"traffic": t["traffic"], "algorithmic_bytes": t["algorithmic_bytes"], "traffic_ratio": t["traffic_ratio"], "hbm_GBps": t["traffic"] / (k["avg_launch_ms"] * 1e-3) / 1e9})
            rf["traffic"], rf["traffic_source"] = t["traffic"], t["traffic_source"]
        k["table_bytes_read_per_launch"] = n * 33792
        k["table_GBps"] = n * 33792 / (k["avg_launch_ms"] * 1e-3) / 1e9
    if mode.startswith("native"):
        prep.close()
    return elapsed, rf, {k: v[0] / max(v[1], 1) for k, v in st.items()}


def bench_prepared(args, eng, dev, world, rank):
    """side metric: prepared-G2 mode - 2^16 pairings of random P against ONE prepared Q per GPU per step"""
    import torch.distributed as dist
    from bn_amd import distributed as D
    n = args.batch or BATCH
    P, Q = D.synthetic_points(eng, rank * n, (rank + 1) * n)
    mode = args.prepared_mode or "native"
    elapsed, rf, kms = run_prepared(eng, dev, dist, P, Q, n, mode, args.steps, args.warmup)
    if rank == 0:
        what = {"native": "against ONE prepared Q: one device-native table (88 lines x 384 B, bn254_g2_prepare) read by all lanes",
                "native_per_q": f"against {n} prepared Q, one device-native table each (33.8 KB per pairing streamed from HBM)",
                "reference": "against ONE prepared Q: 102 x 192 B reference-image coefficients (bn254_g2_precompute) shared by all lanes"}[mode]
        print(json.dumps(_line("BN254 pairings/sec against prepared G2 points (bit-exact vs ref)", "pairings/s", world * n * args.steps / elapsed,
                               world, args, elapsed, "weak", f"{n} random P {what}",
                               {"roofline": rf, "kernel_ms": kms}, {"prepared_mode": mode})), flush=True)


def side_object(eng, dev, dist, P16, Q16):
    """the BASELINE configs besides the headline that fit one GPU, a few steps each (never part of `value`)"""
    from bn_amd import distributed as D
    side = {}
    n = 1 << 20
    elapsed, rf = run_mul(eng, dev, dist, 1, n, 0, 6, 2)             # (3 steps after 1 warm-up read 4 % low: the clock is still ramping)
    side["g1mul_2_20"] = {"config": "BASELINE.json configs[4]: 2^20 G1 scalar muls by random Fr, 1 MI355X", "value": n * 6 / elapsed, "unit": "scalar muls/s",
                          "ms_per_step": elapsed / 6 * 1e3, "roofline": dict({k: rf[k] for k in ("kernel", "achieved", "peak", "frac", "frac_vs_reference_chain", "avg_launch_ms", "traffic", "traffic_source")},
                                           **{k: rf["kernels"]["g1_mul"].get(k) for k in ("algorithmic_bytes", "traffic_ratio", "hbm_GBps")})}
    # prepared-G2 mode: 2^16 P against ONE natively prepared Q - the second roofline point of the path (the table is the one
    # operand every lane streams: 33.8 KB per Q, read 2^16 times per step out of the caches)
    elapsed, prf, kms = run_prepared(eng, dev, dist, P16, Q16, BATCH, "native", 6, 2)
    side["prepared_2_16"] = {"config": "SURVEY 8f-2: 2^16 random P against ONE prepared G2 point (bn254_g2_prepare: device-native table, 88 lines x 384 B), 1 MI355X",
                             "value": BATCH * 6 / elapsed, "unit": "pairings/s", "ms_per_step": elapsed / 6 * 1e3, "kernel_ms": kms,
                             "roofline": dict({k: prf[k] for k in ("kernel", "achieved", "peak", "frac", "frac_executed", "frac_executed_of_occupancy_peak", "frac_vs_reference_chain",
                                                                   "avg_launch_ms", "traffic", "traffic_source")},
                                              **{k: prf["kernels"]["miller_native"].get(k) for k in ("algorithmic_bytes", "traffic_ratio", "hbm_GBps")},
                                              frac_is="over the chain miller_native EXECUTES (7000 Fq-product equivalents per pairing: profiles/executed_chain_lengths.json); "
                                                      "frac_vs_reference_chain over the reference's miller_loop (11 952 multiplications, groups/mod.rs:486-519)",
                                              table_bytes_per_q=33792)}
    el_q, prf_q, kms_q = run_prepared(eng, dev, dist, P16, Q16, BATCH, "native_per_q", 4, 1)
    kq = prf_q["kernels"]["miller_native"]
    side["prepared_2_16"]["one_table_per_pairing"] = {"value": BATCH * 4 / el_q, "unit": "pairings/s", "ms_per_step": el_q / 4 * 1e3, "kernel_ms": kms_q,
                                                      "what": "2^16 distinct prepared Q, p[i] against table i: 33.8 KB per pairing streamed from HBM (2.2 GB per launch)",
                                                      **{k: kq.get(k) for k in ("frac", "frac_executed", "table_GBps", "traffic", "algorithmic_bytes", "traffic_ratio", "hbm_GBps")}}
    _, _, kms_ref = run_prepared(eng, dev, dist, P16, Q16, BATCH, "reference", 3, 1)
    side["prepared_2_16"]["reference_image_kernel_ms"] = dict(kms_ref, what="the same step over the reference-image coefficients (bn254_g2_precompute, 102 x 192 B): the mode kept for the reference's known answers")
    P, Q = D.synthetic_points(eng, 0, PRODUCT_TOTAL)
    for tag, m, what in (("product_2_18", PRODUCT_TOTAL, "BASELINE.json configs[3] on ONE GPU: multi-pairing product of 2^18 pairs -> 1 Gt"),
                         ("product_2_17", PRODUCT_TOTAL // 2, "the per-GPU shard of configs[3] at 2 GPUs: 2^17 pairs -> 1 Gt"),
                         ("product_2_16", PRODUCT_TOTAL // 4, "the per-GPU shard of configs[3] at 4 GPUs: 2^16 pairs -> 1 Gt"),
                         ("product_2_15", PRODUCT_TOTAL // 8, "the per-GPU shard of configs[3] at 8 GPUs: 2^15 pairs -> 1 Gt (Miller loops, one-launch product tree, one final exponentiation)")):
        elapsed, kms, ktr = run_product(eng, dev, dist, P[:m], Q[:m], 3, 1)
        side[tag] = {"config": what, "value": m * 3 / elapsed, "unit": "pairs/s", "ms_per_step": elapsed / 3 * 1e3, "kernel_ms_per_step": kms, "kernel_traffic": ktr}
    # configs[3] with the G2 side PREPARED (what a verifier with fixed G2 points evaluates): one native table per pair, four pairs per accumulator
    elapsed, kms = run_product_prepared(eng, dev, dist, P, Q, 3, 1)
    side["product_prepared_2_18"] = {"config": "BASELINE.json configs[3] on ONE GPU over 2^18 natively prepared G2 points (bn254_miller_product_prepared_native_dev): 33.8 KB of table per pair "
                                               "streamed from HBM, four pairs share one Miller accumulator, one final exponentiation", "value": PRODUCT_TOTAL * 3 / elapsed, "unit": "pairs/s",
                                     "ms_per_step": elapsed / 3 * 1e3, "kernel_ms_per_step": {k: v for k, v in kms.items() if k != "roofline"}, "roofline": kms.get("roofline"),
                                     "vs_fused_product": side["product_2_18"]["ms_per_step"] / (elapsed / 3 * 1e3)}
    out1 = eng.empty(1, 48)
    step = lambda: eng.pairing_batch(P16[:1], Q16[:1], out1)
    elapsed = timed_steps(dist, dev, step, 20, 3)
    side["single_pairing"] = {"config": "BASELINE.json configs[0] on the GPU: ONE pairing, device-resident (the reference's by-value pairing(p, q))",
                              "latency_ms": elapsed / 20 * 1e3,
                              "kernel_ms": {k: v[0] / 5 for k, v in kernel_times(eng, dev, step, ("pairing_wave", "miller", "final_exp_wave", "final_exp"), 5).items()}}
    return side


def scaling_inputs(line=None):
    """the ONE-GPU measurements the scaling prediction is made of: the headline's milliseconds per round of 2^16 pairings and the whole-step
    times of the configs[3] shards (2^18 / 2^17 / 2^16 / 2^15 pairs -> 1 Gt on one GPU).  From `line` (a bench line of THIS run that carries
    the side object) or else from the latest committed one-GPU line (SCALING_SOURCE); None when neither has them."""
    keys = ("product_2_18", "product_2_17", "product_2_16", "product_2_15")
    cands = [("this run", line)] if line is not None and "side" in line else []
    for f in SCALING_SOURCES:
        try:
            cands.append((str(f.relative_to(ROOT)), json.loads(f.read_text().splitlines()[0])))
        except (ValueError, IndexError):
            continue
    for src, ln in cands:
        side = ln.get("side", {})
        if all(k in side for k in keys) and ln.get("n_gpus") == 1:
            return {"round_ms": ln["ms_per_step"] * BATCH / ln["config"]["pairings_per_gpu"], "product_ms": {k: side[k]["ms_per_step"] for k in keys}, "source": src}
    return None


def expected_scaling(workload, world, inputs):
    """The prediction the first multi-GPU run tests, made ONLY of one-GPU measurements (`inputs` = scaling_inputs()):
    nothing here was measured on more than one GPU and nothing is a literal.  configs[2] (independent pairings) has no exchange: a per-GPU
    shard of 2^20/N is 16/N machine rounds of 2^16 pairings at the one-GPU rate - linear.  configs[3] (one product) is NOT: the per-GPU shard
    (2^18/N pairs -> one partial Fq12, measured as a whole step on one GPU: Miller loops + product tree + the single final exponentiation)
    shrinks below what fills the machine, and the tree, the 384-byte all-gather and the final exponentiation do not shrink at all."""
    if inputs is None:
        return None
    if workload == "pairing":
        ms = (TOTAL_MULTI // world) / BATCH * inputs["round_ms"]
        return {"ms_per_step": ms, "speedup_vs_1_gpu": float(world), "model": f"{TOTAL_MULTI // world // BATCH} rounds of 2^16 pairings x {inputs['round_ms']:.3f} ms (one-GPU measurement), no exchange: linear",
                "inputs_from": inputs["source"], "measured_on": "1 GPU only"}
    if workload == "product":
        key = {1: "product_2_18", 2: "product_2_17", 4: "product_2_16", 8: "product_2_15"}.get(world)
        if not key:
            return None
        allgather_ms = 0.05 if world > 1 else 0.0          # an ESTIMATE (3 KiB over xGMI: latency only); the one term not measured on one GPU
        ms = inputs["product_ms"][key] + allgather_ms
        return {"ms_per_step": ms, "speedup_vs_1_gpu": inputs["product_ms"]["product_2_18"] / ms,
                "model": f"one-GPU step of the 2^18/{world}-pair shard ({key}: {inputs['product_ms'][key]:.3f} ms, incl. product tree and the one final exponentiation) + all-gather {allgather_ms} ms (estimate)"
                         f" against {inputs['product_ms']['product_2_18']:.3f} ms for 2^18 pairs on one GPU",
                "inputs_from": inputs["source"], "measured_on": "1 GPU only"}
    return None


def bench_multi_c(args):
    """--mode multi_c: the SAME workloads driven from ONE host process through the C ABI's multi-device entry points
    (bn254_pairing_batch_multi / bn254_pairing_product_multi: one context, host thread and stream set per GPU, host buffers in and
    out) - what a Rust / C++ host of north_star links.  PCIe-inclusive by construction (the entry points take host memory), so this
    line is a different measurement from the default mode's HBM-resident `value` and says so."""
    import numpy as np
    import torch
    import bn_amd
    from bn_amd import distributed as D
    if int(os.environ.get("LOCAL_RANK", "0")) != 0:
        return                                                   # launched under torchrun: one process drives every GPU, the others have nothing to do
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: bn_amd has no CPU path")
    n_gpus = args.gpus
    share = os.environ.get("BN254_BENCH_SHARE_GPU") == "1"      # one-GPU box: every rank on device 0 (peer exchange)
    devices = [0] * n_gpus if share else list(range(n_gpus))
    dev = torch.device("cuda", 0)
    eng = D.TorchEngine(bn_amd.Engine(0), dev)
    product = args.workload == "product"
    total = PRODUCT_TOTAL if product else (args.batch * n_gpus if args.batch else (BATCH if n_gpus == 1 else TOTAL_MULTI))
    P, Q = D.synthetic_points(eng, 0, total)
    Pn = P.cpu().numpy().view(np.uint64); Qn = Q.cpu().numpy().view(np.uint64)
    del P, Q
    m = bn_amd.MultiEngine(devices)
    out = np.zeros((total, 48), np.uint64)                       # the caller's result buffer, pages resident
    step = (lambda: m.pairing_product(Pn, Qn)) if product else (lambda: m.pairing_batch(Pn, Qn, out))
    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    exp = expected_scaling("product" if product else "pairing", n_gpus, scaling_inputs()) if total in (TOTAL_MULTI, PRODUCT_TOTAL) else None
    line = _line(("BN254 pairs/sec folded into one multi-pairing product" if product else "BN254 optimal-ate pairings/sec") + " (bit-exact vs ref)",
                 "pairs/s" if product else "pairings/s", total * args.steps / elapsed, n_gpus, args, elapsed, "strong" if n_gpus > 1 else "weak",
                 (f"product of 2^18 pairs -> 1 Gt (BASELINE.json configs[3])" if product else f"{total} independent pairings per step (BASELINE.json configs[{1 if n_gpus == 1 else 2}])")
                 + f" over {n_gpus} GPU(s) from ONE host process: bn254_{'pairing_product' if product else 'pairing_batch'}_multi, pageable host buffers in and out (PCIe INCLUSIVE)",
                 {"mode": "multi_c", "pcie_inclusive": True, "expected_scaling_kernels_only": exp},
                 {"devices": devices, "exchange": m.exchange, "rank_numa_nodes": m.numa_nodes, "host_threads": n_gpus,
                  "bytes_per_step": {"h2d": total * 288, "d2h": 384 if product else total * 384}})
    m.close()
    print(json.dumps(line), flush=True)


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a rendezvous environment: start N ranks of this script on one node"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(pathlib.Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="pairings per GPU per step (default: 2^16 at N = 1, 2^20/N at N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="skip the sampled power leg (1.2 s more of the headline step after the timed region)")
    ap.add_argument("--no-side", action="store_true", help="skip the `side` object (configs[3], configs[4], single-pairing latency) of the default line")
    ap.add_argument("--workload", choices=["pairing", "g1mul", "g2mul", "gtpow", "prepared", "product", "product_prepared"], default="pairing",
                    help="pairing: the headline metric (default); the others are side metrics with their own line")
    ap.add_argument("--prepared-mode", choices=["native", "native_per_q", "reference"], default=None,
                    help="--workload prepared: the device-native table of bn254_g2_prepare (default) or the reference-image coefficients")
    ap.add_argument("--mode", choices=["dist", "multi_c"], default="dist",
                    help="dist (default, what the driver runs): one process per GPU over torch.distributed/RCCL, inputs resident in HBM; "
                         "multi_c: ONE host process drives all GPUs through bn254_*_multi of the C ABI (host buffers, PCIe inclusive)")
    args = ap.parse_args()

    if args.mode == "multi_c":
        return bench_multi_c(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_under_torchrun(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    import bn_amd
    from bn_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: bn_amd has no CPU path")
    # Test hooks for a ONE-GPU box (never set by the driver): BN254_BENCH_SHARE_GPU=1 puts all ranks on cuda:0 and
    # BN254_BENCH_BACKEND=gloo rendezvous over gloo, which exercises the N>1 control flow (shards, barriers, MAX over ranks, rank-0
    # line) without RCCL's one-GPU-per-rank requirement; BN254_BENCH_FORCE_DIST=1 initialises the RCCL process group even at
    # world 1, so that init_process_group("nccl"), barrier(device_ids), all_reduce(MAX) on a device tensor and
    # all_gather_into_tensor through RCCL execute on a one-GPU box exactly as they will on eight.
    share_gpu = os.environ.get("BN254_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("BN254_BENCH_BACKEND", "nccl")
    force_dist = os.environ.get("BN254_BENCH_FORCE_DIST") == "1"
    gpu_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    if world > 1 or force_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    eng = D.TorchEngine(bn_amd.Engine(gpu_index), dev)
    try:
        if args.workload in ("g1mul", "g2mul"):
            return bench_mul(args, eng, dev, world, rank, 1 if args.workload == "g1mul" else 2)
        if args.workload == "gtpow":
            return bench_gtpow(args, eng, dev, world, rank)
        if args.workload == "prepared":
            return bench_prepared(args, eng, dev, world, rank)
        if args.workload == "product_prepared":
            return bench_product_prepared(args, eng, dev, world, rank)
        if args.workload == "product":
            return bench_product(args, eng, dev, world, rank)

        if args.batch is not None:
            n = args.batch; lo = rank * n; scaling = "weak"; total = world * n
            workload = f"{n} independent pairings per GPU per step (--batch override)"
        elif world == 1:
            n = BATCH; lo = 0; scaling = "weak"; total = n
            workload = "2^16 independent pairings per step on 1 MI355X (BASELINE.json configs[1])"
        else:
            lo, hi = D.shard_range(TOTAL_MULTI, rank, world)
            n = hi - lo; scaling = "strong"; total = TOTAL_MULTI
            workload = f"2^20 independent pairings per step sharded over {world} MI355X, {n} per GPU, contiguous ranges (BASELINE.json configs[2])"
        P, Q = D.synthetic_points(eng, lo, lo + n)              # untimed: inputs resident in HBM before the clock starts
        out = eng.empty(n, 48)
        step = lambda: D.pairing_batch_sharded(eng, P, Q, out)
        elapsed = timed_steps(dist, dev, step, args.steps, args.warmup)          # profiling OFF inside the timed region

        if rank == 0:
            ksteps = min(args.steps, 10)
            # (--batch below the wave-machine thresholds of csrc/bn254_hip.hip: the whole pairing is ONE kernel, "pairing_wave")
            # (--batch in the small / mid-size windows of csrc/bn254_hip.hip: one kernel "pairing_wave", or the four-lane kernels "*_quad")
            stats = kernel_times(eng, dev, step, ("miller", "final_exp", "final_exp_wave", "pairing_wave", "miller_quad", "final_exp_quad"), ksteps)
            rf = roofline(eng, stats, n, MAC32_PER_PAIRING, dict(KERNEL_SHARE, final_exp_wave=KERNEL_SHARE["final_exp"], pairing_wave=1.0,
                                                                 miller_quad=KERNEL_SHARE["miller"], final_exp_quad=KERNEL_SHARE["final_exp"]), traffic_key=True, steps=ksteps)
            rf["algorithmic_hbm_bytes_per_launch"] = n * ALGO_BYTES_PER_PAIRING
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            if cus != 256:
                rf["traffic"] = rf["traffic_source"] = None          # the PMC figures in profiles/ were taken on 256 CUs (launch shapes differ elsewhere)
            if rf["traffic"] is not None:
                rf["hbm_GBps_of_8000"] = rf["traffic"] / (rf["avg_launch_ms"] * 1e-3) / 1e9
            if "traffic_per_step_all_kernels" in rf:
                rf["traffic_ratio_whole_step"] = rf["traffic_per_step_all_kernels"] / (n * ALGO_BYTES_PER_PAIRING)
            # the power-cap bound as a MEASUREMENT of this run: >= 1.2 s more of the same step, outside the timed region, sampled
            if not args.no_power:
                pw = PowerSampler(torch, dev).run(torch, dev, step, n, est_ms_per_step=elapsed / args.steps * 1e3)
                pw["energy_uJ_per_pairing"] = pw.pop("energy_uJ_per_unit")
                rf.update(pw)
                if pw["power_W"] is not None:
                    rf["bound"] = ("valu-int32-mac at the socket power cap (neither hbm nor mfma: SURVEY.md 8d): %.0f W of the %s W cap at sclk %s MHz under these kernels IN THIS RUN "
                                   "(power_W / power_cap_W / sclk_MHz / energy_uJ_per_pairing of this object)"
                                   % (pw["power_W"], "%.0f" % pw["power_cap_W"] if pw["power_cap_W"] else "?", "%.0f" % pw["sclk_MHz"] if pw["sclk_MHz"] else "?"))
            line = _line("BN254 optimal-ate pairings/sec (bit-exact vs ref)", "pairings/s", total * args.steps / elapsed, world, args, elapsed,
                         scaling, workload, {"roofline": rf, **({"expected_scaling": expected_scaling("pairing", world, scaling_inputs())} if world > 1 and args.batch is None else {})},
                         {"pairings_per_gpu": n, "inputs": "r*G1 / s*G2 (Jacobian, z != 1) resident in HBM", "parallelism": f"dp{world} (sharded, no collective)",
                          "number_system": "exact integer: 9 x 29-bit limbs in u32, v_mad_u64_u32 accumulation, Montgomery radix 2^261",
                          "mapping": 1,
                          # the lane-pair kernels are launched in rounds of 256 pairings per CU (two waves on every SIMD): the library
                          # sizes its sub-launches from the CU count of the device it finds (bn_round_pairs in csrc/bn254_hip.hip)
                          "cus": cus, "pairings_per_launch": n if n <= 256 * cus else f"equal parts of at most {256 * cus}",
                          "process_group": (dist.get_backend() if dist.is_initialized() else None)})
            if world == 1 and args.batch is None:
                Pn = P.cpu().numpy().view(np.uint64); Qn = Q.cpu().numpy().view(np.uint64)
                if not args.no_side:
                    line["side"] = side_object(eng, dev, dist, P, Q)
                    si = scaling_inputs(line)
                    line["expected_scaling"] = {"what": "predicted whole-step time at N GPUs from THIS run's one-GPU measurements (what the first multi-GPU run is compared with)",
                                                "configs[2]": {str(w): expected_scaling("pairing", w, si) for w in (2, 4, 8)},
                                                "configs[3]": {str(w): expected_scaling("product", w, si) for w in (1, 2, 4, 8)}}
                if not args.no_host_api:
                    line["host_api"] = host_api_rate(gpu_index, Pn, Qn)
                if not args.no_cpu_baseline:
                    line["cpu_baseline"] = cpu_baseline(Pn[:4096], Qn[:4096])
            print(json.dumps(line), flush=True)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
