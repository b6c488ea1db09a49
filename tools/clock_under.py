#!/usr/bin/env python3
"""Socket power and shader clock while ONE workload loops (bench.py's PowerSampler: hwmon power1_input / freq1_input every ~2 ms, >= 1.5 s):
the headline step, the prepared step (one table / one table per pairing), the fused multi-pairing, the multi-pairing over native tables
(one table per pair / one table for all), Gt::pow, G1 / G2 scalar multiplications.  At the power cap bytes are paid in clock: the clock column is
what the traffic column costs.  usage: tools/clock_under.py [workload ...]"""
import json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import bench
import bn_amd
from bn_amd import distributed as D

dev = torch.device("cuda", 0)
te = D.TorchEngine(bn_amd.Engine(0), dev)
e = te.e
n = 1 << 16
P, Q = D.synthetic_points(te, 0, 1 << 18)
out = te.empty(n, 48)
work = {}
work["pairing_2_16"] = (lambda: te.pairing_batch(P[:n], Q[:n], out), n)
prep1 = e.g2_prepare_dev(Q.data_ptr(), 1, te._stream()); prepq = e.g2_prepare_dev(Q.data_ptr(), 1 << 18, te._stream())
work["prepared_one_table"] = (lambda: e.pairing_prepared_native_dev(P.data_ptr(), prep1, out.data_ptr(), n, stream=te._stream()), n)
work["prepared_table_per_pairing"] = (lambda: e.pairing_prepared_native_dev(P.data_ptr(), prepq, out.data_ptr(), n, stream=te._stream()), n)
work["product_fused_2_18"] = (lambda: D.pairing_product_sharded(te, P, Q), 1 << 18)
work["product_prepared_2_18"] = (lambda: D.pairing_product_prepared_sharded(te, P, prepq), 1 << 18)
work["product_prepared_one_table_2_18"] = (lambda: D.pairing_product_prepared_sharded(te, P, prep1), 1 << 18)
k = D.synthetic_scalars_device(te, 1 << 24, (1 << 24) + (1 << 18), 1)
gt = te.pairing_batch(P[:n], Q[:n]).clone()
work["gt_pow_2_16"] = (lambda: te.gt_pow(gt, k[:n]), n)
work["g1_mul_2_18"] = (lambda: te.g1_mul(P, k), 1 << 18)
work["g2_mul_2_18"] = (lambda: te.g2_mul(Q, k), 1 << 18)
ps = bench.PowerSampler(torch, dev)
for name in (sys.argv[1:] or list(work)):
    step, units = work[name]
    for _ in range(3): step()
    torch.cuda.synchronize(dev)
    r = ps.run(torch, dev, step, units, min_seconds=1.5, est_ms_per_step=8.0)
    print(json.dumps({"workload": name, "power_W": round(r["power_W"] or 0, 1), "sclk_MHz": round(r["sclk_MHz"] or 0, 1), "uJ_per_unit": round(r.get("energy_uJ_per_unit") or 0, 2),
                      "M_units_per_s": round(r["units_per_s_during_leg"] / 1e6, 3)}), flush=True)
