#!/usr/bin/env python3
"""Cross-path soak ON THE GPU BOX: the same pairings through the wave machine, the four-lane kernels, the lane-pair kernels and the native
prepared-G2 kernels (one table per pairing; one shared table) must give the same bytes - for batch sizes around every threshold and wave /
workgroup boundary, with points at infinity sprinkled in -, the multi-pairing product through its three routes likewise, and a slice of every batch against the CPU oracle.
tests/test_gpu_soak.py runs soak() as a -m gpu test (so the driver's round-end run executes it); as a tool:
    python tools/soak_paths.py > gpurun_out/soak.txt"""
import pathlib, sys, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]

SIZES = [1, 2, 3, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 1023, 1024, 1025, 3583, 3584, 3585, 4097, 8191, 8192, 16383, 16384, 16385,
         20001, 32767, 32768, 32769, 65535, 65536, 65537, 100003]
# the option values stay inside what one launch can address (bn254_ctx_set_option rejects more: bn_opt_valid)
PATHS = {"lane_pair": dict(quad_max=0, wave_pairing_max=0, wave_fe_max=0), "quad": dict(wave_pairing_max=0, wave_fe_max=0, quad_max=1 << 20),
         "wave": dict(wave_pairing_max=1 << 20, wave_fe_max=1 << 20), "default": {}}


def soak(te, oracle, sizes=SIZES, log=print):
    """(comparisons, mismatches); every mismatch is logged"""
    import torch
    from bn_amd import distributed as D
    e = te.e
    nmax = max(sizes)
    P, Q = D.synthetic_points(te, 5_000_000, 5_000_000 + nmax)
    rng = np.random.default_rng(4)
    inf = torch.from_numpy(rng.integers(0, 50, nmax) == 0).to(P.device)                       # 2 % of the pairs get an infinite side
    which = torch.from_numpy(rng.integers(0, 3, nmax)).to(P.device)
    g1z = torch.from_numpy(np.array(oracle.g1_zero()).view(np.int64)).to(P.device); g2z = torch.from_numpy(np.array(oracle.g2_zero()).view(np.int64)).to(P.device)
    P[inf & (which != 1)] = g1z; Q[inf & (which != 0)] = g2z
    checked = bad = 0
    for n in sizes:
        p, q = P[:n].contiguous(), Q[:n].contiguous()
        outs = {}
        for name, opts in PATHS.items():
            if name == "wave" and n > 8192: continue                                            # (one pairing per wave: minutes beyond this)
            if name == "quad" and n > 70000: continue
            with e.options(**opts):
                outs[name] = te.pairing_batch(p, q).clone()
                prod = te.final_exp(te.miller_product(p, q)).clone() if n <= 40000 else None     # (the Miller products differ by factors the exponentiation kills)
                outs[name + "_product"] = prod
            torch.cuda.synchronize()
        # the native prepared-G2 path on the same pairs: one table per pairing (infinite Q included: its flag makes the pairing one), and for
        # some sizes ONE shared table against the general path on a tiled Q
        prep = e.g2_prepare_dev(q.data_ptr(), n, te._stream())
        nat = te.empty(n, 48)
        with e.options(wave_pairing_max=0):                                                     # the native kernels at EVERY size (small calls are routed to the wave kernels by default)
            e.pairing_prepared_native_dev(p.data_ptr(), prep, nat.data_ptr(), n, stream=te._stream())
            torch.cuda.synchronize()
        outs["native_per_q"] = nat
        # the multi-pairing over the same tables: plain native kernels, two and four pairs per accumulator (forced at every size; the fused
        # product of the lane-pair kernels is the reference of the `_product` comparisons below)
        if n <= 40000:
            for m in (1, 2, 4):
                part = te.empty(1, 48)
                with e.options(wave_pairing_max=0, miller_shared=m, round_pairs=max(1, min(1 << 16, n // (2 * m)))):
                    e.miller_product_prepared_native_dev(p.data_ptr(), prep, n, part.data_ptr(), stream=te._stream())
                    e.final_exp_batch_dev(part.data_ptr(), part.data_ptr(), 1, te._stream())
                    torch.cuda.synchronize()
                outs[f"native_m{m}_product"] = part.reshape(-1).clone()
        if n <= 4097:                                                                           # ... and the default route on both sides of its limit
            natd = te.empty(n, 48)
            e.pairing_prepared_native_dev(p.data_ptr(), prep, natd.data_ptr(), n, stream=te._stream())
            torch.cuda.synchronize()
            outs["native_default_route"] = natd
        prep.close()
        if n in (1, 33, 257, 1025, 16385, 65537):
            qs = Q[7:8].contiguous()                                                            # one point for all (were it one of the infinite ones, both sides would be one)
            prep1 = e.g2_prepare_dev(qs.data_ptr(), 1, te._stream())
            nat1 = te.empty(n, 48)
            with e.options(wave_pairing_max=0):
                e.pairing_prepared_native_dev(p.data_ptr(), prep1, nat1.data_ptr(), n, stream=te._stream())
                torch.cuda.synchronize()
            with e.options(**PATHS["lane_pair"]):
                want1 = te.pairing_batch(p, qs.expand(n, 24).contiguous())
            torch.cuda.synchronize()
            same = bool(torch.equal(nat1, want1)); checked += 1; bad += not same
            if not same: log(f"MISMATCH n={n} path=native_shared")
            prep1.close()
        ref = outs["lane_pair"]
        for name, o in outs.items():
            if o is None or name == "lane_pair": continue
            r = outs["lane_pair_product"] if name.endswith("_product") else ref
            if r is None: continue
            same = bool(torch.equal(o, r)); checked += 1; bad += not same
            if not same: log(f"MISMATCH n={n} path={name}")
        # a slice against the oracle: first, last, infinite ones
        idx = sorted(set(list(range(min(3, n))) + list(range(max(0, n - 3), n)) + [int(i) for i in torch.nonzero(inf[:n])[:2].flatten().tolist()]))
        want = oracle.pairing_batch(P[idx].cpu().numpy().view(np.uint64), Q[idx].cpu().numpy().view(np.uint64))
        got = ref[idx].cpu().numpy().view(np.uint64)
        ok = bool(np.array_equal(got, want)); checked += 1; bad += not ok
        if not ok: log(f"ORACLE MISMATCH n={n}")
        log(f"n={n:6d} paths={[k for k, v in outs.items() if v is not None and not k.endswith('_product')]} infinite={int(inf[:n].sum())} {'ok' if ok else 'BAD'}")
    return checked, bad


if __name__ == "__main__":
    import torch
    import bn_amd
    from bn_amd import distributed as D
    import bn_oracle
    bn_oracle.build()
    t0 = time.time()
    checked, bad = soak(D.TorchEngine(bn_amd.Engine(0), torch.device("cuda", 0)), bn_oracle.Oracle(), log=lambda s: print(s, flush=True))
    print(f"{checked} comparisons, {bad} mismatches, {time.time() - t0:.1f} s")
    sys.exit(1 if bad else 0)
