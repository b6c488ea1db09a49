"""ctypes binding of libbn254_hip.so (include/bn254_hip.h).  Fails loudly when the HIP library or a GPU is missing:
there is no CPU fallback anywhere in this package."""
import ctypes as C
import os
import pathlib
import subprocess

HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ.get("BN254_LIB_PATH", HERE / "libbn254_hip.so"))     # override: kernel experiments only
SOURCES = [HERE / "csrc" / f for f in ("bn254_hip.hip", "bn254_kernels_b.hip", "bn254_kernels_mul.hip", "bn254_kernels_w.hip", "bn254_kernels_q.hip", "bn254_multi.hip", "bn254_measure.hip")]
OBJ_DIR = HERE / "csrc" / "build"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# EVERY unit is compiled with LLVM's DPP combiner off: with it on, the quad_perm move of a neighbour lane's limb is folded into the
# subtraction that consumes it, and when the DPP value is the SUBTRAHEND the combiner emits v_subrev_u32_dpp d, x, a - which on gfx950 computes
# dpp(a) - x instead of a - dpp(x) (profiles/r05_dpp_fold_bisect.txt; round 4: every four-lane pairing wrong).  The flag costs nothing (34 harmless
# folds into v_add_u32 become moves; profiles/r05_ab_dpp_combine_off.txt); tests/test_build_quality.py checks the result.
DEVICE_FLAGS = ["-mllvm", "-amdgpu-dpp-combine=false"]

_VP = C.c_void_p
_SZ = C.c_size_t

SIGNATURES = {   # name -> argtypes  (every function returns int unless noted)
    "bn254_device_count": [],
    "bn254_ctx_create": [C.c_int, C.POINTER(_VP)],
    "bn254_ctx_destroy": [_VP],
    "bn254_error_string": [C.c_int],
    "bn254_ctx_set_mapping": [_VP, C.c_int],
    "bn254_ctx_set_option": [_VP, C.c_int, C.c_long],
    "bn254_ctx_get_option": [_VP, C.c_int, C.POINTER(C.c_long)],
    "bn254_ctx_get_option_raw": [_VP, C.c_int, C.POINTER(C.c_long)],
    "bn254_pairing_batch": [_VP, _VP, _VP, _VP, _SZ],
    "bn254_pairing_product": [_VP, _VP, _VP, _SZ, _VP],
    "bn254_g1_mul_batch": [_VP, _VP, _VP, _VP, _SZ],
    "bn254_g2_mul_batch": [_VP, _VP, _VP, _VP, _SZ],
    "bn254_g1_add_batch": [_VP, _VP, _VP, _VP, _SZ, C.c_int],
    "bn254_g2_add_batch": [_VP, _VP, _VP, _VP, _SZ, C.c_int],
    "bn254_fr_encode_batch": [_VP, _VP, _VP, _SZ],
    "bn254_fr_decode_batch": [_VP, _VP, _VP, _VP, _SZ],
    "bn254_g1_encode_batch": [_VP, _VP, _VP, _SZ],
    "bn254_g2_encode_batch": [_VP, _VP, _VP, _SZ],
    "bn254_g1_decode_batch": [_VP, _VP, _VP, _VP, _SZ],
    "bn254_g2_decode_batch": [_VP, _VP, _VP, _VP, _SZ],
    "bn254_g1_encode_stream": [_VP, _VP, _SZ, _VP, _SZ, C.POINTER(_SZ)],
    "bn254_g2_encode_stream": [_VP, _VP, _SZ, _VP, _SZ, C.POINTER(_SZ)],
    "bn254_g1_decode_stream": [_VP, _VP, _SZ, _VP, _VP, _SZ, C.POINTER(_SZ), C.POINTER(_SZ)],
    "bn254_g2_decode_stream": [_VP, _VP, _SZ, _VP, _VP, _SZ, C.POINTER(_SZ), C.POINTER(_SZ)],
    "bn254_g2_precompute": [_VP, _VP, _VP, _SZ],
    "bn254_pairing_prepared_batch": [_VP, _VP, _VP, C.c_int, _VP, _SZ],
    "bn254_g2_precompute_dev": [_VP, _VP, _VP, _SZ, _VP],
    "bn254_miller_prepared_dev": [_VP, _VP, _VP, C.c_int, _VP, _SZ, _VP],
    "bn254_g2_prepare": [_VP, _VP, _SZ, C.POINTER(_VP)],
    "bn254_g2_prepare_dev": [_VP, _VP, _SZ, C.POINTER(_VP), _VP],
    "bn254_g2_prepared_destroy": [_VP],
    "bn254_g2_prepared_count": [_VP],
    "bn254_g2_prepared_bytes": [_VP],
    "bn254_g2_prepared_export": [_VP, _VP, _VP, _SZ],
    "bn254_pairing_prepared_native_batch": [_VP, _VP, _VP, _VP, _SZ],
    "bn254_miller_prepared_native_dev": [_VP, _VP, _VP, _SZ, _VP, _SZ, _VP],
    "bn254_pairing_prepared_native_batch_dev": [_VP, _VP, _VP, _SZ, _VP, _SZ, _VP],
    "bn254_pairing_product_prepared_native": [_VP, _VP, _VP, _SZ, _VP],
    "bn254_miller_product_prepared_native_dev": [_VP, _VP, _VP, _SZ, _SZ, _VP, _VP],
    "bn254_gt_mul_batch": [_VP, _VP, _VP, _VP, _SZ],
    "bn254_gt_pow_batch": [_VP, _VP, _VP, _VP, _SZ],
    "bn254_gt_inverse_batch": [_VP, _VP, _VP, _SZ],
    "bn254_gt_inverse_batch_dev": [_VP, _VP, _VP, _SZ, _VP],
    "bn254_exp_by_neg_z_dev": [_VP, _VP, _VP, _SZ, _VP],
    "bn254_multi_create": [C.POINTER(C.c_int), C.c_int, C.POINTER(_VP)],
    "bn254_multi_create_ex": [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(_VP)],
    "bn254_multi_set_option": [_VP, C.c_int, C.c_long],
    "bn254_multi_destroy": [_VP],
    "bn254_multi_device_count": [_VP],
    "bn254_multi_exchange_kind": [_VP],
    "bn254_multi_rank_numa_node": [_VP, C.c_int],
    "bn254_multi_ctx": [_VP, C.c_int],
    "bn254_g2_prepare_multi": [_VP, _VP, _SZ, C.POINTER(_VP)],
    "bn254_multi_prepared_destroy": [_VP],
    "bn254_multi_prepared_count": [_VP],
    "bn254_pairing_prepared_native_batch_multi": [_VP, _VP, _VP, _VP, _SZ],
    "bn254_pairing_product_prepared_native_multi": [_VP, _VP, _VP, _SZ, _VP],
    "bn254_pairing_batch_multi": [_VP, _VP, _VP, _VP, _SZ],
    "bn254_pairing_product_multi": [_VP, _VP, _VP, _SZ, _VP],
    "bn254_synthetic_scalars_dev": [_VP, C.c_uint64, C.c_uint64, _SZ, C.c_int, _VP, _VP],
    "bn254_tile_dev": [_VP, _VP, _SZ, _SZ, _VP, _VP],
    "bn254_ubench_mac32": [_VP, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "bn254_ubench_mac32_ex": [_VP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "bn254_wave_ubench": [_VP, C.c_int, C.c_int, C.POINTER(C.c_double)],
    "bn254_gt_mul_batch_dev": [_VP, _VP, _VP, _VP, _SZ, _VP],
    "bn254_gt_pow_batch_dev": [_VP, _VP, _VP, _VP, _SZ, _VP],
    "bn254_pairing_batch_dev": [_VP, _VP, _VP, _VP, _SZ, _VP],
    "bn254_miller_batch_dev": [_VP, _VP, _VP, _VP, _SZ, _VP],
    "bn254_final_exp_batch_dev": [_VP, _VP, _VP, _SZ, _VP],
    "bn254_gt_product_dev": [_VP, _VP, _SZ, _VP, _VP],
    "bn254_miller_product_dev": [_VP, _VP, _VP, _SZ, _VP, _VP],
    "bn254_gt_product_final_exp_dev": [_VP, _VP, _SZ, _VP, _VP],
    "bn254_g1_mul_batch_dev": [_VP, _VP, _VP, _VP, _SZ, _VP],
    "bn254_g2_mul_batch_dev": [_VP, _VP, _VP, _VP, _SZ, _VP],
    "bn254_g1_mul_jacobian_dev": [_VP, _VP, _VP, _VP, _SZ, _VP],
    "bn254_g2_mul_jacobian_dev": [_VP, _VP, _VP, _VP, _SZ, _VP],
    "bn254_profile_enable": [_VP, C.c_int],
    "bn254_profile_reset": [_VP],
    "bn254_kernel_stats": [_VP, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)],
}


def build(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU; the .so is kept in-tree so it travels to the GPU box.  The translation
    units are compiled in parallel into csrc/build/*.o and only those whose inputs changed are rebuilt."""
    import concurrent.futures
    hdrs = sorted((HERE / "csrc").glob("*.hpp")) + [HERE.parent / "include" / "bn254_hip.h"]
    hdr_m = max(h.stat().st_mtime for h in hdrs)
    extra = os.environ.get("BN254_EXTRA_HIPCC_FLAGS", "").split()           # experiments only
    if any("dpp-combine" in f for f in extra):                              # appended AFTER DEVICE_FLAGS, so it could turn the combiner back on
        raise RuntimeError("BN254_EXTRA_HIPCC_FLAGS must not touch -amdgpu-dpp-combine: the library is only correct with the combiner off (DEVICE_FLAGS)")
    flags = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + DEVICE_FLAGS + extra
    OBJ_DIR.mkdir(exist_ok=True)
    stamp = OBJ_DIR / "flags.txt"
    if not stamp.exists() or stamp.read_text() != " ".join(flags):
        force = True
    jobs = []
    for src in SOURCES:
        obj = OBJ_DIR / (src.stem + ".o")
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_m):
            jobs.append(flags + ["-c", str(src), "-o", str(obj)])
    if jobs:
        if verbose:
            for j in jobs:
                print(" ".join(j))
        with concurrent.futures.ThreadPoolExecutor(len(jobs)) as ex:
            list(ex.map(subprocess.check_call, jobs))
        stamp.write_text(" ".join(flags))
    objs = [OBJ_DIR / (src.stem + ".o") for src in SOURCES]
    if jobs or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < max(o.stat().st_mtime for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [str(o) for o in objs] + ["-ldl", "-lpthread", "-o", str(LIB_PATH)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def _preload_shared_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes in one process do
    not work (the second sees no GPU), so when torch is installed we bind to ITS runtime, whichever of the two gets
    imported first.  Nothing of torch itself is imported here."""
    override = os.environ.get("BN254_HIP_RUNTIME")
    cands = [override] if override else []
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            cands.append(os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so"))
    except Exception:
        pass
    for c in cands:
        if c and os.path.exists(c):
            try:
                C.CDLL(c, mode=C.RTLD_GLOBAL)
                # the multi-device path dlopens RCCL lazily: it must be the one built against THIS runtime
                rccl = os.path.join(os.path.dirname(c), "librccl.so")
                if os.path.exists(rccl):
                    os.environ.setdefault("BN254_RCCL_PATH", rccl)
                return c
            except OSError:
                continue
    return None


def lib():
    global _lib
    if _lib is None:
        _preload_shared_hip_runtime()
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950).  bn_amd has no CPU fallback.")
        l = C.CDLL(str(LIB_PATH))
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)            # AttributeError if the library does not export a declared symbol
            fn.argtypes = args
            fn.restype = C.c_int
        l.bn254_error_string.restype = C.c_char_p
        l.bn254_ctx_destroy.restype = None
        l.bn254_multi_destroy.restype = None
        l.bn254_g2_prepared_destroy.restype = None
        l.bn254_multi_prepared_destroy.restype = None
        l.bn254_multi_prepared_count.restype = C.c_size_t
        l.bn254_g2_prepared_count.restype = C.c_size_t
        l.bn254_g2_prepared_bytes.restype = C.c_size_t
        l.bn254_multi_ctx.restype = C.c_void_p
        _lib = l
    return _lib


# BN254_OPT_* of include/bn254_hip.h
OPTIONS = {"wave_pairing_max": 1, "wave_fe_max": 2, "quad_max": 3, "miller_shared": 4, "gt_pow_mode": 5, "product_chunk": 6,
           "product_per_wave": 7, "product_bfly": 8, "round_pairs": 9, "pipeline_chunk": 10, "pipeline_slots": 11, "stream_stop_at_error": 12}
EXCHANGE = {"auto": -1, "peer": 0, "rccl": 1}


class Bn254Error(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise Bn254Error(f"bn254_hip error {rc}: {lib().bn254_error_string(rc).decode()}")
