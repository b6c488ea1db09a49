"""TEST INFRASTRUCTURE: the one-lane-per-pairing kernels (tests/testdouble/bn254_testdouble.hip - the engine's templates over Fq2A instead of
the lane-pair Fq2B) as a second GPU implementation for full-size cross checks.  Until round 4 this was `bn_amd.Engine(0, mapping=0)` inside the
product library; it left the product in round 5 and lives here.  OneLaneEngine offers the subset of bn_amd.Engine's methods the parity tests use,
so that `bn_amd.distributed.TorchEngine(OneLaneEngine(), dev)` works like the real one.  Needs torch (device memory) - like the tests that use it."""
import ctypes as C
import pathlib
import subprocess

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
SRC = HERE / "testdouble" / "bn254_testdouble.hip"
LIB = HERE / "testdouble" / "libbn254_testdouble.so"
CSRC = HERE.parent / "bn_amd" / "csrc"


def build(force=False):
    """hipcc cross-compiles for gfx950 without a GPU; the .so stays in-tree and travels to the GPU box (__graft_entry__.build calls this)"""
    from bn_amd import _native
    deps = [SRC] + sorted(CSRC.glob("*.hpp"))
    if force or not LIB.exists() or LIB.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.check_call([_native.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + _native.DEVICE_FLAGS +
                              ["-shared", f"-I{CSRC}", str(SRC), "-o", str(LIB)])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        from bn_amd import _native
        _native.lib()                                   # binds the process to ONE HIP runtime first (bn_amd/_native.py _preload_shared_hip_runtime)
        if not LIB.exists():
            raise RuntimeError(f"{LIB} is missing: python -c 'import __graft_entry__ as g; g.build()'")
        l = C.CDLL(str(LIB))
        vp, sz = C.c_void_p, C.c_size_t
        for name, args in {"bntd_miller": [vp, vp, vp, sz, vp], "bntd_final_exp": [vp, vp, sz, vp], "bntd_gt_product": [vp, sz, vp, vp, vp],
                           "bntd_g1_mul": [vp, vp, vp, sz, C.c_int, vp], "bntd_g2_mul": [vp, vp, vp, sz, C.c_int, vp]}.items():
            getattr(l, name).argtypes = args; getattr(l, name).restype = C.c_int
        _lib = l
    return _lib


def _check(rc):
    if rc:
        raise RuntimeError(f"test double: error {rc}")


class OneLaneEngine:
    def __init__(self, device=0):
        import torch
        self._lib = lib()
        self.device = int(device)
        self._dev = torch.device("cuda", self.device)

    def close(self):
        pass

    def _call(self, fn, *args):
        """the bntd_* entry points launch on the process's CURRENT HIP device: make that this engine's device for the call"""
        import torch
        with torch.cuda.device(self._dev):
            _check(fn(*args))

    def _cur_stream(self):
        import torch
        return torch.cuda.current_stream(self._dev).cuda_stream

    # ---- device-pointer entry points (names and argument order of bn_amd.Engine)
    def miller_batch_dev(self, d_p, d_q, d_f, n, stream=0):
        self._call(self._lib.bntd_miller, d_p, d_q, d_f, n, stream)

    def final_exp_batch_dev(self, d_f, d_out, n, stream=0):
        self._call(self._lib.bntd_final_exp, d_f, d_out, n, stream)

    def pairing_batch_dev(self, d_p, d_q, d_out, n, stream=0):
        self._call(self._lib.bntd_miller, d_p, d_q, d_out, n, stream); self._call(self._lib.bntd_final_exp, d_out, d_out, n, stream)

    def g1_mul_dev(self, d_p, d_k, d_out, n, stream=0, normalize=True):
        self._call(self._lib.bntd_g1_mul, d_p, d_k, d_out, n, 1 if normalize else 0, stream)

    def g2_mul_dev(self, d_p, d_k, d_out, n, stream=0, normalize=True):
        self._call(self._lib.bntd_g2_mul, d_p, d_k, d_out, n, 1 if normalize else 0, stream)

    def synthetic_scalars_dev(self, *a, **k):
        raise NotImplementedError("inputs are generated with the product engine")

    # ---- host-buffer conveniences on numpy uint64 arrays (shapes of bn_amd.Engine)
    def _up(self, a, width):
        import torch
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, width)
        return torch.from_numpy(a.view(np.int64)).to(self._dev)

    def _down(self, t):
        import torch
        torch.cuda.synchronize(self._dev)
        return t.cpu().numpy().view(np.uint64)

    def pairing_batch(self, p, q):
        import torch
        dp, dq = self._up(p, 12), self._up(q, 24)
        out = torch.empty(dp.shape[0], 48, dtype=torch.int64, device=self._dev)
        self.pairing_batch_dev(dp.data_ptr(), dq.data_ptr(), out.data_ptr(), dp.shape[0], self._cur_stream())
        return self._down(out)

    def pairing_product(self, p, q):
        """fold(Gt::one(), acc * pairing(p, q)): Miller values -> product tree -> ONE final exponentiation"""
        import torch
        dp, dq = self._up(p, 12), self._up(q, 24)
        n = dp.shape[0]
        f = torch.empty(n, 48, dtype=torch.int64, device=self._dev)
        tmp = torch.empty(2 * ((n + 3) // 4) + 1, 48, dtype=torch.int64, device=self._dev)
        one = torch.empty(1, 48, dtype=torch.int64, device=self._dev)
        st = self._cur_stream()
        self._call(self._lib.bntd_miller, dp.data_ptr(), dq.data_ptr(), f.data_ptr(), n, st)
        self._call(self._lib.bntd_gt_product, f.data_ptr(), n, one.data_ptr(), tmp.data_ptr(), st)
        self._call(self._lib.bntd_final_exp, one.data_ptr(), one.data_ptr(), 1, st)
        return self._down(one)[0]

    def _mul(self, fn, width, p, k):
        import torch
        dp, dk = self._up(p, width), self._up(k, 4)
        out = torch.empty_like(dp)
        fn(dp.data_ptr(), dk.data_ptr(), out.data_ptr(), dp.shape[0], self._cur_stream())
        return self._down(out)

    def g1_mul_batch(self, p, k):
        return self._mul(self.g1_mul_dev, 12, p, k)

    def g2_mul_batch(self, p, k):
        return self._mul(self.g2_mul_dev, 24, p, k)
