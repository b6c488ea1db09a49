"""TEST INFRASTRUCTURE - builds/loads tests/hostsim/libhostsim*.so (the DEVICE headers compiled for the CPU with g++)."""
import ctypes as C
import pathlib
import subprocess

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent / "hostsim"
CSRC = HERE.parents[1] / "bn_amd" / "csrc"
_U32P = C.POINTER(C.c_uint32)


def build(bounds=True):
    name = "libhostsim_bounds.so" if bounds else "libhostsim.so"
    out = HERE / name
    srcs = [HERE / "hostsim.cpp"] + sorted(CSRC.glob("*.hpp"))
    if (not out.exists()) or out.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
        cmd = ["g++", "-std=c++17", "-O1" if bounds else "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-pthread",
               "-o", str(out), str(HERE / "hostsim.cpp")]
        if bounds:
            cmd.insert(1, "-DBN_BOUNDS")
        subprocess.check_call(cmd)
    return out


class HostSim:
    def __init__(self, bounds=True):
        self.lib = C.CDLL(str(build(bounds)))
        assert bool(self.lib.hs_bounds_enabled()) == bounds

    def call(self, fn, *args, out_words):
        """args: numpy uint64 arrays (reference images) or ints; returns uint64 array of out_words/2"""
        o = np.zeros(out_words // 2, np.uint64)
        conv = []
        keep = []
        for a in args:
            if isinstance(a, (int, np.integer)):
                conv.append(C.c_int(int(a)))
            else:
                a = np.ascontiguousarray(a, dtype=np.uint64); keep.append(a)
                conv.append(a.ctypes.data_as(_U32P))
        getattr(self.lib, fn)(*conv, o.ctypes.data_as(_U32P))
        return o
