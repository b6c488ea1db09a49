#!/bin/bash
# Runs ON THE GPU BOX: which power / clock sources can an unprivileged process read?  (input to bench.py's sampler)
for c in /sys/class/drm/card*/device; do echo "== $c"; ls $c/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo
  for f in $c/hwmon/*/power1_average $c/hwmon/*/power1_input $c/hwmon/*/power1_cap $c/hwmon/*/freq1_input $c/hwmon/*/freq2_input $c/pp_dpm_sclk $c/gpu_busy_percent; do [ -r $f ] && echo "$f: $(cat $f 2>&1 | tr '\n' ' ')"; done; done
python - <<'PY'
import ctypes as C, time
for lib in ("/opt/rocm/lib/librocm_smi64.so", "/opt/rocm/lib/libamd_smi.so"):
    try:
        l = C.CDLL(lib); print("loaded", lib)
    except OSError as e:
        print("no", lib, e)
l = C.CDLL("/opt/rocm/lib/librocm_smi64.so")
print("rsmi_init", l.rsmi_init(C.c_uint64(0)))
n = C.c_uint32(); print("num", l.rsmi_num_monitor_devices(C.byref(n)), n.value)
p = C.c_uint64(); t = C.c_int()
t0 = time.time()
print("socket_power", l.rsmi_dev_current_socket_power_get(0, C.byref(p)), p.value, time.time() - t0)
print("power_ave", l.rsmi_dev_power_ave_get(0, 0, C.byref(p)), p.value)
try:
    print("power_get", l.rsmi_dev_power_get(0, C.byref(p), C.byref(t)), p.value, t.value)
except Exception as e: print(e)
class Freq(C.Structure):
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]
f = Freq(); t0 = time.time()
print("clk", l.rsmi_dev_gpu_clk_freq_get(0, 0, C.byref(f)), f.num_supported, f.current, list(f.frequency)[:4], time.time() - t0)
cap = C.c_uint64(); print("cap", l.rsmi_dev_power_cap_get(0, 0, C.byref(cap)), cap.value)
e = C.c_uint64(); res = C.c_float(); ts = C.c_uint64()
print("energy", l.rsmi_dev_energy_count_get(0, C.byref(e), C.byref(res), C.byref(ts)), e.value, res.value, ts.value)
PY
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -30
