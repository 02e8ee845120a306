"""Diagnostic: per-phase cycle stamps of k_bcr_factor (needs a -DSSBA_STAMPS build)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceres_slam_amd import capi, synth
from ceres_slam_amd.solver import StereoBA
prob = synth.make_config("C2")
ba = StereoBA.from_synth(prob)
ba.lm_step(1e4, want_S=False)
ba.lm_step(1e4, want_S=False)
buf = (C.c_ulonglong * 4096)()
lib = capi.load()
lib.ssba_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
print("rc", lib.ssba_debug_stamps(ba.h, buf, 4096))
a = np.array(buf[:], dtype=np.int64)
for lev in range(8):
    s = a[lev * 64: lev * 64 + 64]
    if s[0] == 0:
        continue
    print("level", lev, "load", s[1] - s[0], "total", s[41] - s[0], "store", s[41] - s[40])
    for kb in range(12):
        prev = s[1] if kb == 0 else s[4 + 3 * (kb - 1)]
        print("   step", kb, "diag+bar", s[2 + 3 * kb] - prev, "panel+bar", s[3 + 3 * kb] - s[2 + 3 * kb], "trail", s[4 + 3 * kb] - s[3 + 3 * kb])

lev = 4
tr = a[1024 + lev * 64: 1024 + lev * 64 + 24]
dg = a[2048 + lev * 64: 2048 + lev * 64 + 24]
print("level", lev, "phase-3 duration per step: tile lane 340 | diag wave")
for kb in range(12):
    print("   step", kb, tr[2 * kb + 1] - tr[2 * kb], "|", dg[2 * kb + 1] - dg[2 * kb])
