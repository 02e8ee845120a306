"""Diagnostic: shader-clock stamps of k_wd_schur (one middle work-group, waves 0 and 4) on the LT24 problem.  Needs a library built
with SSBA_EXTRA_FLAGS=-DWD_STAMPS (python -c "import __graft_entry__ as g; g.build()" after touching csrc/ssba_wide.hip)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceres_slam_amd import capi, synth
from ceres_slam_amd.solver import StereoBA
os.environ["SSBA_NO_GRAPH"] = "1"
prob = synth.make_config("LT24")
ba = StereoBA.from_synth(prob)
ba.lm_step(1e4, want_S=False)
ba.lm_step(1e4, want_S=False)
buf = (C.c_ulonglong * 1024)()
lib = capi.load()
lib.ssba_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
print("rc", lib.ssba_debug_stamps(ba.h, buf, 1024))
a = np.array(buf[:], dtype=np.int64)
for w in (0, 1):
    s = a[256 + 64 * w: 256 + 64 * w + 64]
    print(f"k_wd_schur, wave {4 * w}: prologue (poses, slot table, landmark factors, barrier) {s[1] - s[0]} cycles; whole item {s[60] - s[0]}")
    for b in range(8):
        if s[5 + 4 * b] == 0:
            break
        prev = s[1] if b == 0 else s[5 + 4 * (b - 1)]
        print(f"   batch {b}: producer {s[2 + 4 * b] - prev}, barrier {s[3 + 4 * b] - s[2 + 4 * b]}, products {s[4 + 4 * b] - s[3 + 4 * b]}, barrier {s[5 + 4 * b] - s[4 + 4 * b]}")
    last = max(b for b in range(9) if b == 0 or s[5 + 4 * (b - 1)] != 0)
    print(f"   slab store {s[60] - s[5 + 4 * (last - 1)]}")
