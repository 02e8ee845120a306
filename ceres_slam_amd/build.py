"""Builds the HIP library in-tree: hipcc --offload-arch=gfx950 -> ceres_slam_amd/libssba.so.

hipcc cross-compiles for gfx950 without a GPU, so this runs on the CPU-only build box;
the resulting .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libssba.so")
SOURCES = ["ssba_api.hip", "ssba_kernels.hip", "ssba_bcr.hip", "ssba_bcr_mfma.hip", "ssba_phong.hip", "ssba_phong_solver.hip", "ssba_border.hip", "ssba_frontend.hip", "ssba_dense.hip", "ssba_pool.hip", "ssba_wide.hip", "ssba_wide_layout.cpp", "ssba_layout.cpp"]
HEADERS = ["ssba_types.h", "ssba_wide_layout.h", "ssba_pool.h", "ssba_launch.h", "ssba_device.h", "ssba_phong_device.h", "ssba_linesearch.h", "ssba_posefactor_device.h", "libssba.map", os.path.join("..", "..", "include", "ssba.h")]
# -fvisibility=hidden + the version script below: the dynamic symbol table of libssba.so holds the entry points of
# include/ssba.h (SSBA_API) and nothing else -- no unprefixed helpers, no ssba:: C++ symbols next to torch's RCCL or user code
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wall", "-Wno-unused-function", *os.environ.get("SSBA_EXTRA_FLAGS", "").split(),
         "-Wno-unused-value", "-Wno-unused-variable", "-Wno-unused-result"]


# per-file flags: the factor kernel's accumulator tiles are also VALU operands (pivot rows, row scaling), so the matrix
# instructions take them from the architectural registers instead of bouncing them through v_accvgpr_read / _write
FILE_FLAGS = {"ssba_bcr_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        cmd = [hipcc, *[f for f in FLAGS if f], *FILE_FLAGS.get(s, []), "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for cmd, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out.decode(errors="replace"))
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "libssba.map"), "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return LIB


def build_examples(name: str = "dataset_vo_gpu") -> str:
    """C++ drivers written against the Ceres-shaped shim (include/ceres_slam_amd/ceres_shim.hpp):
    dataset_vo_gpu (tests/dataset_vo.cpp), dataset_ba_phong_gpu (tests/dataset_ba_phong.cpp) and dataset_vo_sun_gpu
    (tests/dataset_vo_sun.cpp)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "examples", name + ".cpp")
    out = os.path.join(root, "examples", name)
    build_library()
    inc = os.path.join(root, "include", "ceres_slam_amd")
    deps = [src, os.path.join(root, "include", "ssba.h"), LIB] + [os.path.join(inc, h) for h in sorted(os.listdir(inc)) if h.endswith(".hpp")]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-I" + os.path.join(root, "include"), src, "-o", out,
           "-L" + HERE, "-lssba", "-Wl,-rpath,$ORIGIN/../ceres_slam_amd", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("g++ failed:\n" + r.stdout.decode(errors="replace"))
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
