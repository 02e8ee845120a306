"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, and exports every
symbol include/ssba.h declares.  No compute entry point is called here (no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from ceres_slam_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_loads():
    path = build.build_library()
    assert os.path.exists(path)
    capi.load()


def test_every_header_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "ssba.h")).read()
    declared = set(re.findall(r"\b(ssba_[a-z_]+)\s*\(", hdr))
    declared -= {"ssba_exchange_fn"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    lib = capi.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_the_dynamic_symbol_table_is_exactly_the_header():
    """libssba.so is linked next to torch's RCCL and user code: it defines the entry points of include/ssba.h and nothing
    else (-fvisibility=hidden + csrc/libssba.map) -- no unprefixed globals, no ssba:: C++ symbols."""
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "ssba.h")).read()
    declared = set(re.findall(r"\b(ssba_[a-z_]+)\s*\(", hdr)) - {"ssba_exchange_fn"}
    out = subprocess.run(["nm", "-D", "--defined-only", build.build_library()], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert defined == declared, defined ^ declared


def test_struct_layouts_match_header_sizes():
    # ssba_options: 8 int32 + 9 double + 2 int32 ; ssba_summary: 4 int32 + 4 double + 4 int32 (line-search counters)
    assert ctypes.sizeof(capi.Options) == 8 * 4 + 9 * 8 + 2 * 4
    assert ctypes.sizeof(capi.Summary) == 4 * 4 + 4 * 8 + 4 * 4
    assert ctypes.sizeof(capi.KernelTime) == 48 + 8 + 8
    assert ctypes.sizeof(capi.Camera) == 40


def test_defaults_are_the_ceres_1x_defaults():
    o = capi.default_options()
    assert o.max_num_iterations == 50 and o.use_nonmonotonic_steps == 0
    assert o.initial_trust_region_radius == 1e4 and o.min_relative_decrease == 1e-3
    assert o.function_tolerance == 1e-6 and o.gradient_tolerance == 1e-10 and o.parameter_tolerance == 1e-8
    assert o.min_lm_diagonal == 1e-6 and o.max_lm_diagonal == 1e32


def test_status_strings_and_no_cpu_fallback():
    lib = capi.load()
    assert b"no CPU fallback" in lib.ssba_status_string(-5)
    import torch
    if not torch.cuda.is_available():
        # without a GPU the product path must fail loudly, never compute on the CPU
        cam = capi.Camera(1, 1, 0, 0, 1)
        h = ctypes.c_void_p()
        rc = lib.ssba_create(ctypes.byref(cam), -1, ctypes.byref(h))
        assert rc == -5
        with pytest.raises(capi.SsbaError):
            capi.check(rc, "ssba_create")


def test_brief_report_format():
    s = capi.Summary(0, 11, 8, 3, 799981.7855, 28995.1736, 0.0, 0.0)
    buf = ctypes.create_string_buffer(256)
    assert capi.load().ssba_brief_report(ctypes.byref(s), buf, 256) == 0
    assert buf.value.decode() == ("Ceres Solver Report: Iterations: 11, Initial cost: 7.999818e+05, "
                                  "Final cost: 2.899517e+04, Termination: CONVERGENCE")


def test_reference_style_cpp_driver_compiles_against_the_shim():
    exe = build.build_examples()
    assert os.path.exists(exe)


def test_rccl_describe_names_the_library_without_a_gpu():
    """ssba_rccl_describe: which librccl.so file the process would use, its version and the IPC / debug environment -- the
    text a set-up time-out carries (include/ssba.h); loading the library needs no device."""
    from ceres_slam_amd.solver import StereoBA
    text = StereoBA.rccl_describe()
    assert text.startswith("librccl: ") and "HSA_ENABLE_IPC_MODE_LEGACY=" in text
    if "(not loaded)" not in text:
        assert "version" in text and "librccl" in text.split(";")[0]
