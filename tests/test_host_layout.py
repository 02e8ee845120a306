"""The host phase of ssba_finalize is plain C++ (ceres_slam_amd/csrc/ssba_layout.cpp: landmark order, windows and slots, closure
border, general layout with pair lists and the symbolic factorisation, the threaded fill of the observation arrays;
ssba_wide_layout.cpp: the wide reduced system).  It is compiled here for the CPU with -fsanitize=address,undefined and once with
-fsanitize=thread and run over C1- / C2-shaped, long-track, loop-closure, lighting and pose-graph problems (SURVEY.md section
5: sanitizers on the CPU build only -- GPU ASan is not available on this pool)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ceres_slam_amd", "csrc")


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_wide_layout_builder_under_sanitizers(tmp_path, san):
    exe = str(tmp_path / ("wide_layout_check_" + san.replace(",", "_")))
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=" + san, "-fno-sanitize-recover=all", "-I" + CSRC,
           os.path.join(ROOT, "tests", "host", "wide_layout_check.cpp"), os.path.join(CSRC, "ssba_wide_layout.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all invariants hold" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_finalize_host_phase_under_sanitizers(tmp_path, san):
    exe = str(tmp_path / ("layout_check_" + san.replace(",", "_")))
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=" + san, "-fno-sanitize-recover=all", "-pthread", "-I" + CSRC,
           os.path.join(ROOT, "tests", "host", "layout_check.cpp"), os.path.join(CSRC, "ssba_layout.cpp"), os.path.join(CSRC, "ssba_wide_layout.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "all invariants hold" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "threaded fill" in r.stdout
