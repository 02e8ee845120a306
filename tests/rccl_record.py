"""Child-process runner for the tests that go through RCCL (test infrastructure).

RCCL's communicator set-up has been seen to hang on this pool.  The library bounds its own RCCL set-up
calls (ssba_set_rccl / ssba_rccl_unique_id return SSBA_ERR_TIMEOUT with a description of where they sat); this runner adds
what only the outside can see and keeps ALL of it on disk, so that a hang leaves a record instead of a bare skip:

  * the child runs with NCCL_DEBUG=INFO into NCCL_DEBUG_FILE, PYTHONFAULTHANDLER=1 and SSBA_RCCL_TIMEOUT_S below the outer limit;
  * if the child itself outlives the outer limit, its /proc/<pid>/maps lines (which librccl / libamdhip64 files are mapped),
    the state, wait channel and name of every thread, then SIGABRT (faulthandler prints the Python stacks), then SIGKILL
    of exactly that pid;
  * the record goes to gpurun_out/rccl_records/<tag>/ (merged back from the GPU box) and into the test's message.
"""
import glob
import os
import signal
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _proc_snapshot(pid):
    out = []
    try:
        seen = set()
        for ln in open(f"/proc/{pid}/maps"):
            path = ln.split()[-1] if len(ln.split()) >= 6 else ""
            if any(k in path for k in ("rccl", "amdhip", "hsa-runtime", "libssba")) and path not in seen:
                seen.add(path)
                out.append("mapped: " + path)
    except OSError as e:
        out.append(f"maps unreadable: {e}")
    for task in sorted(glob.glob(f"/proc/{pid}/task/*")):
        def rd(name):
            try:
                return open(os.path.join(task, name)).read().strip()
            except OSError:
                return "?"
        state = next((ln.split(":", 1)[1].strip() for ln in rd("status").splitlines() if ln.startswith("State")), "?")
        out.append(f"thread {os.path.basename(task)} comm={rd('comm')} state={state} wchan={rd('wchan')}")
    return out


def run(cmd, tag, inner_timeout_s=100, outer_timeout_s=240, env_extra=None):
    """Returns (returncode or None if the child had to be killed, stdout, stderr, record_text, record_dir)."""
    base = os.path.join(ROOT, "gpurun_out", "rccl_records")
    rec_dir = os.path.join(base, f"{tag}_{int(time.time())}")
    os.makedirs(rec_dir, exist_ok=True)
    env = dict(os.environ)
    env.update({"NCCL_DEBUG": "INFO", "NCCL_DEBUG_FILE": os.path.join(rec_dir, "nccl_%h_%p.log"), "PYTHONFAULTHANDLER": "1",
                "SSBA_RCCL_TIMEOUT_S": str(inner_timeout_s)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")      # single node: bootstrap over loopback, not the container's external interface
    env.update(env_extra or {})
    t0 = time.time()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    notes = [f"cmd: {' '.join(map(str, cmd))}", f"HSA_ENABLE_IPC_MODE_LEGACY={env.get('HSA_ENABLE_IPC_MODE_LEGACY')}"]
    try:
        out, err = proc.communicate(timeout=outer_timeout_s)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        notes.append(f"child {proc.pid} still running after {outer_timeout_s} s: snapshot, SIGABRT, SIGKILL")
        notes += _proc_snapshot(proc.pid)
        proc.send_signal(signal.SIGABRT)
        try:
            out, err = proc.communicate(timeout=15)
        except subprocess.TimeoutExpired:
            proc.kill()
            try:
                out, err = proc.communicate(timeout=15)
            except subprocess.TimeoutExpired:      # stuck in the driver: leave it, do not block the test run
                out, err = "", "(child did not die after SIGKILL within 15 s)"
        rc = None
    notes.append(f"elapsed {time.time() - t0:.1f} s, returncode {rc}")
    logs = sorted(glob.glob(os.path.join(rec_dir, "nccl_*.log")))
    tail = []
    for lg in logs:
        lines = open(lg, errors="replace").read().splitlines()
        tail.append(f"--- {os.path.basename(lg)} ({len(lines)} lines), last 25:")
        tail += lines[-25:]
    record = "\n".join(notes + ["--- stderr (last 3000 chars):", (err or "")[-3000:]] + tail)
    with open(os.path.join(rec_dir, "record.txt"), "w") as fh:
        fh.write(record + "\n--- stdout:\n" + (out or ""))
    return rc, out, err, record, rec_dir


def library_timed_out(err):
    """True when libssba.so's own time limit fired: the hang sat inside the RCCL call it names."""
    return "did not return within" in (err or "")
