import json, os, subprocess, sys, socket
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from ceres_slam_amd import synth
from oracle import oracle as orc
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); return s.getsockname()[1]
def run(world, size, mode="gpu_part", maxit=1000):
    port = free_port(); procs = []
    out = "/tmp/dbgpart"
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", SSBA_TEST_SIZE=",".join(map(str,size)), SSBA_TEST_MAXIT=str(maxit))
        procs.append(subprocess.Popen([sys.executable, "tests/dist_worker.py", mode, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    import time
    t0 = time.time()
    while time.time() - t0 < 60 and any(p.poll() is None for p in procs):
        time.sleep(0.5)
    for p in procs:
        if p.poll() is None:
            p.kill()
    outs = [p.communicate()[0].decode(errors="replace") for p in procs]
    for r,(p,o) in enumerate(zip(procs,outs)):
        if p.returncode: print("RANK", r, "rc", p.returncode, o[-1500:])
    if any(p.returncode for p in procs):
        return None
    return [json.load(open(f"{out}.{r}.json")) for r in range(world)]
for world, size, maxit in ((2,(300,9000,12),6), (3,(100,4000,12),1000)):
    res = run(world, size, maxit=maxit)
    if res is None:
        continue
    prob = synth.make_problem(size[0], size[1], track_len=size[2], seed=21)
    op = orc.OracleProblem.from_synth(prob); s2, log2 = op.solve(orc.driver_options(num_threads=2, max_num_iterations=maxit))
    print("world", world, size, "partition", res[0]["partition"], "term", [r["termination"] for r in res], "iters", [r["num_iterations"] for r in res], s2.num_iterations)
    n = max(len(res[0]["cost"]), len(log2["cost"]))
    for i in range(n):
        g = lambda k: ("%.9e" % res[0][k][i]) if i < len(res[0]["cost"]) else "-"
        o = lambda k: ("%.9e" % log2[k][i]) if i < len(log2["cost"]) else "-"
        print(i, "cost", g("cost"), o("cost"), "| acc", res[0]["accept"][i] if i < len(res[0]["accept"]) else "-", log2["step_is_successful"][i] if i < len(log2["cost"]) else "-",
              "| gmax", g("gmax"), o("gradient_max_norm"), "| step", g("step_norm"), o("step_norm"))
