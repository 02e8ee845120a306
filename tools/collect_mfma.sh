# MFMA utilisation of the Schur kernels: one rocprofv3 --pmc pass (counters only, --kernel-trace) of bench.py with eager
# launches; per kernel: fp64 MFMA instructions, MFMA-busy cycles and GPU-active cycles.
# usage (GPU box): bash tools/collect_mfma.sh <out.txt> [bench args]
set -e
out=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp SSBA_NO_GRAPH=1 && cd $root
rm -rf gpurun_out/pmc_m
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_m -o m -- python3 bench.py --steps 10 --warmup 2 --no-kernel-timing --no-cpu-baseline "$@" > gpurun_out/pmc_m.log 2>&1
python3 - "$out" <<'PY'
import collections, csv, glob, sys
f = glob.glob("gpurun_out/pmc_m/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("ssba::", "").replace("void ", "").split("<")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1], "w") as o:
    o.write("# per launch (average).  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (64 cycles per v_mfma_f64_16x16x4_f64),\n"
            "# GRBM_GUI_ACTIVE over the 8 XCDs: MFMA-busy fraction = MFMA_busy / 1024 / (GRBM_active / 8)\n")
    o.write("kernel                        launches  MFMA_F64_insts  MFMA_busy_cycles  SQ_busy_cycles  GRBM_active  MFMA_busy_fraction\n")
    for k, c in sorted(acc.items()):
        if not c.get("SQ_INSTS_VALU_MFMA_F64") or sum(c["SQ_INSTS_VALU_MFMA_F64"]) == 0:
            continue
        n = len(c["SQ_INSTS_VALU_MFMA_F64"])
        avg = {m: sum(v) / len(v) for m, v in c.items()}
        frac = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / max(avg["GRBM_GUI_ACTIVE"] / 8, 1)
        o.write(f"{k:28s} {n:9d} {avg['SQ_INSTS_VALU_MFMA_F64']:15.0f} {avg['SQ_VALU_MFMA_BUSY_CYCLES']:17.0f} {avg['SQ_BUSY_CYCLES']:15.0f} {avg['GRBM_GUI_ACTIVE']:12.0f} {frac:10.3f}\n")
print(open(sys.argv[1]).read())
PY
