# HBM traffic per kernel launch: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py with eager
# launches, summarised by tools/pmc_summarize.py.  Usage (on the GPU box): bash tools/collect_pmc.sh <out.json> [bench args]
set -e
out=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp SSBA_NO_GRAPH=1 && cd $root
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 10 --warmup 2 --no-kernel-timing --no-cpu-baseline "$@" > gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 10 --warmup 2 --no-kernel-timing --no-cpu-baseline "$@" > gpurun_out/pmc_w.log 2>&1
python3 tools/pmc_summarize.py $(find gpurun_out/pmc_f -name '*counter_collection.csv') $(find gpurun_out/pmc_w -name '*counter_collection.csv') $out "bench.py --steps 10 --warmup 2 --no-kernel-timing $*, SSBA_NO_GRAPH=1"
