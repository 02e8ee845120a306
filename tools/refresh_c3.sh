# The config-3 and border part of tools/refresh_profiles.sh (benches with the CPU baseline, rocprofv3 stats, PMC traffic of the driver's
# configuration, one 600-case fuzz sweep with the seed given as argument 1): what changes when only the lighting / border kernels do.
set -x
root=$(pwd); out=gpurun_out/prof; mkdir -p $out
python3 bench.py --config C3 > $out/bench_c3.json 2> $out/bench_c3.err
python3 bench.py --config C3 --shared-free 7 > $out/bench_c3_free_shared.json 2> $out/bench_c3_free_shared.err
python3 bench.py --config C3 --shared-free 7 --bounds --dogleg 1 > $out/bench_c3_driver_config.json 2> $out/bench_c3_driver_config.err
python3 tools/bench_general.py > $out/bench_general_structure.json 2> $out/bench_general_structure.err
python3 tools/fuzz_parity.py 600 ${1:-131} > $out/fuzz_parity_600_seed${1:-131}.txt 2>&1; echo fuzz rc=$?
cd /tmp && export TMPDIR=/tmp && cd $root
stats() { n=$1; shift; rm -rf $out/rp_$n; rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp_$n -o s -- "$@" > $out/rp_$n.log 2>&1; cp $(find $out/rp_$n -name '*kernel_stats.csv' | head -1) $out/rocprofv3_kernel_stats_$n.csv; }
stats bench_c3 python3 bench.py --config C3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
stats bench_c3_free_shared python3 bench.py --config C3 --shared-free 7 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
stats bench_c3_driver_config python3 bench.py --config C3 --shared-free 7 --bounds --dogleg 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
stats loop_closure_border python3 tools/bench_general.py --case C2_loop_closure_border --steps 10
bash tools/collect_pmc.sh $out/pmc_traffic_c3_driver_config.json --config C3 --shared-free 7 --bounds --dogleg 1 > $out/pmc_drv.log 2>&1
echo done
