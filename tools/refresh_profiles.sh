# Re-measures everything under profiles/ that comes from bench.py / tools on the GPU box (run from the repo root):
#   bash tools/refresh_profiles.sh benches|rocprof   -> writes gpurun_out/prof/*; copy what is to be judged into profiles/
# rocprofv3 runs keep the number of graph-replayed dispatches of a process below 16 384: rocprofiler-sdk 1.1.0 (ROCm 7.2)
# faults in its queue interceptor when a replayed batch straddles the end of its 1 MiB packet buffer.
set -x
root=$(pwd)
out=gpurun_out/prof
mkdir -p $out
if [ "$1" = benches ]; then
python3 bench.py > $out/bench_c2.json 2> $out/bench_c2.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_c2_driver_args.json 2> $out/bench_c2_driver_args.err
python3 bench.py --config C3 > $out/bench_c3.json 2> $out/bench_c3.err
python3 bench.py --config C3 --shared-free 7 > $out/bench_c3_free_shared.json 2> $out/bench_c3_free_shared.err
python3 bench.py --config C3 --shared-free 7 --bounds --dogleg 1 > $out/bench_c3_driver_config.json 2> $out/bench_c3_driver_config.err
python3 bench.py --config C5 > $out/bench_c5.json 2> $out/bench_c5.err
python3 bench.py --config LT24 > $out/bench_lt24.json 2> $out/bench_lt24.err
python3 bench.py --config C4 --steps 50 --warmup 5 > $out/bench_c4_single_gpu.json 2> $out/bench_c4_single_gpu.err
SSBA_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 > $out/bench_rehearsal_2_ranks_gloo.json 2> $out/bench_rehearsal_2_ranks_gloo.err
python3 tools/rank_compute_time.py 1 2 4 8 > $out/rank_compute_time.json 2> $out/rank_compute_time.err
python3 tools/bench_general.py > $out/bench_general_structure.json 2> $out/bench_general_structure.err
./tools/fp64_calib > $out/fp64_calibration.txt 2>&1
./tools/launch_floor 200 20 1 > $out/launch_floor.txt 2>&1
./tools/launch_floor 200 20 256 >> $out/launch_floor.txt 2>&1
fi
if [ "$1" = rocprof ]; then
cd /tmp && export TMPDIR=/tmp && cd $root
stats() {   # name, then the program and its arguments
    n=$1; shift
    rm -rf $out/rp_$n
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/rp_$n -o s -- "$@" > $out/rp_$n.log 2>&1
    cp $(find $out/rp_$n -name '*kernel_stats.csv' | head -1) $out/rocprofv3_kernel_stats_$n.csv
}
stats bench_c2 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
stats bench_c3 python3 bench.py --config C3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
stats bench_c3_free_shared python3 bench.py --config C3 --shared-free 7 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
stats bench_c3_driver_config python3 bench.py --config C3 --shared-free 7 --bounds --dogleg 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
stats bench_c4_single_gpu python3 bench.py --config C4 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
stats bench_c5 python3 bench.py --config C5 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
stats bench_lt24 python3 bench.py --config LT24 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
stats loop_closure_border python3 tools/bench_general.py --case C2_loop_closure_border --steps 10
stats general_path_p600 python3 tools/bench_general.py --case P600 --steps 5
stats general_path_p200 python3 tools/bench_general.py --case P200 --steps 5
stats rank4_of_8 python3 tools/rank_compute_time.py --partitioned-only 8
bash tools/collect_pmc.sh $out/pmc_traffic_c2.json > $out/pmc_c2.log 2>&1
bash tools/collect_pmc.sh $out/pmc_traffic_c3.json --config C3 > $out/pmc_c3.log 2>&1
bash tools/collect_pmc.sh $out/pmc_traffic_c4.json --config C4 > $out/pmc_c4.log 2>&1
bash tools/collect_pmc.sh $out/pmc_traffic_c3_driver_config.json --config C3 --shared-free 7 --bounds --dogleg 1 > $out/pmc_drv.log 2>&1
bash tools/collect_mfma.sh $out/mfma_utilisation.txt > $out/mfma.log 2>&1
fi
if [ "$1" = fuzz ]; then      # two sweeps against the oracle, seeds as arguments 2 and 3
python3 tools/fuzz_parity.py 300 ${2:-71} > $out/fuzz_parity_300.txt 2>&1
python3 tools/fuzz_parity.py mid 300 ${3:-73} > $out/fuzz_parity_mid_300.txt 2>&1
fi
echo done
