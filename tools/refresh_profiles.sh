# Re-measures everything under profiles/ that comes from bench.py / tools on the GPU box (run from the repo root):
#   bash tools/refresh_profiles.sh        -> writes gpurun_out/prof_*; copy what is to be judged into profiles/
set -x
root=$(pwd)
mkdir -p gpurun_out
python3 bench.py > gpurun_out/prof_c2.json 2> gpurun_out/prof_c2.err
python3 bench.py --config C3 --no-cpu-baseline > gpurun_out/prof_c3.json 2> gpurun_out/prof_c3.err
python3 bench.py --config C3 --shared-free 7 --no-cpu-baseline > gpurun_out/prof_c3_free.json 2> gpurun_out/prof_c3_free.err
python3 bench.py --config C3 --shared-free 7 --bounds --dogleg 1 --no-cpu-baseline > gpurun_out/prof_c3_driver.json 2> gpurun_out/prof_c3_driver.err
python3 bench.py --config C5 > gpurun_out/prof_c5.json 2> gpurun_out/prof_c5.err
python3 tools/rank_compute_time.py 1 2 4 8 > gpurun_out/prof_rank.json 2> gpurun_out/prof_rank.err
python3 tools/bench_general.py > gpurun_out/prof_general.json 2> gpurun_out/prof_general.err
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_rp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rp -o s -- python3 bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_rp.log 2>&1
cp $(find gpurun_out/prof_rp -name '*kernel_stats.csv' | head -1) gpurun_out/prof_kernel_stats_c2.csv
bash tools/collect_pmc.sh gpurun_out/prof_pmc_c2.json > gpurun_out/prof_pmc.log 2>&1
bash tools/collect_mfma.sh gpurun_out/prof_mfma_c2.txt > gpurun_out/prof_mfma.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_c3f
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3f -o s -- python3 bench.py --config C3 --shared-free 7 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_c3f.log 2>&1
cp $(find gpurun_out/prof_c3f -name '*kernel_stats.csv' | head -1) gpurun_out/prof_kernel_stats_c3_free.csv
python3 bench.py --config C4 --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/prof_c4.json 2> gpurun_out/prof_c4.err
echo done
