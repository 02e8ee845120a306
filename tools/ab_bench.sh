# Same-box A/B of library variants (box-to-box spread of bench.py is +-8 %: two variants measured on different gpurun boxes say
# nothing about each other).  Variants: ceres_slam_amd/variants/libssba_<name>.so (built by hand from alternative sources, same
# ABI); each is copied over ceres_slam_amd/libssba.so of the BOX'S scratch copy and benched, alternating, ROUNDS times.
#   bash tools/ab_bench.sh "r03 flagged always" 3 [bench.py arguments]
set -e
names=$1; rounds=${2:-2}; shift 2 || true
cp ceres_slam_amd/libssba.so /tmp/libssba_head.so
for r in $(seq 1 $rounds); do
  for n in $names; do
    cp ceres_slam_amd/variants/libssba_$n.so ceres_slam_amd/libssba.so
    python bench.py --no-cpu-baseline --steps 200 "$@" 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r variant $n: %.4f ms/iteration  k_bcr_factor %.4f  its/s %.1f' % (b['ms_per_step'], b['kernel_ms_per_iter'].get('k_bcr_factor', 0.0), b['config']['joint_iters_per_sec']))"
  done
done
cp /tmp/libssba_head.so ceres_slam_amd/libssba.so
