timeout -k 10 900 python -m pytest tests/test_gpu_phong_solve.py -x -q -m gpu > gpurun_out/t5.log 2>&1; tail -3 gpurun_out/t5.log
for a in "--shared-free 7" "--shared-free 7 --bounds --dogleg 1"; do timeout -k 10 200 python bench.py --config C3 $a --no-cpu-baseline > gpurun_out/b_c3f.json 2>/dev/null; python -c "
import json; j=json.loads(open('gpurun_out/b_c3f.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step']); print(j['kernel_ms_per_iter'])"; done
