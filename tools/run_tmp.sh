timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_pose_factors.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/b1.json 2> gpurun_out/b1.err; python -c "
import json; j=json.loads(open('gpurun_out/b1.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step']); print(j['kernel_ms_per_iter'])"; done
