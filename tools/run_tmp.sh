timeout -k 10 600 python tools/bench_general.py --case border
timeout -k 10 200 python bench.py --config C3 --shared-free 7 --no-cpu-baseline > gpurun_out/b_c3f.json 2>/dev/null; python -c "
import json; j=json.loads(open('gpurun_out/b_c3f.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step']); print(j['kernel_ms_per_iter'])"
timeout -k 10 200 python bench.py --config C3 --shared-free 7 --bounds --dogleg 1 --no-cpu-baseline > gpurun_out/b_c3d.json 2>/dev/null; python -c "
import json; j=json.loads(open('gpurun_out/b_c3d.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step']); print(j['kernel_ms_per_iter'])"
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "closure or c3" 2>&1 | tail -2
