timeout -k 10 300 python -m pytest tests/test_gpu_phong_solve.py tests/test_gpu_general_structure.py -x -q -m gpu -k "free or border or closure or driver" > gpurun_out/t4.log 2>&1; tail -3 gpurun_out/t4.log
timeout -k 10 300 python tools/bench_general.py --case C2_loop_closure_border
timeout -k 10 200 python bench.py --config C3 --shared-free 7 --no-cpu-baseline > gpurun_out/b_c3f.json 2>/dev/null; python -c "
import json; j=json.loads(open('gpurun_out/b_c3f.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step']); print(j['kernel_ms_per_iter'])"
