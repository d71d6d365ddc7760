mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_hip_kernels.py -q -x -k "potrf or chol" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_inversion_gpu.py -q -x -k "optimize or cube32 or cube16 or full_size_64 or logl or drill" 2>&1 | tail -2
python bench.py --size 32 --kernel exp --drill 0 --steps 20 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('32^3', d['ms_per_step'], d['config']['stage_ms_per_step_rank0'])"
} > gpurun_out/leaf4.log 2>&1
