mkdir -p gpurun_out/r2s
python -m pytest tests/test_gpu_batch.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize_batch.py tests/test_gpu_fcpe.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r2s/pytest.log 2>&1
tail -4 gpurun_out/r2s/pytest.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-roofline > gpurun_out/r2s/c2.json 2>gpurun_out/r2s/c2.err; python -c "
import json; d=json.load(open('gpurun_out/r2s/c2.json')); print('c2', d['value'], d['ms_per_step'], d['stage_ms'])"
python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r2s/c3.json 2>gpurun_out/r2s/c3.err; python -c "
import json; d=json.load(open('gpurun_out/r2s/c3.json')); print('c3', d['value'], d['ms_per_step'], d['stage_ms'])"
python bench.py --batch 8 --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline > gpurun_out/r2s/b8.json 2>gpurun_out/r2s/b8.err; python -c "
import json; d=json.load(open('gpurun_out/r2s/b8.json')); print('b8', d['value'], d['ms_per_step'], d['stage_ms'])"
