#!/bin/bash
# GPU-box check: parity tests, then bench with the per-kernel table.  usage: tools/gpu_check.sh TAG [pytest -k expr]
TAG=$1; K=${2:-""}
timeout -k 10 600 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_s1_tail.py tests/test_gpu_ingest.py tests/test_gpu_train.py -m gpu -q -x --timeout 300 --timeout-method thread ${K:+-k "$K"} > gpurun_out/t_$TAG.log 2>&1
rc=$?; echo "tests exit $rc"; tail -3 gpurun_out/t_$TAG.log
[ $rc -ne 0 ] && { grep -E "Error|error|assert" gpurun_out/t_$TAG.log | head -20; exit $rc; }
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --trace-all --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench exit $?"
python -c "import json;d=json.load(open('gpurun_out/bench_$TAG.json'));print('cells/s',round(d['value'],1),'ms/step',round(d['ms_per_step'],3),'loss',d['final_loss'])"
grep -v -E "amdgpu.ids|Warn|run_backward" gpurun_out/bench_$TAG.err | head -${3:-20}
