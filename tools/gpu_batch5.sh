#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_ops.py tests/test_hip_network.py tests/test_hip_ddp.py -m gpu -q -x > gpurun_out/b5_tests.log 2>&1; echo "rc=$?" >> gpurun_out/b5_tests.log; tail -4 gpurun_out/b5_tests.log | cut -c1-250
for v in "A=0" "KODHIP_WAVE_N1=1" "KODHIP_WAVE_N1=2 KODHIP_FORCE_BM=256"; do
  echo "== conv [$v]" >> gpurun_out/b5_conv.log
  env $v timeout -k 10 300 python tools/bench_conv.py >> gpurun_out/b5_conv.log 2>&1
done
grep -v amdgpu.ids gpurun_out/b5_conv.log | cut -c1-160
for v in "A=0" "KODHIP_WAVE_N1=1"; do
  echo "== bench [$v]" >> gpurun_out/b5_bench.log
  env $v timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | cut -c1-160 >> gpurun_out/b5_bench.log
done
cat gpurun_out/b5_bench.log
timeout -k 10 300 python -c "
import cProfile, pstats, sys, runpy
sys.argv=['tools/bench_eval.py']
cProfile.run('runpy.run_path(\"tools/bench_eval.py\", run_name=\"__main__\")', 'gpurun_out/b5_eval.prof')
p=pstats.Stats('gpurun_out/b5_eval.prof'); p.sort_stats('cumulative').print_stats(35)
" > gpurun_out/b5_evalprof.log 2>&1; grep -v amdgpu.ids gpurun_out/b5_evalprof.log | head -70 | cut -c1-150
