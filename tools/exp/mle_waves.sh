#!/bin/bash
# GPU box: parity of the front end with the lane-distributed LU, then k_mle at 2 / 3 / 4 wavefronts per SIMD
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_front_gpu.py tests/test_relmotion_gpu.py tests/test_pose_golden_gpu.py tests/test_operators_gpu.py -x -q -m gpu > gpurun_out/r03a/tests.log 2>&1; tail -5 gpurun_out/r03a/tests.log
for W in 3 2 4; do
  LF_EXTRA_CFLAGS="-DLF_MLE_WAVES=$W" python -m lineslam_amd.build --force > /dev/null 2>&1
  echo "== waves $W"; timeout 300 python tools/mle_stats.py 64 2>&1 | tail -1
  timeout 600 python bench.py --no-cpu --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined', d['value'], d['ms_per_step'])"
  timeout 600 python bench.py --no-cpu --steps 4 --warmup 2 --inflight 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial', d['value'], d['ms_per_step'])"
done 2>&1 | tee gpurun_out/r03a/waves.log
