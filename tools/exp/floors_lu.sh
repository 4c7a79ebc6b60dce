# pipelined floors (tools/stage_floors.py) with the LDS sweep on / off, then the k_mle phase profile (s_memtime, one work item)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_floors
python -m lineslam_amd.build >/dev/null 2>&1
for LU in 0 1; do echo "== LF_SWEEP_LU=$LU"; LF_SWEEP_LU=$LU python tools/stage_floors.py 2>&1 | grep "ms per pass"; done | tee gpurun_out/r04_floors/floors.txt
for G in 32 64; do
  LF_EXTRA_CFLAGS="-DLF_MLE_PROFILE=$G" python -m lineslam_amd.build --force >/dev/null 2>&1
  python tools/mle_stats.py 16 2>&1 | grep -v amdgpu | tail -8
done | tee gpurun_out/r04_floors/mle_prof.txt
