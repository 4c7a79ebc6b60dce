cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03e
for G in 32 64; do
LF_EXTRA_CFLAGS="-DLF_MLE_PROFILE=$G" python -m lineslam_amd.build --force > /dev/null 2>&1
python tools/mle_stats.py 4 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03e/mleprof.log
done
