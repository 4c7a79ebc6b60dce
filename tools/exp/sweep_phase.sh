# sweep phase profile (s_memtime per phase) + work counters on the bench frames: bash tools/exp/sweep_phase.sh [tag]
cd "$GRAFT_REPO_ROOT"
tag=${1:-base}
mkdir -p gpurun_out/r04_$tag
LF_EXTRA_CFLAGS="-DLF_SWEEP_STATS=1 -DLF_SWEEP_PROFILE" python -m lineslam_amd.build --force >/dev/null 2>&1
python tools/lsd_perf.py 1147 40 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r04_$tag/phase.txt
python -m lineslam_amd.build --force >/dev/null 2>&1
python tools/lsd_perf.py 1147 40 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r04_$tag/plain.txt
cat gpurun_out/r04_$tag/phase.txt gpurun_out/r04_$tag/plain.txt
