cd "$GRAFT_REPO_ROOT"
LF_EXTRA_CFLAGS="-DLF_SWEEP_STATS=1 -DLF_SWEEP_PROFILE" python -m lineslam_amd.build --force >/dev/null 2>&1
python tools/lsd_perf.py 1147 40 2>&1 | grep -v amdgpu.ids | tail -6
