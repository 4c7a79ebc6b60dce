cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03q
for V in "$@"; do
LF_EXTRA_CFLAGS="$V" python -m lineslam_amd.build --force > gpurun_out/r03q/build.log 2>&1 || tail -5 gpurun_out/r03q/build.log
echo "== $V"; timeout 600 python tools/pose_prof.py 2>&1 | tail -3
done
