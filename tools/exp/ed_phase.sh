# EDLines: batch time of the detector on the bench frames (per-kernel times: rocprofv3 --kernel-trace --stats on bench.py --detector edlines)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_ed
python -m lineslam_amd.build >/dev/null 2>&1
python tools/ed_perf.py 1147 2>&1 | grep iter | tee gpurun_out/r04_ed/perf.txt
