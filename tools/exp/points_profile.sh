# the ORB extractor alone on 1147 frames: per-kernel table (config 3's point front end without the line front end beside it)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_points
python -m lineslam_amd.build >/dev/null 2>&1
sed "s/^B = 256/B = 1147/" tools/exp/orb_time.py > tools/exp/_orb_time_1147.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_points/orb -o orb -- python tools/exp/_orb_time_1147.py > gpurun_out/r04_points/orb.log 2>&1
grep orb_extract gpurun_out/r04_points/orb.log
find gpurun_out/r04_points/orb -name "*kernel_stats.csv" | head -1 | xargs head -10 | cut -c1-110
find gpurun_out/r04_points -name "*kernel_trace.csv" -delete; rm -f tools/exp/_orb_time_1147.py
