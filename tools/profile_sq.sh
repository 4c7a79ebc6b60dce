#!/bin/bash
# Run on the GPU box (through gpurun): SQ / GRBM counter passes (own runs, --kernel-trace only) of ONE serial pass
# of the default workload, to back the "which pipe is each kernel bound by" statements of DESIGN.md with counters.
#   tools/profile_sq.sh <tag>      -> gpurun_out/sq_<tag>/...
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
P3="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"
P4="TCC_HIT_sum TCC_MISS_sum"
P5="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu --no-config4 --no-legs --inflight 1 --h2d-steps 0 ${BENCH_EXTRA:-} > $OUT/pass$i.log 2>&1
  echo "pass $i exit $?" >> $OUT/passes.txt
done
python tools/sq_summary.py $OUT $TAG
find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT
python tools/valu_summary.py $OUT/${TAG}_sq_counters_by_kernel.csv $OUT/${TAG}_valu_utilisation.json
