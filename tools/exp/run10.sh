cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03j
BENCH_EXTRA="--h2d-steps 0" bash tools/exp/variants.sh r03j "-DLF_MLE_PAIR64=1" "-DLF_MLE_PAIR64=0"
BENCH_EXTRA="--h2d-steps 0 --unique 16" bash tools/exp/variants.sh r03j_u16 "-DLF_MLE_PAIR64=0"
