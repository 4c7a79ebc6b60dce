cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03o
BENCH_EXTRA="--h2d-steps 0" bash tools/exp/variants.sh r03o "" "-DLF_NI_IMPROVE=__noinline__" "-DLF_NI_IMPROVE=__noinline__ -DLF_NI_R2R=__noinline__" "-DLF_NI_IMPROVE=__noinline__ -DLF_NI_GROW=__noinline__"
