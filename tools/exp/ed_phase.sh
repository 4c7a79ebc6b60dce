# EDLines back end: how the batch time splits between the anchor walk and split / join / validate (compile-time early exits)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_ed
for V in "-DLF_ED_EXP_WALK_ONLY" "-DLF_ED_EXP_NO_VALIDATE" ""; do
  LF_EXTRA_CFLAGS="$V" python -m lineslam_amd.build --force >/dev/null 2>&1
  echo "== $V"; python tools/ed_perf.py 1147 2>&1 | grep iter | tail -1
done | tee gpurun_out/r04_ed/phase.txt
