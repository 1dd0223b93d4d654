cd $GRAFT_REPO_ROOT
L=endless-memory-gym_amd/lib/lab
for rep in 1 2; do
for cfg in "base|$L/libmemgym_hip_lab.so|" "xcd|$L/libmemgym_xcd.so|" "stride|$L/libmemgym_stride.so|MEMGYM_BENCH_OBS_PAD_FRAMES=400" "stridexcd|$L/libmemgym_stridexcd.so|MEMGYM_BENCH_OBS_PAD_FRAMES=400"; do
  bash tools/ab_bench.sh gpurun_out/ab_store.tsv MortarMayhem-Grid-v0 65536 "$cfg"
done; done
for cfg in "base|$L/libmemgym_hip_lab.so|" "xcd|$L/libmemgym_xcd.so|" "stride|$L/libmemgym_stride.so|MEMGYM_BENCH_OBS_PAD_FRAMES=400" "base|$L/libmemgym_hip_lab.so|" "xcd|$L/libmemgym_xcd.so|" "stride|$L/libmemgym_stride.so|MEMGYM_BENCH_OBS_PAD_FRAMES=400"; do
  bash tools/ab_bench.sh gpurun_out/ab_store_c3.tsv MysteryPath-v0 32768 "$cfg"
  bash tools/ab_bench.sh gpurun_out/ab_store_c5.tsv Endless-MortarMayhem-v0 32768 "$cfg"
  bash tools/ab_bench.sh gpurun_out/ab_store_c4.tsv Endless-SearingSpotlights-v0 16384 "$cfg"
done
cat gpurun_out/ab_store.tsv; echo; cat gpurun_out/ab_store_c3.tsv; echo; cat gpurun_out/ab_store_c5.tsv; echo; cat gpurun_out/ab_store_c4.tsv
