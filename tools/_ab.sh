cd $GRAFT_REPO_ROOT
L=endless-memory-gym_amd/lib/lab
rm -f gpurun_out/ab_bo*.tsv
for rep in 1 2; do
for cfg in "shipped (plain path downwards)|$L/libmemgym_hip_lab.so|" "buffer stores downwards too|$L/libmemgym_bufdown.so|"; do
  bash tools/ab_bench.sh gpurun_out/ab_bo_c4.tsv Endless-SearingSpotlights-v0 16384 "$cfg"
  bash tools/ab_bench.sh gpurun_out/ab_bo_ss.tsv SearingSpotlights-v0 16384 "$cfg"
  bash tools/ab_bench.sh gpurun_out/ab_bo_ess64.tsv Endless-SearingSpotlights-v0 65536 "$cfg"
done
for cfg in "upwards (rounds 1-5)|$L/libmemgym_up.so|" "shipped (downwards)|$L/libmemgym_hip_lab.so|"; do
  bash tools/ab_bench.sh gpurun_out/ab_bo_c2.tsv MortarMayhem-Grid-v0 65536 "$cfg"
  bash tools/ab_bench.sh gpurun_out/ab_bo_mm.tsv MortarMayhem-v0 65536 "$cfg"
  bash tools/ab_bench.sh gpurun_out/ab_bo_mpg.tsv MysteryPath-Grid-v0 32768 "$cfg"
done
done
for f in c4 ss ess64 c2 mm mpg; do echo $f; cat gpurun_out/ab_bo_$f.tsv | cut -f1-5; done
