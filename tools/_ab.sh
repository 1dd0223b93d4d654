cd $GRAFT_REPO_ROOT
L=endless-memory-gym_amd/lib/lab
rm -f gpurun_out/ab_fd*.tsv
for rep in 1 2; do
for cfg in "shipped|$L/libmemgym_hip_lab.so|" "frames from the end of the buffer|$L/libmemgym_fdown.so|"; do
  bash tools/ab_bench.sh gpurun_out/ab_fd_c2.tsv MortarMayhem-Grid-v0 65536 "$cfg"
  bash tools/ab_bench.sh gpurun_out/ab_fd_c3.tsv MysteryPath-v0 32768 "$cfg"
done; done
for f in c2 c3; do echo $f; cat gpurun_out/ab_fd_$f.tsv | cut -f1-5; done
