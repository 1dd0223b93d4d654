cd $GRAFT_REPO_ROOT
L=endless-memory-gym_amd/lib/lab
rm -f gpurun_out/ab_lim.tsv
for rep in 1 2; do
for cfg in "shipped|$L/libmemgym_hip_lab.so|" "cleared frame instead of template|$L/libmemgym_notmpl1.so|" "no template (stale LDS)|$L/libmemgym_notmpl2.so|" "no stamps|$L/libmemgym_nostamps.so|" "no compose at all|$L/libmemgym_nothing.so|"; do
  bash tools/ab_bench.sh gpurun_out/ab_lim.tsv MortarMayhem-Grid-v0 65536 "$cfg"
done; done
cat gpurun_out/ab_lim.tsv
