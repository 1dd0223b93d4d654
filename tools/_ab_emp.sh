cd $GRAFT_REPO_ROOT
L=endless-memory-gym_amd/lib/lab
rm -f gpurun_out/ab_emp_*.tsv
for rep in 1 2; do
for cfg in "pre|$L/libmemgym_prewalk.so|" "new|$L/libmemgym_hip_lab.so|"; do
  bash tools/ab_bench.sh gpurun_out/ab_emp_follow.tsv Endless-MysteryPath-v0 32768 "$cfg MEMGYM_BENCH_POLICY=follower:0.02"
  bash tools/ab_bench.sh gpurun_out/ab_emp_random.tsv Endless-MysteryPath-v0 32768 "$cfg"
done; done
echo follower; cat gpurun_out/ab_emp_follow.tsv; echo random; cat gpurun_out/ab_emp_random.tsv
