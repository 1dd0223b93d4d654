#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r03g_tests.log
# mass resets: kernel traces of runs that cross the lock-step truncation (MysteryPath-Grid: step 128; MysteryPath-v0: step 512)
for E in MysteryPath-Grid-v0 MysteryPath-v0; do
  S=$(echo $E | tr -d '-' | tr 'A-Z' 'a-z')
  rocprofv3 --kernel-trace --stats -d gpurun_out/r03g_${S}_kt -o kt -- python bench.py --env $E --steps 300 --warmup 30 --settle 200 --no-cpu-baseline --no-secondary --no-traffic --no-c1 --no-events > gpurun_out/r03g_${S}.log 2>&1
  { echo "# r03g -- $E, rocprofv3 --kernel-trace --stats of bench.py --steps 300 --warmup 30 --settle 200 (530 steps after the reset)"; echo; grep '^{' gpurun_out/r03g_${S}.log | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("bench line: value %.1f M env-steps/s, %.4f ms/step, obs_placement zones %s" % (j["value"]/1e6, j["ms_per_step"], (j.get("obs_placement") or {}).get("zones")))'; echo; python tools/rocpd_summary.py gpurun_out/r03g_${S}_kt/*/*_results.db 2>/dev/null | grep -v "at::native\|__amd_rocclr\|elementwise_kernel" ; } > gpurun_out/r03g_${S}.md
  rm -rf gpurun_out/r03g_${S}_kt
done
MEMGYM_OBS_DEBUG=1 python bench.py --steps 20 --warmup 5 > gpurun_out/r03g_bench_driver.json 2> gpurun_out/r03g_bench_driver.err
for N in 16384 65536 262144; do
  python bench.py --env Endless-SearingSpotlights-v0 --envs-per-gpu $N --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-traffic --no-c1 2>/dev/null | grep '^{' | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("ESS n=%d value %.1f M raster %.1f us = %.0f GB/s logic %.1f us zones %s" % (j["config"]["envs_per_gpu"], j["value"]/1e6, r["avg_launch_ms"]*1e3, r["achieved"], r["logic_kernel_avg_ms"]*1e3, (j.get("obs_placement") or {}).get("zones")))' >> gpurun_out/r03g_ess_sizes.log
done
cat gpurun_out/r03g_tests.log; cat gpurun_out/r03g_mysterypathgridv0.md gpurun_out/r03g_mysterypathv0.md; grep 'mg_obs_alloc' gpurun_out/r03g_bench_driver.err | head -30; cat gpurun_out/r03g_ess_sizes.log; python -c '
import json; j=json.loads([l for l in open("gpurun_out/r03g_bench_driver.json") if l.startswith("{")][-1])
print("value %.1f M, ms/step %.4f (wall %.4f)" % (j["value"]/1e6, j["ms_per_step"], j["wall_ms_per_step"]))
r=j["roofline"]; print("roofline frac %.3f achieved %.0f traffic %s" % (r["frac"], r["achieved"], r["traffic"])); print("placement", j.get("obs_placement"))
for s in j.get("secondary_workloads", []): print(s["config"], "%.1f M" % (s["value"]/1e6), "zones", s.get("obs_placement_zones"), "reset_share", s.get("reset_share"))
'
