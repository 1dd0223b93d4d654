#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03f_tests.log
MEMGYM_OBS_DEBUG=1 python bench.py --steps 20 --warmup 5 > gpurun_out/r03f_bench_driver.json 2> gpurun_out/r03f_bench_driver.err
for N in 16384 65536 262144; do
  python bench.py --env Endless-SearingSpotlights-v0 --envs-per-gpu $N --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-traffic --no-c1 2>/dev/null | grep '^{' | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("ESS n=%d value %.1f M raster %.1f us = %.0f GB/s logic %.1f us zones %s" % (j["config"]["envs_per_gpu"], j["value"]/1e6, r["avg_launch_ms"]*1e3, r["achieved"], r["logic_kernel_avg_ms"]*1e3, (j.get("obs_placement") or {}).get("zones")))' >> gpurun_out/r03f_ess_sizes.log
done
cat gpurun_out/r03f_tests.log; grep -c . gpurun_out/r03f_bench_driver.err; grep 'mg_obs_alloc' gpurun_out/r03f_bench_driver.err | head -20; cat gpurun_out/r03f_ess_sizes.log; python -c '
import json; j=json.loads([l for l in open("gpurun_out/r03f_bench_driver.json") if l.startswith("{")][-1])
print("value %.1f M, ms/step %.4f (wall %.4f), timing: %s" % (j["value"]/1e6, j["ms_per_step"], j["wall_ms_per_step"], j["timing"]))
r=j["roofline"]; print("roofline frac %.3f achieved %.0f traffic %s source %s" % (r["frac"], r["achieved"], r["traffic"], (r["traffic_source"] or "")[:90])); print("traffic passes", json.dumps(r.get("traffic_passes"))[:900])
print("c1", json.dumps(j.get("c1"))[:700]); 
for s in j.get("secondary_workloads", []): print(s["config"], "%.1f M" % (s["value"]/1e6), "reset_share", s.get("reset_share"), json.dumps(s.get("reset_share_detail"))[:300])
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
'
