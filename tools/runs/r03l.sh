#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --steps 300 --warmup 30 --settle 200 --no-cpu-baseline --no-secondary --no-traffic --no-c1 --env Endless-MysteryPath-v0"
fmt='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("value %.1f M  ms/step %.4f  raster %.1f us  logic %.1f us  zones %s" % (j["value"]/1e6, j["ms_per_step"], r["avg_launch_ms"]*1e3, r["logic_kernel_avg_ms"]*1e3, (j.get("obs_placement") or {}).get("zones")))'
for rep in 1 2; do
  echo "shipped (lb5 svc512) | $($B 2>/dev/null | grep '^{' | python -c "$fmt")" >> gpurun_out/r03l_emp.log
  for L in endless-memory-gym_amd/lib/lab/*.so; do
    echo "$(basename $L) | $(MEMGYM_HIP_LIB=$PWD/$L $B 2>/dev/null | grep '^{' | python -c "$fmt")" >> gpurun_out/r03l_emp.log
  done
done
cat gpurun_out/r03l_emp.log
