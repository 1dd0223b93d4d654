#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_mystery.py tests/test_gpu_golden_replay.py tests/test_gpu_c_abi.py tests/test_gpu_full_batch.py tests/test_gpu_option_fuzz.py tests/test_gpu_obs_alloc.py -q 2>&1 | tail -6 > gpurun_out/r03i_tests.log
B="python bench.py --steps 300 --warmup 30 --settle 200 --no-cpu-baseline --no-secondary --no-traffic --no-c1"
fmt='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("value %.1f M  ms/step %.4f  raster %.1f us  logic %.1f us  zones %s" % (j["value"]/1e6, j["ms_per_step"], r["avg_launch_ms"]*1e3, r["logic_kernel_avg_ms"]*1e3, (j.get("obs_placement") or {}).get("zones")))'
for rep in 1 2 3; do for P in 1 0; do
  echo "Endless-MysteryPath-v0 | MEMGYM_SVC_PRIO=$P | $(MEMGYM_SVC_PRIO=$P $B --env Endless-MysteryPath-v0 2>/dev/null | grep '^{' | python -c "$fmt")" >> gpurun_out/r03i_emp.log
done; done
cat gpurun_out/r03i_tests.log gpurun_out/r03i_emp.log
