#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_mystery.py tests/test_gpu_golden_replay.py tests/test_gpu_c_abi.py tests/test_gpu_checkpoint.py tests/test_gpu_debug_render.py tests/test_gpu_groups.py tests/test_gpu_vector_api.py tests/test_gpu_graph_capture.py tests/test_gpu_obs_format.py -q -k "Mystery or mystery or c_host or checkpoint or graph or format" 2>&1 | tail -8 > gpurun_out/r03k_tests.log
python -m pytest tests/test_gpu_option_fuzz.py tests/test_gpu_full_batch.py -q 2>&1 | tail -4 >> gpurun_out/r03k_tests.log
B="python bench.py --steps 300 --warmup 30 --settle 200 --no-cpu-baseline --no-secondary --no-traffic --no-c1"
fmt='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("value %.1f M  ms/step %.4f  raster %.1f us  logic %.1f us  zones %s" % (j["value"]/1e6, j["ms_per_step"], r["avg_launch_ms"]*1e3, r["logic_kernel_avg_ms"]*1e3, (j.get("obs_placement") or {}).get("zones")))'
for rep in 1 2 3; do for L in 1 0; do
  echo "Endless-MysteryPath-v0 | MEMGYM_EMP_LAZY=$L | $(MEMGYM_EMP_LAZY=$L $B --env Endless-MysteryPath-v0 2>/dev/null | grep '^{' | python -c "$fmt")" >> gpurun_out/r03k_emp.log
done; done
cat gpurun_out/r03k_tests.log gpurun_out/r03k_emp.log
