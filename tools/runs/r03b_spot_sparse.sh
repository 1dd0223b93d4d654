#!/bin/bash
# gpurun script: sparse spotlight compose -- parity first, then A/B of store kind x resident workgroups (bench.py, 1 GPU)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_spot.py tests/test_gpu_sticky_background.py tests/test_gpu_obs_format.py tests/test_gpu_debug_render.py -x -q 2>&1 | tail -5 > gpurun_out/r03b_tests.log
python -m pytest tests/test_gpu_full_batch.py -x -q -k "Spot" 2>&1 | tail -3 >> gpurun_out/r03b_tests.log
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --no-traffic --no-c1"
for E in Endless-SearingSpotlights-v0 SearingSpotlights-v0; do
  for V in "NT=1" "NT=0" "NT=0 LDS=28672" "NT=1 LDS=22752" "NT=0 LDS=24576" "NT=1 LDS=32768"; do
    nt=$(echo $V | sed -n 's/.*NT=\([01]\).*/\1/p'); lds=$(echo $V | sed -n 's/.*LDS=\([0-9]*\).*/\1/p')
    for rep in 1 2; do
      line=$(MEMGYM_RASTER_NT=$nt ${lds:+MEMGYM_RASTER_LDS=$lds} $B --env $E 2>/dev/null | grep '^{')
      echo "$E | $V | rep $rep | $(echo $line | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("value %.1f M  ms/step %.4f  raster %.1f us  logic %.1f us  zones %s" % (j["value"]/1e6, j["ms_per_step"], r["avg_launch_ms"]*1e3, r["logic_kernel_avg_ms"]*1e3, (j.get("obs_placement") or {}).get("zones")))')" >> gpurun_out/r03b_ab.log
    done
  done
done
cat gpurun_out/r03b_tests.log; cat gpurun_out/r03b_ab.log
