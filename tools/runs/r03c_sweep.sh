#!/bin/bash
# gpurun script: WaveRng parity (mystery family) + sweep of persistent grid x store kind x resident workgroups for the spotlight raster
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_mystery.py tests/test_gpu_golden_replay.py -x -q -k "Mystery" 2>&1 | tail -4 > gpurun_out/r03c_tests.log
python -m pytest tests/test_gpu_full_batch.py -x -q -k "Mystery" 2>&1 | tail -3 >> gpurun_out/r03c_tests.log
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --no-traffic --no-c1"
fmt='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("value %.1f M  ms/step %.4f  raster %.1f us  logic %.1f us  zones %s" % (j["value"]/1e6, j["ms_per_step"], r["avg_launch_ms"]*1e3, r["logic_kernel_avg_ms"]*1e3, (j.get("obs_placement") or {}).get("zones")))'
for E in Endless-MysteryPath-v0 MysteryPath-Grid-v0 MysteryPath-v0; do
  line=$($B --env $E 2>/dev/null | grep '^{'); echo "$E | default | $(echo $line | python -c "$fmt")" >> gpurun_out/r03c_ab.log
done
E=Endless-SearingSpotlights-v0
for NT in 1 0; do for LDS in 22752 28672; do for GRID in 1280 1792 2560 3584 7168 14336; do
  line=$(env MEMGYM_RASTER_NT=$NT MEMGYM_RASTER_LDS=$LDS MEMGYM_RASTER_GRID=$GRID $B --env $E 2>/dev/null | grep '^{')
  echo "$E | NT=$NT LDS=$LDS GRID=$GRID | $(echo $line | python -c "$fmt")" >> gpurun_out/r03c_ab.log
done; done; done
cat gpurun_out/r03c_tests.log; cat gpurun_out/r03c_ab.log
