#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_mystery.py tests/test_gpu_golden_replay.py tests/test_gpu_c_abi.py tests/test_gpu_full_batch.py tests/test_gpu_option_fuzz.py tests/test_gpu_obs_alloc.py tests/test_gpu_debug_render.py tests/test_gpu_groups.py -q 2>&1 | tail -6 > gpurun_out/r03j_tests.log
B="python bench.py --steps 300 --warmup 30 --settle 200 --no-cpu-baseline --no-secondary --no-traffic --no-c1"
fmt='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("value %.1f M  ms/step %.4f  raster %.1f us  logic %.1f us  zones %s" % (j["value"]/1e6, j["ms_per_step"], r["avg_launch_ms"]*1e3, r["logic_kernel_avg_ms"]*1e3, (j.get("obs_placement") or {}).get("zones")))'
for rep in 1 2 3; do for P in 1 0; do
  echo "Endless-MysteryPath-v0 | MEMGYM_SVC_PRIO=$P | $(MEMGYM_SVC_PRIO=$P $B --env Endless-MysteryPath-v0 2>/dev/null | grep '^{' | python -c "$fmt")" >> gpurun_out/r03j_emp.log
done; done
for E in MysteryPath-Grid-v0 MysteryPath-v0; do
  S=$(echo $E | tr -d '-' | tr 'A-Z' 'a-z')
  for MODE in 1 2; do
    MEMGYM_PATH_HELP=$MODE rocprofv3 --kernel-trace --stats -d gpurun_out/r03j_${S}_${MODE} -o kt -- $B --env $E --no-events > gpurun_out/r03j_${S}_${MODE}.log 2>&1
    { echo "## $E, MEMGYM_PATH_HELP=$MODE (1 = lane-per-path generator for long queues, 2 = cooperative generator on all resident workgroups)"; grep '^{' gpurun_out/r03j_${S}_${MODE}.log | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("bench line: value %.1f M env-steps/s, %.4f ms/step, obs_placement zones %s" % (j["value"]/1e6, j["ms_per_step"], (j.get("obs_placement") or {}).get("zones")))'; echo; python tools/rocpd_summary.py $(find gpurun_out/r03j_${S}_${MODE} -name '*_results.db') | grep -v "at::native\|__amd_rocclr\|elementwise_kernel\|^## \|verify_\|zone_probe\|init_kernel"; } >> gpurun_out/r03j_mass_resets.md
    rm -rf gpurun_out/r03j_${S}_${MODE}
  done
done
cat gpurun_out/r03j_tests.log gpurun_out/r03j_emp.log gpurun_out/r03j_mass_resets.md
