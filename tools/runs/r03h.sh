#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --steps 300 --warmup 30 --settle 200 --no-cpu-baseline --no-secondary --no-traffic --no-c1"
for E in MysteryPath-Grid-v0 MysteryPath-v0; do
  S=$(echo $E | tr -d '-' | tr 'A-Z' 'a-z')
  for MODE in new old; do
    if [ $MODE = old ]; then X="env MEMGYM_MYSTERY_DEFER=$([ $E = MysteryPath-v0 ] && echo 0 || echo 1) MEMGYM_PATH_HELP=0"; else X="env"; fi
    $X rocprofv3 --kernel-trace --stats -d gpurun_out/r03h_${S}_${MODE} -o kt -- $B --env $E --no-events > gpurun_out/r03h_${S}_${MODE}.log 2>&1
    { echo "## $E, $MODE"; grep '^{' gpurun_out/r03h_${S}_${MODE}.log | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("bench line: value %.1f M env-steps/s, %.4f ms/step, obs_placement zones %s" % (j["value"]/1e6, j["ms_per_step"], (j.get("obs_placement") or {}).get("zones")))'; echo; python tools/rocpd_summary.py $(find gpurun_out/r03h_${S}_${MODE} -name '*_results.db') | grep -v "at::native\|__amd_rocclr\|elementwise_kernel\|^## "; } >> gpurun_out/r03h_mass_resets.md
    rm -rf gpurun_out/r03h_${S}_${MODE}
  done
done
fmt='import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("value %.1f M  ms/step %.4f  raster %.1f us  logic %.1f us  zones %s" % (j["value"]/1e6, j["ms_per_step"], r["avg_launch_ms"]*1e3, r["logic_kernel_avg_ms"]*1e3, (j.get("obs_placement") or {}).get("zones")))'
for E in Endless-MysteryPath-v0 MysteryPath-Grid-v0; do for P in 0 1 0 1; do
  echo "$E | MEMGYM_SVC_PRIO=$P | $(MEMGYM_SVC_PRIO=$P $B --env $E 2>/dev/null | grep '^{' | python -c "$fmt")" >> gpurun_out/r03h_prio.log
done; done
# dynamic instruction mix of the Endless-MysteryPath queue server (separate launch: MEMGYM_EMP_FUSE=0)
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
P2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES"
i=0; for P in "$P1" "$P2"; do i=$((i+1)); MEMGYM_EMP_FUSE=0 rocprofv3 --pmc $P --kernel-trace -d gpurun_out/r03h_sq_p$i -o p -- python bench.py --env Endless-MysteryPath-v0 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-events --no-traffic --no-c1 > gpurun_out/r03h_sq_p$i.log 2>&1; done
python tools/rocpd_summary.py $(find gpurun_out/r03h_sq_p1 gpurun_out/r03h_sq_p2 -name '*_results.db') | grep -v "at::native\|__amd_rocclr\|elementwise_kernel" > gpurun_out/r03h_emp_sq.md
rm -rf gpurun_out/r03h_sq_p1 gpurun_out/r03h_sq_p2
cat gpurun_out/r03h_mass_resets.md; cat gpurun_out/r03h_prio.log; cat gpurun_out/r03h_emp_sq.md
