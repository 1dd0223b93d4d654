#!/bin/bash
# tools/float_formats_profile.sh <tag> LABEL=lib.so[:VAR=value] [LABEL=lib.so ...] -- run on the GPU box (through gpurun): for every library and each of
# the fused float output formats, a rocprofv3 kernel trace of `bench.py --obs-format F` (MortarMayhem-Grid-v0, 65,536 instances) and two
# SQ counter passes (instruction mix, LDS bank conflicts, who waits for what); summaries in gpurun_out/<tag>_<label>_<format>.md.
# PMC passes carry --kernel-trace only (gpurun refuses other trace domains with --pmc).
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM_WR"
FORMATS=${FORMATS:-bf16_chw f16_chw f32_chw u8_xyc}
B="--env MortarMayhem-Grid-v0 --envs-per-gpu 65536 --no-cpu-baseline --no-secondary --no-traffic --no-c1 --no-other --long-window 0"
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}; extra=""
  case "$lib" in *:*) extra=${lib#*:}; lib=${lib%%:*};; esac   # e.g. MEMGYM_OBS_FRAME_BYTES=21168: the piece order of rounds 2-5 (tools/fmt_ab.sh)
  for F in $FORMATS; do
    O=gpurun_out/${TAG}_${label}_${F}
    env $extra MEMGYM_HIP_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats -d ${O}_kt -o kt -- python bench.py $B --obs-format $F --steps 200 --warmup 30 > ${O}_kt.log 2>&1
    env $extra MEMGYM_HIP_LIB=$PWD/$lib rocprofv3 --pmc $P1 --kernel-trace -d ${O}_p1 -o p -- python bench.py $B --obs-format $F --steps 20 --warmup 5 --no-events > ${O}_p1.log 2>&1
    env $extra MEMGYM_HIP_LIB=$PWD/$lib rocprofv3 --pmc $P2 --kernel-trace -d ${O}_p2 -o p -- python bench.py $B --obs-format $F --steps 20 --warmup 5 --no-events > ${O}_p2.log 2>&1
    {
      echo "## ${label}, ${F}"; echo
      echo "bench.py line of the kernel-trace run:"; echo '```'; grep '^{' ${O}_kt.log | python -c 'import json,sys
j=json.loads(sys.stdin.read()); r=j["roofline"]; print(json.dumps({k:j[k] for k in ("value","ms_per_step","dtype")} | {"roofline":{k:r.get(k) for k in ("achieved","frac","avg_launch_ms","bytes_per_launch")}, "obs_placement": j.get("obs_placement")}))'; echo '```'; echo
      python tools/rocpd_summary.py ${O}_kt/kt_results.db ${O}_p1/p_results.db ${O}_p2/p_results.db | grep "raster_kernel\|^| kernel\|^|---\|^## "
      echo
    } > ${O}.md
    tail -3 ${O}_p2.log | grep -i "error\|invalid" >> ${O}.md
    rm -rf ${O}_kt ${O}_p1 ${O}_p2
  done
done
ls -la gpurun_out/${TAG}_*.md
