#!/bin/bash
# tools/pmc_sq.sh <env-id> <tag> -- SQ counter passes (instruction mix / stall buckets) for one bench.py workload.
# Run on the GPU box through gpurun; PMC passes carry --kernel-trace only (gpurun refuses other trace domains with --pmc).
set -u
E=${1:-Endless-SearingSpotlights-v0}
TAG=${2:-sq}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
P2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace -d gpurun_out/${TAG}_p$i -o p -- python bench.py --env $E --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-events --no-traffic --no-c1 > gpurun_out/${TAG}_p$i.log 2>&1
done
python tools/rocpd_summary.py gpurun_out/${TAG}_p1/p_results.db gpurun_out/${TAG}_p2/p_results.db | grep -v "at::native\|__amd_rocclr\|elementwise_kernel" > gpurun_out/${TAG}.md
rm -rf gpurun_out/${TAG}_p1 gpurun_out/${TAG}_p2
