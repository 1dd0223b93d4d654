#!/bin/bash
# tools/driver_call_profile.sh <tag> -- run on the GPU box (through gpurun): the driver's bench call, once plain (the line the
# driver records: `roofline.traffic` measured by its own rocprofv3 child passes) and once under `rocprofv3 --kernel-trace --stats`
# (no nested passes then); the summary goes to gpurun_out/<tag>_driver_call.md for profiles/.
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_drv_plain.log 2> gpurun_out/${TAG}_drv_plain.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_drv_kt -o kt -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-c1 > gpurun_out/${TAG}_drv_kt.log 2>&1
{
  echo "# ${TAG} — the driver's call \`python bench.py --gpus 1 --steps 20 --warmup 5\` (1x MI355X, one gpurun call)"; echo
  echo "## A. plain (what the driver records): headline + secondary workloads + C1 + CPU baseline, \`roofline.traffic\` measured by the run's own rocprofv3 child passes"; echo
  echo '```'; grep '^{' gpurun_out/${TAG}_drv_plain.log; echo '```'; echo
  echo "## B. the headline workload alone under \`rocprofv3 --kernel-trace --stats\` (no nested PMC passes under a profiler: traffic falls back to profiles/pmc_latest.json, labelled)"; echo
  echo '```'; grep '^{' gpurun_out/${TAG}_drv_kt.log; echo '```'; echo
  python tools/rocpd_summary.py $(find gpurun_out/${TAG}_drv_kt -name '*_results.db') | grep -v "at::native\|__amd_rocclr\|elementwise_kernel"
} > gpurun_out/${TAG}_driver_call.md
rm -rf gpurun_out/${TAG}_drv_kt
ls -la gpurun_out/${TAG}_driver_call.md
