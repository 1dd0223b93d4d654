#!/bin/bash
# tools/profile_round.sh <tag> [env ids ...] -- run on the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes of
# bench.py for the BASELINE configs that fit one GPU; summaries go to gpurun_out/<tag>_*.md via tools/rocpd_summary.py.
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
shift
ENVS=${@:-MortarMayhem-Grid-v0 MysteryPath-v0 Endless-SearingSpotlights-v0 Endless-MortarMayhem-v0 Endless-MysteryPath-v0 SearingSpotlights-v0 MysteryPath-Grid-v0}
for E in $ENVS; do
  S=$(echo $E | tr -d '-' | tr 'A-Z' 'a-z')
  rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_${S}_kt -o kt -- python bench.py --env $E --steps 200 --warmup 30 --no-cpu-baseline --no-secondary --no-traffic --no-c1 > gpurun_out/${TAG}_${S}_kt.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/${TAG}_${S}_w -o w -- python bench.py --env $E --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-events --no-traffic --no-c1 > gpurun_out/${TAG}_${S}_w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/${TAG}_${S}_r -o r -- python bench.py --env $E --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-events --no-traffic --no-c1 > gpurun_out/${TAG}_${S}_r.log 2>&1
  {
    echo "# ${TAG} — $E (bench.py default size, 1x MI355X)"; echo
    echo "bench.py line of the kernel-trace run:"; echo '```'; grep '^{' gpurun_out/${TAG}_${S}_kt.log; echo '```'; echo
    echo "observation-buffer placement of the three passes (kernel trace, --pmc WRITE_SIZE, --pmc FETCH_SIZE; the PMC passes perturb the allocator's probe: profiles/r03_pmc_passes.md):"
    for L in kt w r; do grep '^{' gpurun_out/${TAG}_${S}_${L}.log | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("*", json.dumps(j.get("obs_placement")))'; done; echo
    python tools/rocpd_summary.py gpurun_out/${TAG}_${S}_kt/kt_results.db gpurun_out/${TAG}_${S}_w/w_results.db gpurun_out/${TAG}_${S}_r/r_results.db | grep -v "at::native\|__amd_rocclr\|elementwise_kernel"
  } > gpurun_out/${TAG}_${S}.md
  if [ "$E" = "MortarMayhem-Grid-v0" ]; then
    python tools/pmc_json.py gpurun_out/${TAG}_${S}_w/w_results.db gpurun_out/${TAG}_${S}_r/r_results.db $E 65536 ${TAG}_${S}.md > gpurun_out/pmc_latest.json
  fi
  rm -rf gpurun_out/${TAG}_${S}_kt gpurun_out/${TAG}_${S}_w gpurun_out/${TAG}_${S}_r
done
ls -la gpurun_out/${TAG}_*.md
