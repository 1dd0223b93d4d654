#!/bin/bash
# tools/store_counters.sh <tag> -- on the GPU box (through gpurun): the L2's memory-side write counters for the store shapes of
# mg_store_probe and for the shipped raster launch, one rocprofv3 pass per counter pair (find the cause of the frame-walk
# vs linear-fill gap with counters, not timings).  Summaries -> gpurun_out/<tag>_store_counters.md (tools/rocpd_summary.py).
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_store_counters.md
echo "# ${TAG} -- store shapes and the shipped raster under the L2's memory-side counters (tools/store_counters.sh)" > $OUT
k=0
for SET in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCP_TCC_WRITE_REQ_sum TCC_WRITE_sum" "TCC_WRITEBACK_sum TCC_TAG_STALL_sum" "TCC_REQ_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_NORMAL_WRITEBACK_sum"; do
  k=$((k+1))
  D=gpurun_out/${TAG}_sc_$k
  timeout 600 rocprofv3 --pmc $SET --kernel-trace -d $D -o p -- python tools/store_counters_run.py > gpurun_out/${TAG}_sc_$k.log 2>&1
  echo; echo "## pass $k: --pmc $SET" >> $OUT; echo >> $OUT
  grep "buffer placement" gpurun_out/${TAG}_sc_$k.log >> $OUT; echo >> $OUT
  DB=$(find $D -name "*_results.db" | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB | grep -v "at::native\|__amd_rocclr\|elementwise_kernel\|mg::zone_probe\|verify\|## " >> $OUT; else echo "(no results: $(tail -3 gpurun_out/${TAG}_sc_$k.log))" >> $OUT; fi
  rm -rf $D
done
cat $OUT | head -150
