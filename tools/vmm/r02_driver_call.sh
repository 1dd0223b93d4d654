# profiles/r02_driver_call.md: the driver's bench command under rocprofv3 --kernel-trace --stats, and the same headline
# without the secondary workloads (whose Endless-MortarMayhem raster is the same kernel symbol as the headline's)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_driver_call.md
rocprofv3 --kernel-trace --stats -d gpurun_out/drv_a -o kt -- python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/drv_a.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/drv_b -o kt -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/drv_b.log 2>&1
{
echo "# r02 — the driver's call under \`rocprofv3 --kernel-trace --stats\` (1x MI355X, fresh box)"; echo
echo "## A. \`python bench.py --gpus 1 --steps 20 --warmup 5\` (headline + secondary workloads + CPU baseline in one process)"; echo
echo '```'; grep '^{' gpurun_out/drv_a.log; echo '```'; echo
python tools/rocpd_summary.py gpurun_out/drv_a/kt_results.db | grep -v "at::native\|__amd_rocclr\|elementwise_kernel"
echo; echo "(\`raster_kernel<MortarComposer>\` is the raster of the headline AND of the Endless-MortarMayhem secondary workload at half the"
echo "size: its average mixes the two; B isolates the headline.)"; echo
echo "## B. the same with \`--no-secondary --no-cpu-baseline\`: every \`raster_kernel<MortarComposer>\` launch is the headline's"; echo
echo '```'; grep '^{' gpurun_out/drv_b.log; echo '```'; echo
python tools/rocpd_summary.py gpurun_out/drv_b/kt_results.db | grep -v "at::native\|__amd_rocclr\|elementwise_kernel"
} > $OUT
rm -rf gpurun_out/drv_a gpurun_out/drv_b
wc -l $OUT
