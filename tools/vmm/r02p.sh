mkdir -p gpurun_out/r02p
for lib in hip plain; do for lds in 22176 25600 28672 32768; do for n in 16384 32768; do
  MEMGYM_HIP_LIB=$PWD/endless-memory-gym_amd/lib/libmemgym_$lib.so MEMGYM_RASTER_LDS=$lds python bench.py --env Endless-SearingSpotlights-v0 --envs-per-gpu $n --steps 200 --no-secondary --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$lib lds $lds n $n: %.1f M, raster %.1f us, logic %.1f us, zones %s' % (j['value']/1e6, j['roofline']['avg_launch_ms']*1e3, j['roofline']['logic_kernel_avg_ms']*1e3, (j.get('obs_placement') or {}).get('zones')))" >> gpurun_out/r02p/ess.log
done; done; done
cat gpurun_out/r02p/ess.log
