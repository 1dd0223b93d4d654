mkdir -p gpurun_out/r02k
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02k/b$i.log 2>&1; grep "^{" gpurun_out/r02k/b$i.log >> gpurun_out/r02k/bench.log; done
MEMGYM_OBS_PLACEMENT=plain python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" >> gpurun_out/r02k/bench.log
python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | grep "^{" >> gpurun_out/r02k/bench.log
python - <<PY
import json
for l in open("gpurun_out/r02k/bench.log"):
    j=json.loads(l); p=j.get("obs_placement") or {}
    print(round(j["value"]/1e6,1), round(j["roofline"]["avg_launch_ms"]*1e3,1), round(j["roofline"]["frac"],3), "zones", p.get("zones"), "searched GiB %.1f" % (p.get("searched_bytes",0)/2**30), "ms %.0f" % (p.get("search_ms",0)), " | ", "  ".join("%s %.1fM %.0fus" % (w["config"], w["value"]/1e6, w["raster_avg_ms"]*1e3) for w in j.get("secondary_workloads", [])))
PY
