mkdir -p gpurun_out/r02o
for i in 1 2 3; do MEMGYM_OBS_DEBUG=1 python bench.py --env MysteryPath-v0 --steps 200 --no-secondary --no-cpu-baseline > gpurun_out/r02o/c3_$i.log 2>&1; done
MEMGYM_OBS_DEBUG=1 python bench.py --steps 100 --no-cpu-baseline > gpurun_out/r02o/seq.log 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r02o/*.log")):
    lines=open(f).read().splitlines()
    j=json.loads([l for l in lines if l.startswith("{")][-1])
    print(f, round(j["value"]/1e6,1), round(j["roofline"]["avg_launch_ms"]*1e3,1), j.get("obs_placement",{}).get("zones"), [ (w["config"], round(w["raster_avg_ms"]*1e3,1), w["obs_placement_zones"]) for w in j.get("secondary_workloads",[])])
    for l in lines:
        if l.startswith("mg_obs_alloc"): print("   ", l)
PY
