mkdir -p gpurun_out/r02h
for i in 1 2 3 4 5; do MEMGYM_OBS_SEARCH_GB=200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary 2>&1 | grep "^{" >> gpurun_out/r02h/bench.log; done
python - <<PY
import json
for l in open("gpurun_out/r02h/bench.log"):
    j=json.loads(l); p=j.get("obs_placement") or {}
    print(round(j["value"]/1e6,1), round(j["roofline"]["avg_launch_ms"]*1e3,1), round(j["roofline"]["frac"],3), "zones", p.get("zones"), "searched GiB", p.get("searched_bytes",0)/2**30, "same %.2f cross %.2f ms %.0f" % (p.get("probe_same_tbps",0), p.get("probe_cross_tbps",0), p.get("search_ms",0)))
PY
