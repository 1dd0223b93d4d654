mkdir -p gpurun_out/r02j
python -m pytest tests/test_gpu_render_again.py -x -q 2>&1 | tail -30
for i in 1 2 3 4; do MEMGYM_OBS_DEBUG=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/r02j/b$i.log 2>&1; grep "^{" gpurun_out/r02j/b$i.log >> gpurun_out/r02j/bench.log; done
grep -h "mg_obs_alloc" gpurun_out/r02j/b1.log | tail -25
python - <<PY
import json
for l in open("gpurun_out/r02j/bench.log"):
    j=json.loads(l); p=j.get("obs_placement") or {}
    print(round(j["value"]/1e6,1), round(j["roofline"]["avg_launch_ms"]*1e3,1), round(j["roofline"]["frac"],3), "zones", p.get("zones"), "searched GiB %.1f" % (p.get("searched_bytes",0)/2**30), "same %.2f cross %.2f ms %.0f" % (p.get("probe_same_tbps",0), p.get("probe_cross_tbps",0), p.get("search_ms",0)))
PY
