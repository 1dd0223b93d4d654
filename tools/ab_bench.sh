#!/bin/bash
# tools/ab_bench.sh OUT.tsv ENV_ID N_ENVS 'label|lib|VAR=val VAR=val' ...   -- same-box A/B of library variants / lab switches:
# every configuration runs `bench.py --steps 200` for one workload (no secondary legs) and contributes one line
# (label, M env-steps/s, raster / logic launch averages in us, the five value windows' min-max).  Run the list twice (A/B/A/B) by repeating it.
out=$1; env_id=$2; n=$3; shift 3
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  IFS='|' read -r label lib vars <<< "$cfg"
  [ -z "$lib" ] && lib=endless-memory-gym_amd/lib/lab/libmemgym_hip_lab.so
  line=$(env MEMGYM_HIP_LIB=$PWD/$lib $vars python bench.py --env "$env_id" --envs-per-gpu "$n" --steps 200 --no-cpu-baseline --no-secondary --no-traffic --no-c1 --no-other 2>/dev/null | grep '^{' | tail -1)
  python - "$label" "$line" >> "$out" <<'PY'
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    j = json.loads(line)
    r = j.get("roofline", {})
    w = j.get("value_windows") or [0]
    op = j.get("obs_placement") or {}
    print("%s\t%.1f M\traster %.1f us\tlogic %s us\twindows %.1f-%.1f\tbox frame-shaped %.0f GB/s\tzones %s pieces %s walked %.0f GiB" % (
        label, j["value"] / 1e6, (r.get("avg_launch_ms") or 0) * 1e3,
        ("%.1f" % (r["logic_kernel_avg_ms"] * 1e3)) if r.get("logic_kernel_avg_ms") else "-", min(w) / 1e6, max(w) / 1e6,
        (r.get("box_ceiling_GBps") or {}).get("frame_shaped", 0), op.get("zones"), op.get("pieces"), (op.get("searched_bytes") or 0) / 2**30))
except Exception as e:
    print("%s\tFAILED %s %s" % (label, e, line[:200]))
PY
done
