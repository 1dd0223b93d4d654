#!/bin/bash
# tools/fmt_ab.sh OUT.tsv LABEL LIB.so FORMAT [ENV_ID N] -- one line per run of `bench.py --obs-format FORMAT --steps 200` with library LIB
# (same-box A/B of the stream-out of the float output formats; like tools/ab_bench.sh)
out=$1; label=$2; lib=$3; fmt=$4; envid=${5:-MortarMayhem-Grid-v0}; n=${6:-65536}
line=$(env MEMGYM_HIP_LIB=$PWD/$lib python bench.py --env $envid --envs-per-gpu $n --obs-format $fmt --steps 200 --no-cpu-baseline --no-secondary --no-traffic --no-c1 --no-other --long-window 0 2>/dev/null | grep '^{' | tail -1)
python - "$label" "$fmt" "$line" >> $out <<'PY'
import json, sys
label, fmt, line = sys.argv[1:4]
try:
    j = json.loads(line); r = j.get("roofline", {})
    op = j.get("obs_placement") or {}
    w = j.get("value_windows") or [0]
    print("%s\t%s\t%.1f M\traster %.1f us\t%.0f GB/s\tfrac %.3f\twindows %.1f-%.1f\tzones %s pieces %s walked %.0f GiB\tlinear fill %.0f GB/s" % (
        label, fmt, j["value"]/1e6, (r.get("avg_launch_ms") or 0)*1e3, r.get("achieved") or 0, r.get("frac") or 0, min(w)/1e6, max(w)/1e6,
        op.get("zones"), op.get("pieces"), (op.get("searched_bytes") or 0)/2**30, (r.get("box_ceiling_GBps") or {}).get("linear_fill", 0)))
except Exception as e:
    print("%s\t%s\tFAILED %s %s" % (label, fmt, e, line[:200]))
PY
