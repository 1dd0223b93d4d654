# r02r: fresh processes on a fresh box -- what the zone search finds (bench.py --steps 20 --warmup 5, headline only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
one() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d['obs_placement']; print('$1', round(d['value']/1e6,1), 'M frac', round(d['roofline']['frac'],3), 'zones', p['zones'], 'walked GiB', round(p['searched_bytes']/2**30,1), 'ms', round(p['search_ms']))
"; }
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | one first
rocprofv3 --kernel-trace -d gpurun_out/r02r_kt -o kt -- python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | one rocprof
rm -rf gpurun_out/r02r_kt
for i in 3 4 5; do python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | one p$i; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | one full
