for lazy in 0 1; do for lab in 0 1 2 3; do for svc in 256 512 1024; do
echo -n "lazy $lazy lab $lab svc $svc: "; MEMGYM_EMP_LAZY=$lazy MEMGYM_EMP_LAB=$lab MEMGYM_EMP_SVC=$svc python bench.py --env Endless-MysteryPath-v0 --steps 200 --no-cpu-baseline --no-secondary 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step']*1000,1), round(d['roofline']['avg_launch_ms']*1000,1))"
done; done; done
