for occ in 7 5 4; do for svc in 384 512 768 1024 1536; do
echo -n "occ $occ svc $svc: "; MEMGYM_EMP_OCC=$occ MEMGYM_EMP_SVC=$svc python bench.py --env Endless-MysteryPath-v0 --steps 200 --no-cpu-baseline --no-secondary 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step']*1000,1))"
done; done
