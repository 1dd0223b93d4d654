mkdir -p gpurun_out/r02d
for rep in 1 2; do for k in contig reversed halves_il split5 split20 split50 split80 regions4 rr2 rr4 rr8; do python tools/placement_lab.py far $k 2>&1 | grep "far \|Error\|error" >> gpurun_out/r02d/far.log; done; done
cat gpurun_out/r02d/far.log
