mkdir -p gpurun_out/r02c
ls /sys/kernel/debug/dri/ > gpurun_out/r02c/debugfs.txt 2>&1
ls /sys/kernel/debug/dri/*/ >> gpurun_out/r02c/debugfs.txt 2>&1
rocm-smi --showmemuse --showclocks > gpurun_out/r02c/smi.txt 2>&1
cat /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition >> gpurun_out/r02c/smi.txt 2>&1
for rep in 1 2 3; do for k in torch vmm1 vmm1_exact vmm2 vmm8 vmm21; do python tools/placement_lab.py fresh $k 2>&1 | grep fresh >> gpurun_out/r02c/fresh.log; done; done
cat gpurun_out/r02c/fresh.log
