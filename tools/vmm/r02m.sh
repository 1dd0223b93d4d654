mkdir -p gpurun_out/r02m
python -m pytest tests/test_gpu_render_again.py -x -q 2>&1 | tail -3
for i in 1 2 3 4; do MEMGYM_OBS_DEBUG=1 python tools/settle_probe.py MortarMayhem-Grid-v0 65536 12 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02m/settle.log; done
for e in "MysteryPath-v0 32768" "Endless-MortarMayhem-v0 32768" "Endless-SearingSpotlights-v0 16384" "Endless-SearingSpotlights-v0 32768"; do  python tools/settle_probe.py $e 12 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02m/settle.log; MEMGYM_OBS_PLACEMENT=plain python tools/settle_probe.py $e 12 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02m/settle.log; done
cut -c1-330 gpurun_out/r02m/settle.log
