#!/usr/bin/env python3
"""tools/vmm_stress.py -- allocate / fill / verify / free balanced observation buffers in a loop: does memory written through
a freshly assembled virtual range always read back?  (tests/test_gpu_mystery.py::test_full_size_sample once read an all-zero
frame from such a buffer after many allocate/free cycles in one process.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
sizes = [32768, 65536, 24576, 49152]
keep = []
for it in range(iters):
    n = sizes[it % len(sizes)]
    try:
        t, info = memory_gym_amd.alloc_obs_buffer((n, 84, 84, 3), torch.uint8, "cuda:0")
    except RuntimeError as e:
        print("iteration %d: %s" % (it, e), flush=True)
        bad += 1
        continue
    v = it % 250 + 1
    t.fill_(v)
    torch.cuda.synchronize()
    wrong = int((t.view(-1, 21168) != v).any(dim=1).sum().item())
    if wrong:
        bad += 1
        rows = (t.view(-1, 21168) != v).any(dim=1).nonzero().flatten()
        print("iteration %d (n=%d, va 0x%x, zones %d): %d frames do not read back; first rows %s" % (it, n, t.data_ptr(), info["zones"], wrong, rows[:8].tolist()), flush=True)
    # a second writer/reader pair: an env handle rasterises into it
    if it % 4 == 0:
        env = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=n, device=0, obs_buffer=t)
        ref = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=n, device=0, obs_placement="plain")
        a, _ = env.reset(seed=0)
        b, _ = ref.reset(seed=0)
        if not torch.equal(a, b):
            bad += 1
            rows = (a.view(n, -1) != b.view(n, -1)).any(dim=1).nonzero().flatten()
            print("iteration %d: %d reset frames differ from the plain buffer's; first %s" % (it, rows.numel(), rows[:8].tolist()), flush=True)
        env.close()
        ref.close()
    if it % 3 == 0:
        keep.append(t)  # some buffers stay alive for a while
    if len(keep) > 3:
        keep.pop(0)
    del t
print("done: %d iterations, %d bad" % (iters, bad))
