#!/usr/bin/env python3
"""examples/rollout_with_policy.py -- what a trainer's sampling loop looks like on top of memory_gym_amd: a (randomly
initialised) convolutional policy reads the observations where the raster kernel wrote them, its actions go straight
back into mg_step; nothing leaves the GPU.  Two env groups are double-buffered on two HIP streams so that one group's
kernels run under the other group's policy forward.

    python examples/rollout_with_policy.py [--env Endless-MortarMayhem-v0] [--envs 16384] [--steps 200] [--obs-format bf16_chw]

Prints env-steps/s with the policy in the loop (sampling only: no learner, no storage of the rollout)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import memory_gym_amd  # noqa: E402


class Policy(nn.Module):
    """The Atari-style encoder recurrent-PPO baselines use for Memory Gym's 84x84x3 frames, with a categorical head per
    action dimension (no recurrence here: the point is the data path)."""

    def __init__(self, nvec):
        super().__init__()
        self.enc = nn.Sequential(nn.Conv2d(3, 32, 8, 4), nn.ReLU(), nn.Conv2d(32, 64, 4, 2), nn.ReLU(), nn.Conv2d(64, 64, 3, 1), nn.ReLU(),
                                 nn.Flatten(), nn.Linear(64 * 7 * 7, 512), nn.ReLU())
        self.heads = nn.ModuleList([nn.Linear(512, n) for n in nvec])

    @torch.no_grad()
    def act(self, obs):
        h = self.enc(obs)
        return torch.stack([torch.distributions.Categorical(logits=head(h).float()).sample() for head in self.heads], 1).to(torch.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Endless-MortarMayhem-v0")
    ap.add_argument("--envs", type=int, default=16384, help="instances in total (two groups of half that)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--obs-format", default="bf16_chw", choices=["f32_chw", "f16_chw", "bf16_chw"])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    dtype = {"f32_chw": torch.float32, "f16_chw": torch.float16, "bf16_chw": torch.bfloat16}[args.obs_format]
    groups, streams = [], [torch.cuda.Stream(), torch.cuda.Stream()]
    for g in range(2):
        env = memory_gym_amd.make(args.env, num_envs=args.envs // 2, device=0, obs_format=args.obs_format)
        obs, _ = env.reset(seed=torch.arange(g * (args.envs // 2), (g + 1) * (args.envs // 2), dtype=torch.int64, device=dev))
        groups.append([env, obs])
    nvec = [4] if groups[0][0].action_dim == 1 else [3, 3]
    policy = Policy(nvec).to(dev, dtype).eval()
    torch.cuda.synchronize()

    def run(steps):
        returns = []
        for _ in range(steps):
            for g, (env, obs) in enumerate(groups):
                with torch.cuda.stream(streams[g]):  # forward + step of one group overlap the other group's
                    a = policy.act(obs)
                    a = a[:, 0].contiguous() if env.action_dim == 1 else a.contiguous()
                    obs, rew, done, _, info = env.step(a)
                    groups[g][1] = obs
                    returns.append((info["reward"], done))
        torch.cuda.synchronize()
        return returns

    run(10)
    t0 = time.perf_counter()
    ret = run(args.steps)
    dt = time.perf_counter() - t0
    fin = sum(int(d.sum()) for _, d in ret[-2:])
    print("%s: %d instances, policy in the loop (%s observations): %.2f M env-steps/s, %.3f ms per step of all instances; "
          "%d episodes finished in the last step" % (args.env, args.envs, args.obs_format, args.envs * args.steps / dt / 1e6,
                                                     dt / args.steps * 1e3, fin))


if __name__ == "__main__":
    main()
