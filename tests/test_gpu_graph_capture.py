"""GPU (-m gpu): mg_step only enqueues kernels on the caller's stream (no allocation, no synchronisation), so a trainer can
capture it in a HIP graph; the replay must reproduce the eager results exactly."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id,adim,n_act", [("MortarMayhem-Grid-v0", 1, 4), ("Endless-SearingSpotlights-v0", 2, 3),
                                                ("MysteryPath-Grid-v0", 1, 4), ("Endless-MysteryPath-v0", 1, 4)])
def test_step_is_graph_capturable(env_id, adim, n_act):
    import memory_gym_amd
    import torch

    n, K = 512, 12
    g = torch.Generator(device="cuda").manual_seed(2)
    acts = [torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32) for _ in range(K)]
    eager = memory_gym_amd.make(env_id, num_envs=n, device=0)
    eager.reset(seed=9)
    want = []
    for a in acts:
        o, r, d, _, _ = eager.step(a)
        want.append((o.clone(), r.clone(), d.clone()))

    env = memory_gym_amd.make(env_id, num_envs=n, device=0)
    env.reset(seed=9)
    snap = env.state_dict()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm-up on a side stream, as torch's graph recipe asks
        env.step(acts[0])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    env.load_state_dict(snap)
    outs = []
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for a in acts:
            o, r, d, _, _ = env.step(a)
            outs.append((o.clone(), r.clone(), d.clone()))
    env.load_state_dict(snap)  # capture does not execute: state is still the snapshot; make that explicit
    graph.replay()
    torch.cuda.synchronize()
    for k, ((o1, r1, d1), (o2, r2, d2)) in enumerate(zip(want, outs)):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), "step %d of the replay differs" % k
    env.check_errors()
    eager.close()
    env.close()
