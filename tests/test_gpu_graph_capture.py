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


@pytest.mark.parametrize("env_id,adim,n_act,K,options", [("MortarMayhem-Grid-v0", 1, 4, 64, None),
                                                         ("Endless-SearingSpotlights-v0", 2, 3, 30, {"max_steps": 9}),
                                                         ("SearingSpotlights-v0", 2, 3, 30, {"max_steps": 9}),
                                                         ("MysteryPath-Grid-v0", 1, 4, 30, {"max_steps": 9}),
                                                         ("Endless-MysteryPath-v0", 1, 4, 30, None)])
def test_step_with_terminal_observations_is_graph_capturable(env_id, adim, n_act, K, options):
    """mg_step with mg_info_buffers.final_obs_dev (the gymnasium vector convention) under capture: the launches that keep terminal
    observations themselves (spotlight and Mystery Path families), the generic path (copy, masked reset, frames by the mask) the mortar
    family falls back to while a graph is being captured, and Endless-MysteryPath's.  No allocation, no synchronisation inside; the replay
    equals the eager run incl. the terminal observations of the instances that finished."""
    import memory_gym_amd
    import torch

    n = 512
    g = torch.Generator(device="cuda").manual_seed(2)
    acts = [torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32) for _ in range(K)]
    eager = memory_gym_amd.VecMemoryGym(env_id, num_envs=n, device=0, final_observation=True)
    eager.reset(seed=9, options=options)
    want, finished = [], 0
    for a in acts:
        o, r, d, _, info = eager.step(a)
        want.append((o.clone(), r.clone(), d.clone(), info["final_observation"].clone()))
        finished += int(d.sum())
    assert finished > 0

    env = memory_gym_amd.VecMemoryGym(env_id, num_envs=n, device=0, final_observation=True)
    env.reset(seed=9, options=options)
    snap = env.state_dict()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.step(acts[0])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    env.load_state_dict(snap)
    outs = []
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for a in acts:
            o, r, d, _, info = env.step(a)
            outs.append((o.clone(), r.clone(), d.clone(), info["final_observation"].clone()))
    env.load_state_dict(snap)
    graph.replay()
    torch.cuda.synchronize()
    for k, ((o1, r1, d1, f1), (o2, r2, d2, f2)) in enumerate(zip(want, outs)):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), "step %d of the replay differs" % k
        assert torch.equal(f1[d1.bool()], f2[d1.bool()]), "terminal observations of step %d differ in the replay" % k
    env.check_errors()
    eager.close()
    env.close()
