"""memory_gym_amd -- MI355X-native batched Memory Gym (hot path of MarcoMeter/endless-memory-gym).

Same environment ids and reset-options dictionaries as the reference (memory_gym/__init__.py:13-61):

    import memory_gym_amd as memory_gym
    env = memory_gym.make("MortarMayhem-Grid-v0", num_envs=65536)     # batched, tensors on the GPU
    env = memory_gym.make("MortarMayhem-Grid-v0")                      # num_envs=1: reference-shaped numpy API

If gymnasium is installed on the host the ids are also registered there (gymnasium.make(id) returns the
single-instance adapter), mirroring the reference's registration side effect on import.
"""
from .reset_params import DEFAULTS, process_reset_params  # noqa: F401
from .vec_env import ENV_IDS, MemoryGymEnv, VecMemoryGym, alloc_obs_buffer  # noqa: F401
from .vector import GymnasiumVectorEnv  # noqa: F401
from . import envs  # noqa: F401,E402



def make(env_id, num_envs=None, device=None, render_mode=None, obs_format="u8_xyc", final_observation=False, obs_buffer=None,
         obs_placement=None, ground_truth64=False, on_capacity="raise", capacity=None):
    if num_envs is None:
        if env_id not in envs.CLASSES:
            raise ValueError("unknown env id %r" % (env_id,))
        return envs.CLASSES[env_id](render_mode=render_mode, device=device)
    return VecMemoryGym(env_id, num_envs=num_envs, device=device, render_mode=render_mode, obs_format=obs_format,
                        final_observation=final_observation, obs_buffer=obs_buffer, obs_placement=obs_placement,
                        ground_truth64=ground_truth64, on_capacity=on_capacity, capacity=capacity)


def _register_with_gymnasium():
    """The reference registers its ids as a side effect of `import memory_gym` (memory_gym/__init__.py:13-61), with
    gymnasium's defaults (passive env checker and order enforcing on, no TimeLimit); so does this package."""
    try:
        from gymnasium.envs.registration import register
    except Exception:
        return
    for env_id, entry_point in envs.ENTRY_POINTS.items():
        try:
            register(id=env_id, entry_point=entry_point)
        except Exception:  # e.g. the reference package registered the id already in this interpreter
            pass


_register_with_gymnasium()
