"""Per-id environment classes under the reference's class names (memory_gym/__init__.py:1-10), so that the gymnasium
registration uses `module:Class` entry points exactly like the reference's (memory_gym/__init__.py:13-61) and
`isinstance(env.unwrapped, GridMortarMayhemEnv)` style checks keep working.  Each class is the single-instance adapter
(`MemoryGymEnv`) bound to one env id; constructor signature as in the reference: `Class(render_mode=None)`."""
from .vec_env import MemoryGymEnv


def _bind(name, env_id, ref):
    def __init__(self, render_mode=None, device=None):
        MemoryGymEnv.__init__(self, env_id, device=device, render_mode=render_mode)

    cls = type(name, (MemoryGymEnv,), {"env_id": env_id, "__init__": __init__, "__module__": __name__,
                                       "__doc__": "%s (reference: memory_gym/%s)" % (env_id, ref)})
    return cls


SearingSpotlightsEnv = _bind("SearingSpotlightsEnv", "SearingSpotlights-v0", "searing_spotlights.py:16")
EndlessSearingSpotlightsEnv = _bind("EndlessSearingSpotlightsEnv", "Endless-SearingSpotlights-v0", "endless_searing_spotlights.py:15")
MortarMayhemEnv = _bind("MortarMayhemEnv", "MortarMayhem-v0", "mortar_mayhem.py:15")
EndlessMortarMayhemEnv = _bind("EndlessMortarMayhemEnv", "Endless-MortarMayhem-v0", "endless_mortar_mayhem.py:15")
GridMortarMayhemEnv = _bind("GridMortarMayhemEnv", "MortarMayhem-Grid-v0", "mortar_mayhem_grid.py:15")
MortarMayhemTaskBEnv = _bind("MortarMayhemTaskBEnv", "MortarMayhemB-v0", "mortar_mayhem_b.py:15")
GridMortarMayhemTaskBEnv = _bind("GridMortarMayhemTaskBEnv", "MortarMayhemB-Grid-v0", "mortar_mayhem_b_grid.py:15")
MysteryPathEnv = _bind("MysteryPathEnv", "MysteryPath-v0", "mystery_path.py:15")
EndlessMysteryPathEnv = _bind("EndlessMysteryPathEnv", "Endless-MysteryPath-v0", "endless_mystery_path.py:16")
GridMysteryPathEnv = _bind("GridMysteryPathEnv", "MysteryPath-Grid-v0", "mystery_path_grid.py:15")

ENTRY_POINTS = {c.env_id: "memory_gym_amd.envs:%s" % c.__name__ for c in (
    SearingSpotlightsEnv, EndlessSearingSpotlightsEnv, MortarMayhemEnv, EndlessMortarMayhemEnv, GridMortarMayhemEnv,
    MortarMayhemTaskBEnv, GridMortarMayhemTaskBEnv, MysteryPathEnv, EndlessMysteryPathEnv, GridMysteryPathEnv)}
CLASSES = {c.env_id: c for c in (
    SearingSpotlightsEnv, EndlessSearingSpotlightsEnv, MortarMayhemEnv, EndlessMortarMayhemEnv, GridMortarMayhemEnv,
    MortarMayhemTaskBEnv, GridMortarMayhemTaskBEnv, MysteryPathEnv, EndlessMysteryPathEnv, GridMysteryPathEnv)}
