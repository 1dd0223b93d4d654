#!/usr/bin/env python3
"""Capture golden LOGIC trajectories from the unmodified reference (/root/reference/memory_gym),
imported under the functional pygame/gymnasium shims of `ref_shims.py`.

Runs ONLY in the build container (where /root/reference exists):

    python tests/golden/make_golden.py            # writes tests/golden/logic_<env>.npz

The fixtures are DATA (inputs: seeds, options, actions; outputs: per-row state snapshots, rewards,
dones, infos and the numpy PCG64 state after every call).  No reference source text is stored.

Row format (one row per API call):
    kind    : 0 = reset(seed=seed_or_-1 -> None), 1 = step(action)
    seed    : seed passed to reset (-1 = None, i.e. continue the RNG stream)
    action  : [a0, a1] (Discrete envs use a0)
    reward  : float64 step reward (0 for reset rows)
    done    : 0/1
    snap    : float64 [K] named state fields (names in `fields`)
    rng     : uint64 [6] = PCG64 state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger
    lists   : variable-length per-row payloads (command lists, paths, spotlights) padded with -1/NaN
"""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
import ref_shims  # noqa: E402

ref_shims.install()
import memory_gym  # noqa: E402,F401
from gymnasium.envs.registration import registry  # noqa: E402
import memory_gym.character_controller as cc  # noqa: E402
import memory_gym.pygame_assets as pa  # noqa: E402

# ---- instrumentation that does not change behaviour ------------------------------------------
_orig_sprites = cc.CharacterController.create_character_sprites


def _tagged_sprites(self):
    s = _orig_sprites(self)
    for k, surf in enumerate(s):
        surf._sprite_idx = k
    return s


cc.CharacterController.create_character_sprites = _tagged_sprites

LAST_GLYPH = [None]
_orig_cmd_init = pa.Command.__init__


def _cmd_init(self, command_type, scale):
    LAST_GLYPH[0] = command_type
    _orig_cmd_init(self, command_type, scale)


pa.Command.__init__ = _cmd_init

CMD_IDS = {k: i for i, k in enumerate(pa.Command.COMMANDS.keys())}
CMD_IDS[""] = 9  # blank glyph
CMD_DELTA = list(pa.Command.COMMANDS.values())


def make(env_id):
    mod, cls = registry[env_id].split(":")
    return getattr(importlib.import_module(mod), cls)()


def rng_words(env):
    st = env.np_random.bit_generator.state
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return np.array([s >> 64, s & m, inc >> 64, inc & m, st["has_uint32"], st["uinteger"]], dtype=np.uint64)


def glyph_id():
    g = LAST_GLYPH[0]
    return -1 if g is None else CMD_IDS[g]


def sprite_of(env):
    s = getattr(env, "rotated_agent_surface", None)
    if s is None:
        return -1, -1, -1
    r = env.rotated_agent_rect
    return s._sprite_idx, r.center[0], r.center[1]


# ---- snapshots --------------------------------------------------------------------------------
def snap_mm(env, info):
    si, dx, dy = sprite_of(env)
    d = dict(ax=env.agent.rect.center[0], ay=env.agent.rect.center[1], arot=env.agent.rotation,
             disp_sprite=si, disp_x=dx, disp_y=dy, glyph=glyph_id(),
             cur_cmd=env._current_command, cmd_steps=env._command_steps, verify_step=env._command_verify_step,
             tiles_on=int(env.arena.tiles_on), tx=env._target_pos[0], ty=env._target_pos[1],
             vis_len=len(env._command_visualization or []), num_commands=env.num_commands,
             nx=env.normalized_agent_position[0], ny=env.normalized_agent_position[1],
             expl_dur=env._explosion_duration, expl_delay=env._explosion_delay,
             max_episode_steps=env.max_episode_steps)
    if hasattr(env, "_total_commands_completed"):
        d["total_completed"] = env._total_commands_completed
        d["t"] = env.t
        d["show_dur"] = env.show_duration
        d["show_delay"] = env.show_delay
    for k in ("reward", "length", "success", "commands_completed", "max_command_sequence"):
        d["info_" + k] = float(info[k]) if k in info else np.nan
    if "ground_truth" in info:
        d["gt0"], d["gt1"] = [float(v) for v in info["ground_truth"]]
    lists = dict(cmds=[CMD_IDS[c] for c in env._commands])
    if hasattr(env, "_commands_one_hot"):  # MortarMayhemB*: the vector observation
        lists["vec"] = [float(v) for v in env._commands_one_hot]
    return d, lists


def snap_mp(env, info):
    si, dx, dy = sprite_of(env)
    d = dict(ax=env.agent.rect.center[0], ay=env.agent.rect.center[1], arot=env.agent.rotation,
             disp_sprite=si, disp_x=dx, disp_y=dy,
             off=int(env.is_off_path), fails=env.num_fails, t=env.t,
             cross_alpha=env.fall_off_surface.get_alpha(), cross_x=env.fall_off_rect.center[0],
             cross_y=env.fall_off_rect.center[1],
             sx=env.start[0], sy=env.start[1], ex=env.end[0], ey=env.end[1],
             nx=env.normalized_agent_position[0], ny=env.normalized_agent_position[1])
    for k in ("reward", "length", "success", "num_fails"):
        d["info_" + k] = float(info[k]) if k in info else np.nan
    path = env.mystery_path.path
    lists = dict(path=[v for n in path for v in (n.x, n.y)],
                 visited=[int(n.reward_visited) for n in path],
                 walls=[v for n in env.mystery_path.wall_nodes for v in (n.x, n.y)])
    return d, lists


def snap_emp(env, info):
    si, dx, dy = sprite_of(env)
    d = dict(ax=env.agent.rect.center[0], ay=env.agent.rect.center[1], arot=env.agent.rotation,
             disp_sprite=si, rect_y=env.rotated_agent_rect.y, agent_draw_x=env.agent_draw_x,
             camera_x=env.camera_x, bg_scroll=env.bg_scroll,
             off=int(env.is_off_path), fails=env.num_fails, t=env.t, stamina=env.stamina,
             max_x=env.max_x_reached, tiles_visited=env.tiles_visited, cur_seg=env.current_segment,
             num_seg=env.endless_path.num_segments,
             cross_alpha=env.fall_off_surface.get_alpha(), cross_x=env.fall_off_rect.center[0],
             cross_y=env.fall_off_rect.center[1],
             cur_nx=env.current_node.x, cur_ny=env.current_node.y,
             nx=env.normalized_agent_position[0], ny=env.normalized_agent_position[1],
             n_falloff=len(env.fall_off_locations),
             gt0=int(env.target_direction[0]), gt1=int(env.target_direction[1]), gt2=int(env.target_direction[2]))
    for k in ("reward", "length", "num_fails", "max_x", "tiles_visited"):
        d["info_" + k] = float(info[k]) if k in info else np.nan
    nodes = [n for seg in env.endless_path.path for n in seg]
    lists = dict(path=[v for n in nodes for v in (n.x, n.y)],
                 seglen=[len(seg) for seg in env.endless_path.path],
                 rvis=[int(n.reward_visited) for n in nodes],
                 svis=[int(n.stamina_visited) for n in nodes])
    return d, lists


def snap_ss(env, info):
    si, dx, dy = sprite_of(env)
    endless = hasattr(env, "coin_t")
    d = dict(ax=env.agent.rect.center[0], ay=env.agent.rect.center[1], arot=env.agent.rotation,
             disp_sprite=si, disp_x=dx, disp_y=dy,
             health=env.current_agent_health, alpha=env.spotlight_surface.get_alpha(),
             spawn_timer=env.spawn_timer, n_spots=len(env.spotlights), t=env.t,
             la0=int(env.last_action[0]), la1=int(env.last_action[1]), last_reward=getattr(env, "last_reward", np.nan),
             bg_red=int(env.bg is env.red_background_surface), coins_collected=env.coins_collected)
    if endless:
        d.update(coin_t=env.coin_t, coin_x=env.coin.location[0], coin_y=env.coin.location[1])
        keys = ("reward", "length", "agent_health", "coins_collected")
    else:
        d.update(num_coins=env.num_coins, n_coins_left=len(env.coins), exit_x=env.exit.location[0],
                 exit_y=env.exit.location[1], exit_open=int(env.exit.open), n_intervals=len(env.spawn_intervals))
        keys = ("reward", "length", "agent_health", "coins_collected", "success")
    for k in keys:
        d["info_" + k] = float(info[k]) if k in info else np.nan
    if "ground_truth" in info:
        for i, v in enumerate(info["ground_truth"]):
            d["gt%d" % i] = float(v)
    spots = []
    for s in env.spotlights:
        spots += [s.radius, s.speed, s.t, float(s.done), s.spawn_location.x, s.spawn_location.y,
                  s.target_location.x, s.target_location.y, s.offset_location.x, s.offset_location.y,
                  s.current_location.x, s.current_location.y]
    lists = dict(spots=spots)
    if not endless:
        lists["coins"] = [v for c in env.coins for v in c.location]
    return d, lists


# ---- expert-ish policies (use env internals; only to reach deep states) --------------------------
def _toward(d):
    return 0 if d == 0 else (1 if d < 0 else 2)


def pol_mm_grid(env, prng, skill):
    if env._command_visualization or env.arena.tiles_on or prng.random() > skill:
        return int(prng.integers(0, 4)) if prng.random() > skill else 0
    gx, gy = env.agent.grid_position
    tx, ty = env._target_pos
    if (gx, gy) == (tx, ty):
        return 0
    if tx > gx:
        want = 270
    elif tx < gx:
        want = 90
    elif ty < gy:
        want = 0
    else:
        want = 180
    if env.agent.rotation == want:
        return 3
    return 1 if (want - env.agent.rotation) % 360 in (90, 180) else 2


def pol_mm_free(env, prng, skill, wrap=False):
    if prng.random() > skill:
        return prng.integers(0, 3, 2)
    if env._command_visualization or env.arena.tiles_on:
        return np.array([0, 0])
    td = env.arena.tile_dim
    cx = env.arena.rect[0] + env._target_pos[0] * td + td / 2
    cy = env.arena.rect[1] + env._target_pos[1] * td + td / 2
    dx, dy = cx - env.agent.rect.center[0], cy - env.agent.rect.center[1]
    if wrap:
        dx = (dx + 42) % 84 - 42
        dy = (dy + 42) % 84 - 42
    dx = 0 if abs(dx) < 3 else dx
    dy = 0 if abs(dy) < 3 else dy
    return np.array([_toward(dx), _toward(dy)])


def pol_mp(env, prng, skill):
    if prng.random() > skill:
        return prng.integers(0, 3, 2)
    path = env.mystery_path.path  # end-first
    pos = env.normalized_agent_position
    idx = None
    for i, n in enumerate(path):
        if (n.x, n.y) == pos:
            idx = i
            break
    if idx is None or idx == 0:
        return np.array([0, 0])
    nxt = path[idx - 1]
    dx = nxt.x * 12 + 6 - env.agent.rect.center[0]
    dy = nxt.y * 12 + 6 - env.agent.rect.center[1]
    # stay centred on the off-axis so the 12-px body does not clip a neighbour tile
    return np.array([_toward(dx), _toward(dy)])


def pol_mp_grid(env, prng, skill):
    if prng.random() > skill:
        return int(prng.integers(0, 4))
    path = env.mystery_path.path  # end-first
    pos = env.normalized_agent_position
    idx = None
    for i, n in enumerate(path):
        if (n.x, n.y) == pos:
            idx = i
            break
    if idx is None or idx == 0:
        return 0
    nxt = path[idx - 1]
    dx, dy = nxt.x - pos[0], nxt.y - pos[1]
    want = 270 if dx > 0 else (90 if dx < 0 else (0 if dy < 0 else 180))
    if env.agent.rotation == want:
        return 3
    return 1 if (want - env.agent.rotation) % 360 in (90, 180) else 2


def pol_emp(env, prng, skill):
    if prng.random() > skill:
        return int(prng.integers(0, 4))
    n = env.current_node.next_node
    if n is None:
        return 0
    dx = n.x * 12 + 6 - env.agent.rect.center[0]
    dy = n.y * 12 + 6 - env.agent.rect.center[1]
    if dy < 0:
        return 2
    if dy > 0:
        return 3
    if dx > 0:
        return 1
    return 0


def pol_ss(env, prng, skill):
    if prng.random() > skill:
        return prng.integers(0, 3, 2)
    if hasattr(env, "coin"):
        tgt = env.coin.location
    elif env.coins:
        tgt = env.coins[0].location
    else:
        tgt = env.exit.location
    dx, dy = tgt[0] - env.agent.rect.center[0], tgt[1] - env.agent.rect.center[1]
    dx = 0 if abs(dx) < 3 else dx
    dy = 0 if abs(dy) < 3 else dy
    return np.array([_toward(dx), _toward(dy)])


ENVS = {
    "MortarMayhem-Grid-v0": dict(snap=snap_mm, pol=pol_mm_grid, disc=True),
    "MortarMayhem-v0": dict(snap=snap_mm, pol=pol_mm_free, disc=False),
    "Endless-MortarMayhem-v0": dict(snap=snap_mm, pol=lambda e, p, s: pol_mm_free(e, p, s, True), disc=False),
    "MortarMayhemB-Grid-v0": dict(snap=snap_mm, pol=pol_mm_grid, disc=True),
    "MortarMayhemB-v0": dict(snap=snap_mm, pol=pol_mm_free, disc=False),
    "MysteryPath-v0": dict(snap=snap_mp, pol=pol_mp, disc=False),
    "MysteryPath-Grid-v0": dict(snap=snap_mp, pol=pol_mp_grid, disc=True),
    "Endless-MysteryPath-v0": dict(snap=snap_emp, pol=pol_emp, disc=True),
    "SearingSpotlights-v0": dict(snap=snap_ss, pol=pol_ss, disc=False),
    "Endless-SearingSpotlights-v0": dict(snap=snap_ss, pol=pol_ss, disc=False),
}

# sessions: (seed, options, skill, n_rows_of_steps).  skill=0 -> uniform random actions.
SESSIONS = {
    "MortarMayhem-Grid-v0": [
        (0, None, 0.0, 200), (1, None, 0.0, 200), (2, None, 1.0, 400), (3, None, 0.9, 400), (4, None, 0.97, 600),
        (5, dict(arena_size=6, allowed_commands=9, command_count=[3, 5, 10], command_show_duration=[1, 2, 3],
                 command_show_delay=[0, 1, 2], explosion_duration=[2, 3], explosion_delay=[4, 6, 8],
                 reward_command_failure=-0.1, reward_episode_success=1.0), 0.97, 800),
        (6, dict(arena_size=2, allowed_commands=4, command_count=[4], visual_feedback=False), 0.95, 300),
        (7, dict(arena_size=3, allowed_commands=5, command_count=[2, 8]), 0.9, 300),
        (123456789012, None, 0.95, 300),
    ],
    "MortarMayhem-v0": [
        (0, None, 0.0, 200), (1, None, 1.0, 600), (2, None, 0.97, 600),
        (3, dict(arena_size=6, allowed_commands=5, command_count=[3, 6], explosion_duration=[4, 6],
                 explosion_delay=[12, 18], reward_command_failure=-0.5, reward_episode_success=2.0), 0.98, 800),
        (4, dict(arena_size=3, command_count=[5], command_show_duration=[2], command_show_delay=[0]), 0.95, 400),
    ],
    "Endless-MortarMayhem-v0": [
        (0, None, 0.0, 150), (1, None, 1.0, 1500), (2, None, 0.985, 1500),
        (3, dict(max_steps=200, initial_command_count=3, allowed_commands=5, command_show_duration=[2, 3],
                 command_show_delay=[0, 1], explosion_duration=[4, 6], explosion_delay=[12, 18],
                 reward_new_command_success=0.5, reward_command_failure=-0.25), 1.0, 900),
        (4, dict(initial_command_count=2, visual_feedback=False), 0.99, 600),
    ],
    "MortarMayhemB-Grid-v0": [
        (0, None, 0.0, 200), (1, None, 1.0, 400), (2, None, 0.95, 500),
        (3, dict(arena_size=6, allowed_commands=9, command_count=[3, 5, 20], explosion_duration=[2, 3], explosion_delay=[4, 6, 8],
                 reward_command_failure=-0.1, reward_episode_success=1.0), 0.97, 800),
        (4, dict(arena_size=3, allowed_commands=4, command_count=[4], visual_feedback=False), 0.95, 300),
    ],
    "MortarMayhemB-v0": [
        (0, None, 0.0, 200), (1, None, 1.0, 600), (2, None, 0.97, 600),
        (3, dict(arena_size=6, allowed_commands=5, command_count=[3, 6, 20], explosion_duration=[4, 6],
                 explosion_delay=[12, 18], reward_command_failure=-0.5, reward_episode_success=2.0), 0.98, 800),
        (4, dict(arena_size=3, command_count=[5], agent_speed=2.0, visual_feedback=False), 0.95, 400),
    ],
    "MysteryPath-v0": [
        (0, None, 0.0, 600), (1, None, 1.0, 300), (2, None, 0.93, 900), (3, None, 0.8, 700),
        (4, dict(max_steps=64, cardinal_origin_choice=[2], show_origin=True, show_goal=True,
                 reward_fall_off=-0.1, reward_step=-0.01, reward_goal=2.0), 0.9, 300),
        (5, dict(cardinal_origin_choice=[1, 3], visual_feedback=False), 0.95, 400),
    ],
    "MysteryPath-Grid-v0": [
        (0, None, 0.0, 300), (1, None, 1.0, 300), (2, None, 0.9, 600),
        (3, dict(max_steps=40, cardinal_origin_choice=[0, 3], show_origin=True, show_goal=True, reward_fall_off=-0.1,
                 reward_step=-0.01, reward_goal=2.0, reward_path_progress=0.1), 0.92, 400),
        (4, dict(visual_feedback=False), 0.8, 300),
    ],
    "Endless-MysteryPath-v0": [
        (0, None, 0.0, 200), (1, None, 1.0, 800), (2, None, 0.97, 1200), (3, None, 0.9, 600),
        (4, dict(max_steps=300, stamina_level=12, reward_fall_off=-0.1, reward_path_progress_dense=0.05,
                 reward_step=-0.001, camera_offset_scale=3.0), 0.97, 700),
    ],
    "SearingSpotlights-v0": [
        (0, None, 0.0, 300), (1, None, 1.0, 600), (2, None, 0.9, 600),
        (3, dict(num_coins=[1, 2, 3], agent_health=20, initial_spawns=2, max_steps=128, reward_death=-1.0,
                 reward_inside_spotlight=-0.01, reward_outside_spotlight=0.001), 1.0, 800),
        # NOTE: use_exit=False crashes the reference itself (searing_spotlights.py:432 reads self.exit) -> not a fixture
        (4, dict(num_coins=[2], agent_health=50, light_dim_off_duration=3, exit_scale=0.5), 1.0, 500),
        (5, dict(sample_agent_position=False, agent_health=100), 0.95, 600),
    ],
    "Endless-SearingSpotlights-v0": [
        (0, None, 0.0, 300), (1, None, 1.0, 800), (2, None, 0.9, 600),
        (3, dict(agent_health=40, steps_per_coin=60, initial_spawns=5, spawn_interval=20, max_steps=400,
                 reward_death=-1.0, reward_inside_spotlight=-0.01, reward_outside_spotlight=0.001), 1.0, 1200),
        (4, dict(agent_health=1000, spot_min_speed=0.01, spot_max_speed=0.05, spawn_interval=10), 1.0, 900),
        (5, dict(sample_agent_position=False, light_dim_off_duration=0, visual_feedback=False), 0.9, 400),
    ],
}


# `--long`: "sample one per episode" option lists LONGER than eight entries (the reference samples from any length:
# mortar_mayhem_grid.py:181,253-254,268-269, mystery_path.py:154, searing_spotlights.py:408) -> tests/golden/long_<env>.npz.
# 16-entry command_count lists, 40-entry duration lists (longer than the 32 entries the HIP path keeps in kernel arguments),
# many short episodes so that many list positions are drawn.
_CC16 = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 2, 3, 1, 4, 12, 5]
_D40 = [2 + (7 * k) % 5 for k in range(40)]
_L40 = [4 + (11 * k) % 9 for k in range(40)]
LONG_SESSIONS = {
    "MortarMayhem-Grid-v0": [
        (11, dict(command_count=_CC16, command_show_duration=[1, 2, 3, 1, 2, 3, 1, 2, 3, 2], command_show_delay=[0, 1, 2, 0, 1, 2, 0, 1, 2],
                  explosion_duration=_D40, explosion_delay=_L40), 0.9, 900),
        (12, dict(arena_size=6, allowed_commands=9, command_count=list(range(1, 33)), explosion_duration=[2] * 33 + [3],
                  explosion_delay=[4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]), 0.8, 900),
        # entries beyond a byte (the reference takes any int; the HIP path kept these draws in bytes until round 5)
        (13, dict(command_count=[1, 2], command_show_duration=[300, 2], command_show_delay=[260, 0], explosion_duration=[270, 2],
                  explosion_delay=[400, 3]), 0.95, 3000),
    ],
    "MortarMayhem-v0": [
        (11, dict(command_count=_CC16, command_show_duration=[1, 2, 3, 4, 1, 2, 3, 4, 1, 2, 3], explosion_duration=[d + 2 for d in _D40],
                  explosion_delay=[d + 8 for d in _L40]), 0.97, 1200),
    ],
    "Endless-MortarMayhem-v0": [
        (11, dict(command_show_duration=[1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 1], command_show_delay=[0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 2, 2],
                  explosion_duration=[d + 2 for d in _D40], explosion_delay=[d + 8 for d in _L40]), 0.6, 1500),
        (13, dict(initial_command_count=1, command_show_duration=[1, 280], explosion_duration=[2, 260], explosion_delay=[300, 4],
                  max_steps=900), 0.97, 3000),
    ],
    "MortarMayhemB-Grid-v0": [
        (11, dict(command_count=_CC16 + [20, 18], explosion_duration=_D40, explosion_delay=_L40), 0.9, 900),
    ],
    "MortarMayhemB-v0": [
        (11, dict(command_count=_CC16 + [20, 18], explosion_duration=[d + 2 for d in _D40], explosion_delay=[d + 8 for d in _L40]), 0.95, 1200),
        (13, dict(command_count=[2, 3], explosion_duration=[3, 300], explosion_delay=[320, 8]), 0.97, 2500),
    ],
    "MysteryPath-v0": [
        (11, dict(max_steps=24, cardinal_origin_choice=[0, 1, 2, 3, 3, 2, 1, 0, 2, 2, 1, 3]), 0.9, 700),
        (12, dict(max_steps=16, cardinal_origin_choice=[(5 * k) % 4 for k in range(37)]), 0.5, 700),
    ],
    "MysteryPath-Grid-v0": [
        (11, dict(max_steps=20, cardinal_origin_choice=[0, 1, 2, 3, 3, 2, 1, 0, 2, 2, 1, 3]), 0.9, 700),
        (12, dict(max_steps=12, cardinal_origin_choice=[(3 * k) % 4 for k in range(41)]), 0.5, 700),
    ],
    "SearingSpotlights-v0": [
        (11, dict(num_coins=[1, 2, 3, 4, 1, 2, 3, 4, 2, 1, 3], max_steps=24, agent_health=50), 0.9, 700),
        (12, dict(num_coins=[1 + (3 * k) % 4 for k in range(35)], max_steps=16, agent_health=50), 0.5, 600),
    ],
}


def main_long():
    for env_id, sess in LONG_SESSIONS.items():
        rows_all, meta = [], []
        for (seed, options, skill, n) in sess:
            rows = run_session(env_id, seed, options, skill, n)
            rows_all.append(rows)
            n_eps = sum(r["done"] for r in rows)
            meta.append(dict(seed=seed, options=options, skill=skill, n_steps=n, episodes=n_eps))
            print(env_id, "long seed", seed, "rows", len(rows), "episodes", n_eps)
        out = pack(rows_all, meta)
        fn = os.path.join(HERE, "long_" + env_id.replace("-", "_") + ".npz")
        np.savez_compressed(fn, **out)
        print("  ->", fn, os.path.getsize(fn) // 1024, "KiB")


def run_session(env_id, seed, options, skill, n_steps):
    spec = ENVS[env_id]
    env = make(env_id)
    prng = np.random.Generator(np.random.PCG64(987654321 + (seed % 1000)))
    rows = []

    def record(kind, seed_v, action, reward, done, info):
        d, lists = spec["snap"](env, info)
        rows.append(dict(kind=kind, seed=seed_v, action=action, reward=float(reward), done=int(done), snap=d,
                         lists=lists, rng=rng_words(env)))

    LAST_GLYPH[0] = None
    _, info = env.reset(seed=seed, options=options)
    record(0, seed, (0, 0), 0.0, 0, info)
    for _ in range(n_steps):
        a = spec["pol"](env, prng, skill)
        LAST_GLYPH[0] = None
        if spec["disc"]:
            a = int(a)
            _, r, done, _, info = env.step(a)
            act = (a, 0)
        else:
            a = np.asarray(a)
            _, r, done, _, info = env.step(a)
            act = (int(a[0]), int(a[1]))
        record(1, -1, act, r, done, info)
        if done:
            LAST_GLYPH[0] = None
            _, info = env.reset(options=options)  # seed=None: continue the stream (auto-reset semantics)
            record(0, -1, (0, 0), 0.0, 0, info)
    return rows


def pack(sessions_rows, sessions_meta):
    fields = []
    for rows in sessions_rows:
        for r in rows:
            for k in r["snap"]:
                if k not in fields:
                    fields.append(k)
    list_names = []
    for rows in sessions_rows:
        for r in rows:
            for k in r["lists"]:
                if k not in list_names:
                    list_names.append(k)
    out = dict(fields=np.array(fields), meta=np.array(json.dumps(sessions_meta)))
    for si, rows in enumerate(sessions_rows):
        n = len(rows)
        p = "s%d_" % si
        out[p + "kind"] = np.array([r["kind"] for r in rows], dtype=np.int8)
        out[p + "seed"] = np.array([r["seed"] for r in rows], dtype=np.int64)
        out[p + "action"] = np.array([r["action"] for r in rows], dtype=np.int8)
        out[p + "reward"] = np.array([r["reward"] for r in rows], dtype=np.float64)
        out[p + "done"] = np.array([r["done"] for r in rows], dtype=np.int8)
        out[p + "rng"] = np.stack([r["rng"] for r in rows])
        snap = np.full((n, len(fields)), np.nan)
        for i, r in enumerate(rows):
            for k, v in r["snap"].items():
                snap[i, fields.index(k)] = float(v)
        out[p + "snap"] = snap
        for ln in list_names:
            m = max(len(r["lists"].get(ln, [])) for r in rows)
            arr = np.full((n, max(m, 1)), np.nan)
            for i, r in enumerate(rows):
                v = r["lists"].get(ln, [])
                arr[i, :len(v)] = v
            out[p + "L_" + ln] = arr
    return out


# options that fix geometry a handle's instances share (tests/test_gpu_option_sets.py pairs sessions by them)
GEOMETRY_KEYS = ("arena_size", "agent_scale", "agent_speed", "coin_scale", "show_last_action", "initial_spawn_interval",
                 "spawn_interval_threshold", "exit_scale", "camera_offset_scale")


def main_fuzz():
    """`--fuzz`: sessions with seeded random option dictionaries (tests/option_fuzz.py) -> tests/golden/fuzz_<env>.npz."""
    sys.path.insert(0, os.path.dirname(HERE))
    from option_fuzz import CASES
    # optional: --seed S --trials T --out DIR (a wider one-off hunt into a scratch directory; the committed fixtures use the defaults)
    arg = {sys.argv[k]: sys.argv[k + 1] for k in range(len(sys.argv) - 1) if sys.argv[k].startswith("--")}
    seed0, trials, out_dir = int(arg.get("--seed", 1000)), int(arg.get("--trials", 6)), arg.get("--out", HERE)
    # --default-geometry: the same dictionaries without their *_scale keys -> fuzzd_<env>.npz (sessions that can share one handle:
    # tests/test_gpu_option_sets.py replays them two per handle, and geometry is per handle)
    default_geometry = "--default-geometry" in sys.argv
    os.makedirs(out_dir, exist_ok=True)
    for env_id, gen in CASES:
        rng = np.random.Generator(np.random.PCG64(seed0 + sum(map(ord, env_id))))
        rows_all, meta = [], []
        # (--default-geometry: one more session whose geometry options are those of trial 0, so that EVERY id has at least one pair
        # of sessions that can share a handle -- MortarMayhem-v0's arena_size x agent_speed and Endless-MysteryPath-v0's
        # camera_offset_scale x agent_speed left their six trials without one, and the two-per-handle replay skipped them: VERDICT r5)
        for trial in range(trials + (1 if default_geometry else 0)):
            options = gen(rng, env_id)
            if default_geometry:
                options = {k: v for k, v in options.items() if k not in ("agent_scale", "coin_scale", "exit_scale")}
                if trial == trials:
                    options.update({k: v for k, v in meta[0]["options"].items() if k in GEOMETRY_KEYS and k in options})
            try:
                rows = run_session(env_id, 100 + trial, options, 0.9, 160)
            except Exception as e:  # an option set the reference itself cannot run is not a fixture
                print(env_id, "trial", trial, "skipped:", type(e).__name__, e)
                continue
            rows_all.append(rows)
            n_eps = sum(r["done"] for r in rows)
            meta.append(dict(seed=100 + trial, options=options, skill=0.9, n_steps=160, episodes=n_eps))
            print(env_id, "fuzz trial", trial, "rows", len(rows), "episodes", n_eps)
        out = pack(rows_all, meta)
        fn = os.path.join(out_dir, ("fuzzd_" if default_geometry else "fuzz_") + env_id.replace("-", "_") + ".npz")
        np.savez_compressed(fn, **out)
        print("  ->", fn, os.path.getsize(fn) // 1024, "KiB")


def main():
    if "--fuzz" in sys.argv:
        return main_fuzz()
    if "--long" in sys.argv:
        return main_long()
    only = sys.argv[1:] or list(SESSIONS)
    for env_id in only:
        rows_all, meta = [], []
        for (seed, options, skill, n) in SESSIONS[env_id]:
            rows = run_session(env_id, seed, options, skill, n)
            rows_all.append(rows)
            n_eps = sum(r["done"] for r in rows)
            meta.append(dict(seed=seed, options=options, skill=skill, n_steps=n, episodes=n_eps))
            print(env_id, "seed", seed, "skill", skill, "rows", len(rows), "episodes", n_eps)
        out = pack(rows_all, meta)
        fn = os.path.join(HERE, "logic_" + env_id.replace("-", "_") + ".npz")
        np.savez_compressed(fn, **out)
        print("  ->", fn, os.path.getsize(fn) // 1024, "KiB")


if __name__ == "__main__":
    main()
