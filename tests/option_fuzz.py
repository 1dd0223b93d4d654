"""Seeded generators of reset-option dictionaries; shared by tests/test_gpu_option_fuzz.py (HIP vs oracle) and
tests/golden/make_golden.py (oracle vs the reference: "fuzz" sessions).  Every key of the reference's reset-options dictionaries
is drawn, the geometry options (`agent_scale`, `coin_scale`, `exit_scale`: 0.5x .. 2x their defaults) included; what stays out
are values the REFERENCE cannot run and the capacity limits include/memgym.h names (16 live spotlights per instance)."""
import numpy as np  # noqa: F401


def _lst(rng, lo, hi, kmax=3):
    k = int(rng.integers(1, kmax + 1))
    return sorted({int(v) for v in rng.integers(lo, hi + 1, k)})


def _scale(rng, default):
    """0.5x .. 2x the default of a *_scale option; crosses the sprite sizes the composers keep in registers (csrc/mg_raster.hpp
    StampRegs) and the width regimes of pygame.draw.circle (filled / 1 px / thick ring)"""
    return float(default * rng.choice([0.5, 0.7, 0.8, 1.0, 1.0, 1.2, 1.36, 1.6, 1.8, 2.0]))


def _rew(rng):
    return float(rng.choice([0.0, 0.1, -0.1, 0.25, 1.0, -0.01, 2.0]))


def mortar_opts(rng, env_id):
    b = env_id.startswith("MortarMayhemB")
    endless = env_id.startswith("Endless")
    grid = "Grid" in env_id
    o = dict(allowed_commands=int(rng.integers(4, 10)), explosion_duration=_lst(rng, 1, 6), explosion_delay=_lst(rng, 2, 18),
             visual_feedback=bool(rng.integers(0, 2)), reward_command_failure=_rew(rng), reward_command_success=_rew(rng))
    if not b:
        o.update(command_show_duration=_lst(rng, 1, 4), command_show_delay=_lst(rng, 0, 3))
    if endless:
        o.update(max_steps=int(rng.choice([-1, 60, 150])), initial_command_count=int(rng.integers(1, 5)), reward_new_command_success=_rew(rng))
    else:
        o.update(arena_size=int(rng.integers(2, 7)), command_count=_lst(rng, 1, 12), reward_episode_success=_rew(rng))
    if not grid:
        o.update(agent_speed=float(rng.choice([2.0, 3.0, 4.0])))
    # drawn last, so that the other values of a trial are the ones earlier fixtures had (the grid ids accept the key and ignore it)
    o.update(agent_scale=_scale(rng, 0.25))
    return o


def mystery_opts(rng, env_id):
    if env_id == "Endless-MysteryPath-v0":
        return dict(max_steps=int(rng.choice([-1, 80, 200])), stamina_level=int(rng.integers(6, 30)), show_stamina=bool(rng.integers(0, 2)),
                    show_past_path=bool(rng.integers(0, 2)), visual_feedback=bool(rng.integers(0, 2)), reward_fall_off=_rew(rng),
                    reward_path_progress=_rew(rng), reward_path_progress_dense=_rew(rng), reward_step=_rew(rng),
                    camera_offset_scale=float(rng.choice([3.0, 5.0, 7.0])), show_background=bool(rng.integers(0, 2)),
                    agent_speed=float(rng.choice([2.0, 3.0, 4.0])), agent_scale=_scale(rng, 0.25))
    o = dict(max_steps=int(rng.integers(20, 200)), cardinal_origin_choice=_lst(rng, 0, 3, 4), show_origin=bool(rng.integers(0, 2)),
             show_goal=bool(rng.integers(0, 2)), visual_feedback=bool(rng.integers(0, 2)), reward_goal=_rew(rng), reward_fall_off=_rew(rng),
             reward_path_progress=_rew(rng), reward_step=_rew(rng))
    o.update(agent_scale=_scale(rng, 0.25))
    return o


def spot_opts(rng, env_id):
    o = dict(initial_spawns=int(rng.integers(1, 6)), spot_min_radius=float(rng.choice([7.5, 8.0, 9.0])), spot_max_radius=float(rng.choice([11.0, 13.75])),
             spot_min_speed=float(rng.choice([0.0025, 0.01])), spot_max_speed=float(rng.choice([0.02, 0.0075 * 4])), spot_damage=float(rng.choice([0.5, 1.0, 2.0])),
             visual_feedback=bool(rng.integers(0, 2)), light_dim_off_duration=int(rng.integers(0, 10)), light_threshold=int(rng.choice([255, 200, 128, 30])),
             coins_visible=bool(rng.integers(0, 2)), agent_health=int(rng.integers(3, 40)), sample_agent_position=bool(rng.integers(0, 2)),
             show_last_positive_reward=bool(rng.integers(0, 2)), agent_visible=bool(rng.integers(0, 2)), reward_inside_spotlight=_rew(rng),
             reward_outside_spotlight=_rew(rng), reward_death=_rew(rng), reward_coin=_rew(rng))
    if env_id.startswith("Endless"):
        # the HIP path holds at most 16 live spotlights per instance (error bit 1 otherwise, include/memgym.h): keep
        # initial_spawns + lifetime / spawn_interval below that (lifetime <= 1 / spot_min_speed steps)
        life = int(np.ceil(1.0 / o["spot_min_speed"]))
        floor_interval = int(np.ceil(life / (14 - o["initial_spawns"])))
        o.update(max_steps=int(rng.choice([-1, 100, 300])), steps_per_coin=int(rng.integers(30, 200)),
                 spawn_interval=max(int(rng.integers(5, 60)), floor_interval), coin_show_duration=int(rng.integers(1, 12)))
    else:
        # show_last_action = False crashes the ENDLESS reference (endless_searing_spotlights.py:422 uses action_colors,
        # which only exists when the flag is set, :343); the finite env guards the use (:465)
        o.update(max_steps=int(rng.integers(60, 300)), num_spawns=int(rng.integers(0, 12)), num_coins=_lst(rng, 1, 4), reward_exit=_rew(rng),
                 show_last_action=bool(rng.integers(0, 2)), exit_visible=bool(rng.integers(0, 2)))
    # drawn last, so that the other values of a trial are the ones earlier fixtures had
    o.update(black_background=bool(rng.integers(0, 4) == 0), hide_chessboard=bool(rng.integers(0, 4) == 0))
    o.update(agent_scale=_scale(rng, 0.25), coin_scale=_scale(rng, 0.375))
    if not env_id.startswith("Endless"):
        o.update(exit_scale=_scale(rng, 0.5))
    return o


CASES = [("MortarMayhem-Grid-v0", mortar_opts), ("MortarMayhem-v0", mortar_opts), ("Endless-MortarMayhem-v0", mortar_opts),
         ("MortarMayhemB-Grid-v0", mortar_opts), ("MortarMayhemB-v0", mortar_opts), ("MysteryPath-v0", mystery_opts),
         ("MysteryPath-Grid-v0", mystery_opts), ("Endless-MysteryPath-v0", mystery_opts), ("SearingSpotlights-v0", spot_opts),
         ("Endless-SearingSpotlights-v0", spot_opts)]
