"""Reset-option dictionaries of the reference, key for key (names, defaults and validation behaviour):

  MortarMayhem-Grid-v0          memory_gym/mortar_mayhem_grid.py:21-53
  MortarMayhem-v0               memory_gym/mortar_mayhem.py:21-54
  Endless-MortarMayhem-v0       memory_gym/endless_mortar_mayhem.py:21-53
  MortarMayhemB-Grid-v0         memory_gym/mortar_mayhem_b_grid.py:21-53
  MortarMayhemB-v0              memory_gym/mortar_mayhem_b.py:21-54
  MysteryPath-v0                memory_gym/mystery_path.py:21-50
  MysteryPath-Grid-v0           memory_gym/mystery_path_grid.py:21-49
  Endless-MysteryPath-v0        memory_gym/endless_mystery_path.py:22-54
  SearingSpotlights-v0          memory_gym/searing_spotlights.py:22-81
  Endless-SearingSpotlights-v0  memory_gym/endless_searing_spotlights.py:21-73

`process_reset_params(env_id, options)` behaves like the per-class static method of the same name: unknown keys
raise AssertionError with the reference's message, missing keys are filled with the defaults.
"""
SCALE = 0.25

DEFAULTS = {
    "MortarMayhem-Grid-v0": {
        "agent_scale": 1.0 * SCALE, "arena_size": 5, "allowed_commands": 5, "command_count": [10],
        "command_show_duration": [3], "command_show_delay": [1], "explosion_duration": [2], "explosion_delay": [6],
        "visual_feedback": True, "reward_command_failure": 0.0, "reward_command_success": 0.1,
        "reward_episode_success": 0.0,
    },
    "MortarMayhem-v0": {
        "agent_scale": 1.0 * SCALE, "agent_speed": 12.0 * SCALE, "arena_size": 5, "allowed_commands": 9,
        "command_count": [10], "command_show_duration": [3], "command_show_delay": [1], "explosion_duration": [6],
        "explosion_delay": [18], "visual_feedback": True, "reward_command_failure": 0.0,
        "reward_command_success": 0.1, "reward_episode_success": 0.0,
    },
    "Endless-MortarMayhem-v0": {
        "max_steps": -1, "agent_scale": 1.0 * SCALE, "agent_speed": 12.0 * SCALE, "allowed_commands": 9,
        "initial_command_count": 1, "command_show_duration": [3], "command_show_delay": [1],
        "explosion_duration": [6], "explosion_delay": [18], "visual_feedback": True, "reward_command_failure": 0.0,
        "reward_command_success": 0.1, "reward_new_command_success": 0.0,
    },
    "MortarMayhemB-Grid-v0": {
        "agent_scale": 1.0 * SCALE, "arena_size": 5, "allowed_commands": 5, "command_count": [10],
        "explosion_duration": [2], "explosion_delay": [6], "visual_feedback": True, "reward_command_failure": 0.0,
        "reward_command_success": 0.1, "reward_episode_success": 0.0,
    },
    "MortarMayhemB-v0": {
        "agent_scale": 1.0 * SCALE, "agent_speed": 12.0 * SCALE, "arena_size": 5, "allowed_commands": 9,
        "command_count": [10], "explosion_duration": [6], "explosion_delay": [18], "visual_feedback": True,
        "reward_command_failure": 0.0, "reward_command_success": 0.1, "reward_episode_success": 0.0,
    },
    "MysteryPath-v0": {
        "max_steps": 512, "agent_scale": 1.0 * SCALE, "agent_speed": 12.0 * SCALE,
        "cardinal_origin_choice": [0, 1, 2, 3], "show_origin": False, "show_goal": False, "visual_feedback": True,
        "reward_goal": 1.0, "reward_fall_off": 0.0, "reward_path_progress": 0.1, "reward_step": 0.0,
    },
    "MysteryPath-Grid-v0": {
        "max_steps": 128, "agent_scale": 1.0 * SCALE, "cardinal_origin_choice": [0, 1, 2, 3], "show_origin": False,
        "show_goal": False, "visual_feedback": True, "reward_goal": 1.0, "reward_fall_off": 0.0,
        "reward_path_progress": 0.0, "reward_step": 0.0,
    },
    "Endless-MysteryPath-v0": {
        "max_steps": -1, "agent_scale": 1.0 * SCALE, "agent_speed": 12.0 * SCALE, "show_origin": False,
        "show_past_path": True, "show_background": False, "show_stamina": False, "visual_feedback": True,
        "camera_offset_scale": 5.0, "stamina_level": 20, "reward_fall_off": 0.0, "reward_path_progress": 0.1,
        "reward_path_progress_dense": 0.0, "reward_step": 0.0,
    },
    "SearingSpotlights-v0": {
        "max_steps": 256, "initial_spawns": 4, "num_spawns": 30, "initial_spawn_interval": 30,
        "spawn_interval_threshold": 10, "spawn_interval_decay": 0.95, "spot_min_radius": 30.0 * SCALE,
        "spot_max_radius": 55.0 * SCALE, "spot_min_speed": 0.0025, "spot_max_speed": 0.0075, "spot_damage": 1.0,
        "visual_feedback": True, "black_background": False, "hide_chessboard": False, "light_dim_off_duration": 6,
        "light_threshold": 255, "num_coins": [1], "coin_scale": 1.5 * SCALE, "coins_visible": False,
        "use_exit": True, "exit_scale": 2.0 * SCALE, "exit_visible": False, "agent_speed": 12.0 * SCALE,
        "agent_health": 5, "agent_scale": 1.0 * SCALE, "agent_visible": False, "sample_agent_position": True,
        "show_last_action": True, "show_last_positive_reward": True, "reward_inside_spotlight": 0.0,
        "reward_outside_spotlight": 0.0, "reward_death": 0.0, "reward_exit": 1.0, "reward_max_steps": 0.0,
        "reward_coin": 0.25,
    },
    "Endless-SearingSpotlights-v0": {
        "max_steps": -1, "steps_per_coin": 160, "initial_spawns": 3, "spawn_interval": 50,
        "spot_min_radius": 30.0 * SCALE, "spot_max_radius": 55.0 * SCALE, "spot_min_speed": 0.0025,
        "spot_max_speed": 0.0075, "spot_damage": 1.0, "visual_feedback": True, "black_background": False,
        "hide_chessboard": False, "light_dim_off_duration": 6, "light_threshold": 255, "coin_enabled": True,
        "coin_scale": 1.5 * SCALE, "coin_show_duration": 6, "coins_visible": False, "agent_speed": 12.0 * SCALE,
        "agent_health": 10, "agent_scale": 1.0 * SCALE, "agent_visible": False, "sample_agent_position": True,
        "show_last_action": True, "show_last_positive_reward": True, "reward_inside_spotlight": 0.0,
        "reward_outside_spotlight": 0.0, "reward_death": 0.0, "reward_coin": 0.25,
    },
}


def process_reset_params(env_id, reset_params):
    cloned = dict(DEFAULTS[env_id])
    if reset_params is not None:
        for k, v in reset_params.items():
            assert k in cloned.keys(), "Provided reset parameter (" + str(k) + ") is not valid. Check spelling."
            cloned[k] = v
    if "allowed_commands" in cloned:
        assert cloned["allowed_commands"] >= 4 and cloned["allowed_commands"] <= 9
    if "arena_size" in cloned:
        assert cloned["arena_size"] >= 2 and cloned["arena_size"] <= 6
    if env_id.startswith("MortarMayhemB"):  # mortar_mayhem_b_grid.py:52
        assert max(cloned["command_count"]) <= 20, "20 commands are allowed at maximum"
    return cloned


def calc_max_episode_steps(command_count, show_duration, show_delay, execution_duration, execution_delay):
    """memory_gym/pygame_assets.py:420-436"""
    clue_task_steps = (show_duration + show_delay) * command_count
    act_task_steps = (execution_duration + execution_delay) * command_count
    act_task_steps = act_task_steps - execution_delay + 1
    return clue_task_steps + act_task_steps
