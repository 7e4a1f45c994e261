"""ctypes loader for liboakgpu.so (the HIP product library).  Fails loudly: there is no CPU
fallback anywhere in oak_amd -- if the library is missing or no HIP device is usable, calls raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OAKGPU_LIB") or os.path.join(_HERE, "liboakgpu.so")   # (OAKGPU_LIB: an alternative build of the SAME library, e.g. tools/engine_variants.sh)

# every symbol include/oakgpu.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "oakgpu_create", "oakgpu_destroy", "oakgpu_last_error", "oakgpu_set_stream", "oakgpu_get_stream", "oakgpu_synchronize", "oakgpu_set_kernel_timing", "oakgpu_get_leaf_kernel_ms", "oakgpu_set_playouts_per_lane", "oakgpu_set_regroup", "oakgpu_set_tail_pack", "oakgpu_set_queue_order", "oakgpu_set_spread", "oakgpu_set_migration", "oakgpu_set_migration_window", "oakgpu_set_standstill_skip", "oakgpu_get_queue_counters", "oakgpu_set_rollout_engine",
    "oakgpu_device_count", "oakgpu_rollout_dev", "oakgpu_rollout", "oakgpu_rollout_group_dev", "oakgpu_rollout_group",
    "oakgpu_mt19937_fill", "oakgpu_rollout_draws_dev", "oakgpu_rollout_shared_device", "oakgpu_update_dev", "oakgpu_update",
    "oakgpu_choices_dev", "oakgpu_choices", "oakgpu_init_battles_dev", "oakgpu_init_battles",
    "oakgpu_set_ou_pools", "oakgpu_random_ou_battles_dev",
    "oakgpu_net_load", "oakgpu_net_load_memory", "oakgpu_net_free", "oakgpu_net_shape", "oakgpu_net_set_main_precision", "oakgpu_net_main_precision",
    "oakgpu_leaf_eval_dev", "oakgpu_leaf_eval", "oakgpu_leaf_eval_cached_dev", "oakgpu_leaf_cache_last_count", "oakgpu_leaf_eval_policy_dev", "oakgpu_leaf_eval_policy",
    "oakgpu_heap_create", "oakgpu_heap_destroy", "oakgpu_heap_empty", "oakgpu_heap_clear", "oakgpu_heap_kind", "oakgpu_heap_nodes", "oakgpu_heap_update",
    "oakgpu_heap_root_stats", "oakgpu_heap_child_stats", "oakgpu_search_heap", "oakgpu_search_agent_heap", "oakgpu_heap_check_shards", "oakgpu_heap_selftest",
    "oakgpu_tree_step_dev", "oakgpu_search", "oakgpu_search_many", "oakgpu_search_agent", "oakgpu_agent_networks_clear", "oakgpu_bandit_replay", "oakgpu_bandit_select_run", "oakgpu_solve_matrix",
    "oakgpu_segment_mean_dev", "oakgpu_comm_unique_id", "oakgpu_comm_create", "oakgpu_comm_destroy", "oakgpu_all_gather_dev",
    "oakgpu_root_steps_create", "oakgpu_root_steps_destroy", "oakgpu_root_steps_launch_dev", "oakgpu_root_steps_capacity", "oakgpu_root_steps_reserve",
    "oakgpu_endless_battle_check", "oakgpu_frames_size", "oakgpu_frames_write", "oakgpu_frames_read", "oakgpu_selfplay_game", "oakgpu_selfplay_games", "oakgpu_poke_engine_eval_dev", "oakgpu_poke_engine_eval",
]


class RolloutBatch(C.Structure):      # oakgpu_rollout_batch (include/oakgpu.h)
    _fields_ = [("battles", C.c_void_p), ("durations", C.c_void_p), ("results_in", C.c_void_p), ("prng_state", C.c_void_p),
                ("n", C.c_uint32), ("results_out", C.c_void_p), ("steps_out", C.c_void_p), ("values_out", C.c_void_p),
                ("battles_out", C.c_void_p), ("durations_out", C.c_void_p)]


class SearchParams(C.Structure):      # oakgpu_search_params (include/oakgpu.h)
    _fields_ = [("iterations", C.c_uint64), ("batch", C.c_uint32), ("ucb_c", C.c_float), ("bandit", C.c_int32),
                ("eval", C.c_int32), ("max_depth", C.c_uint32), ("root_rolls", C.c_uint32), ("other_rolls", C.c_uint32),
                ("seed", C.c_uint64), ("matrix_ucb", C.c_int32), ("mucb_delay", C.c_uint32), ("mucb_minimum", C.c_uint32),
                ("mucb_c", C.c_float), ("exp3_alpha", C.c_float), ("duration_us", C.c_uint64)]


class Agent(C.Structure):             # oakgpu_agent
    _fields_ = [("budget", C.c_char_p), ("bandit", C.c_char_p), ("eval", C.c_char_p), ("matrix_ucb", C.c_char_p),
                ("discrete", C.c_int), ("table", C.c_int)]


class SearchOutput(C.Structure):      # oakgpu_search_output
    _fields_ = [("m", C.c_uint8), ("n", C.c_uint8), ("p1_choices", C.c_uint8 * 9), ("p2_choices", C.c_uint8 * 9),
                ("visit_matrix", C.c_uint64 * 81), ("value_matrix", C.c_double * 81), ("iterations", C.c_uint64),
                ("empirical_value", C.c_double), ("initial_value", C.c_double), ("p1_empirical", C.c_double * 9),
                ("p2_empirical", C.c_double * 9), ("nodes", C.c_uint64), ("total_depth", C.c_uint64),
                ("duration_us", C.c_double), ("nash_value", C.c_double), ("p1_nash", C.c_double * 9), ("p2_nash", C.c_double * 9),
                ("p1_logit", C.c_double * 9), ("p2_logit", C.c_double * 9), ("p1_prior", C.c_double * 9), ("p2_prior", C.c_double * 9)]

class FrameUpdate(C.Structure):       # oakgpu_frame_update
    _fields_ = [("m", C.c_uint8), ("n", C.c_uint8), ("c1", C.c_uint8), ("c2", C.c_uint8), ("iterations", C.c_uint32),
                ("empirical_value", C.c_double), ("nash_value", C.c_double), ("p1_empirical", C.c_double * 9),
                ("p1_nash", C.c_double * 9), ("p2_empirical", C.c_double * 9), ("p2_nash", C.c_double * 9)]


class SelfplayParams(C.Structure):    # oakgpu_selfplay_params
    _fields_ = [("search", SearchParams), ("policy_mode", C.c_char * 16), ("policy_temp", C.c_double), ("policy_min", C.c_double),
                ("max_battle_length", C.c_uint32), ("seed", C.c_uint64), ("keep_node", C.c_int32), ("nodes_kept", C.c_uint32)]


# include/pkmn.h: the libpkmn-named single-battle ABI (batch-of-one wrappers, pkmn_shim.hip)
PKMN_SYMBOLS = [
    "pkmn_gen1_battle_update", "pkmn_gen1_battle_choices", "pkmn_gen1_battle_options_set",
    "pkmn_gen1_battle_options_chance_actions", "pkmn_gen1_battle_options_chance_durations",
    "pkmn_result_type", "pkmn_result_p1", "pkmn_result_p2",
]

_lib = None


class OakGpuError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OakGpuError("liboakgpu.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    lib.oakgpu_create.argtypes = [C.POINTER(vp), i32]
    lib.oakgpu_destroy.argtypes = [vp]
    lib.oakgpu_destroy.restype = None
    lib.oakgpu_last_error.restype = C.c_char_p
    lib.oakgpu_set_stream.argtypes = [vp, vp]
    lib.oakgpu_get_stream.argtypes = [vp]
    lib.oakgpu_get_stream.restype = vp
    lib.oakgpu_synchronize.argtypes = [vp]
    lib.oakgpu_set_kernel_timing.argtypes = [vp, i32]
    lib.oakgpu_get_leaf_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.oakgpu_set_playouts_per_lane.argtypes = [vp, i32]
    lib.oakgpu_set_regroup.argtypes = [vp, i32, i32, i32]
    lib.oakgpu_set_tail_pack.argtypes = [vp, i32, i32, i32]
    lib.oakgpu_set_queue_order.argtypes = [vp, i32]
    lib.oakgpu_set_spread.argtypes = [vp, i32]
    lib.oakgpu_set_migration.argtypes = [vp, i32, i32, i32]
    lib.oakgpu_set_migration_window.argtypes = [vp, i32]
    lib.oakgpu_set_standstill_skip.argtypes = [vp, i32]
    lib.oakgpu_get_queue_counters.argtypes = [vp, vp]
    lib.oakgpu_set_rollout_engine.argtypes = [vp, i32]
    lib.oakgpu_rollout_dev.argtypes = [vp, vp, vp, vp, vp, u32, u32, i32, vp, vp, vp, vp, vp]
    lib.oakgpu_rollout.argtypes = [vp, vp, vp, vp, vp, u32, u32, i32, vp, vp, vp, vp, vp]
    lib.oakgpu_rollout_group_dev.argtypes = [vp, C.POINTER(RolloutBatch), u32, u32, i32]
    lib.oakgpu_rollout_group.argtypes = [vp, C.POINTER(RolloutBatch), u32, u32, i32]
    lib.oakgpu_mt19937_fill.argtypes = [u32, u64, vp, C.c_size_t]
    lib.oakgpu_rollout_draws_dev.argtypes = [vp, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, u32, i32, vp, vp, vp, vp, vp, vp]
    lib.oakgpu_rollout_shared_device.argtypes = [vp, vp, vp, C.c_uint8, vp, u32, u32, u32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.oakgpu_poke_engine_eval_dev.argtypes = [vp, vp, u32, C.c_float, vp, vp]
    lib.oakgpu_poke_engine_eval.argtypes = [vp, vp, u32, C.c_float, vp, vp]
    lib.oakgpu_tree_step_dev.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, vp, vp, vp, vp, vp]
    lib.oakgpu_search.argtypes = [vp, vp, vp, vp, C.c_uint8, C.POINTER(SearchParams), C.POINTER(SearchOutput)]
    lib.oakgpu_heap_create.argtypes = [C.POINTER(vp)]
    lib.oakgpu_heap_destroy.argtypes = [vp]
    lib.oakgpu_heap_destroy.restype = None
    lib.oakgpu_heap_empty.argtypes = [vp]
    lib.oakgpu_heap_clear.argtypes = [vp]
    lib.oakgpu_heap_clear.restype = None
    lib.oakgpu_heap_kind.argtypes = [vp]
    lib.oakgpu_heap_nodes.argtypes = [vp]
    lib.oakgpu_heap_nodes.restype = u64
    lib.oakgpu_heap_update.argtypes = [vp, C.c_uint8, C.c_uint8, vp]
    lib.oakgpu_heap_root_stats.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.oakgpu_heap_check_shards.argtypes = [vp]
    lib.oakgpu_heap_check_shards.restype = u64
    lib.oakgpu_heap_selftest.argtypes = [u32, u32, u64, i32, C.POINTER(u64 * 4)]
    lib.oakgpu_heap_child_stats.argtypes = [vp, C.c_uint8, C.c_uint8, vp, i32, vp, vp, vp, vp]
    lib.oakgpu_search_heap.argtypes = [vp, vp, vp, vp, vp, C.c_uint8, C.POINTER(SearchParams), C.POINTER(SearchOutput), C.POINTER(SearchOutput)]
    lib.oakgpu_search_many.argtypes = [C.POINTER(vp), vp, C.POINTER(vp), vp, vp, vp, C.POINTER(SearchParams), u32, i32, C.POINTER(SearchOutput)]
    lib.oakgpu_search_agent_heap.argtypes = [vp, vp, vp, vp, C.c_uint8, C.POINTER(Agent), u32, u64, C.POINTER(SearchOutput), C.POINTER(SearchOutput)]
    lib.oakgpu_segment_mean_dev.argtypes = [vp, vp, u32, u32, vp]
    lib.oakgpu_root_steps_create.argtypes = [vp, u32, u32, u32, u32, C.POINTER(vp)]
    lib.oakgpu_root_steps_destroy.argtypes = [vp]
    lib.oakgpu_root_steps_destroy.restype = None
    lib.oakgpu_root_steps_launch_dev.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp]
    lib.oakgpu_root_steps_capacity.argtypes = [vp, C.POINTER(u32)]
    lib.oakgpu_root_steps_reserve.argtypes = [vp, u64]
    lib.oakgpu_comm_unique_id.argtypes = [vp]
    lib.oakgpu_comm_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    lib.oakgpu_comm_destroy.argtypes = [vp]
    lib.oakgpu_comm_destroy.restype = None
    lib.oakgpu_all_gather_dev.argtypes = [vp, vp, vp, vp, C.c_size_t]
    lib.oakgpu_endless_battle_check.argtypes = [vp]
    lib.oakgpu_frames_size.restype = C.c_size_t
    lib.oakgpu_frames_size.argtypes = [C.POINTER(FrameUpdate), u32]
    lib.oakgpu_frames_write.argtypes = [vp, C.c_uint8, C.POINTER(FrameUpdate), u32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.oakgpu_frames_read.argtypes = [vp, C.c_size_t, vp, C.POINTER(C.c_uint8), C.POINTER(FrameUpdate), u32, C.POINTER(u32), C.POINTER(C.c_size_t)]
    lib.oakgpu_selfplay_game.argtypes = [vp, vp, vp, u64, C.POINTER(SelfplayParams), vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(u32),
                                         C.POINTER(C.c_uint8)]
    lib.oakgpu_selfplay_games.argtypes = [C.POINTER(vp), vp, vp, C.POINTER(u64), C.POINTER(SelfplayParams), u32, i32, vp, C.c_size_t, C.POINTER(C.c_size_t),
                                          C.POINTER(u32), C.POINTER(C.c_uint8)]
    lib.oakgpu_search_agent.argtypes = [vp, vp, vp, C.c_uint8, C.POINTER(Agent), u32, u64, C.POINTER(SearchOutput)]
    lib.oakgpu_agent_networks_clear.argtypes = [vp]
    lib.oakgpu_agent_networks_clear.restype = None
    lib.oakgpu_solve_matrix.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    lib.oakgpu_bandit_select_run.argtypes = [i32, C.c_float, C.c_float, u32, vp, vp, vp, u32, vp, vp, vp, vp]
    lib.oakgpu_bandit_replay.argtypes = [i32, C.c_float, C.c_float, u32, vp, u32, vp, vp, vp, vp, vp, vp]
    lib.oakgpu_update_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, vp]
    lib.oakgpu_update.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, vp]
    lib.oakgpu_choices_dev.argtypes = [vp, vp, vp, i32, vp, vp, u32]
    lib.oakgpu_choices.argtypes = [vp, vp, vp, i32, vp, vp, u32]
    lib.oakgpu_init_battles_dev.argtypes = [vp, vp, vp, u32, i32, vp, vp, vp]
    lib.oakgpu_init_battles.argtypes = [vp, vp, vp, u32, i32, vp, vp, vp]
    lib.oakgpu_set_ou_pools.argtypes = [vp, vp, i32, vp, vp]
    lib.oakgpu_random_ou_battles_dev.argtypes = [vp, u64, u32, vp, vp, vp, vp]
    lib.oakgpu_net_load.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    lib.oakgpu_net_load_memory.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp)]
    lib.oakgpu_net_free.argtypes = [vp, vp]
    lib.oakgpu_net_free.restype = None
    lib.oakgpu_net_shape.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.oakgpu_net_set_main_precision.argtypes = [vp, C.c_int]
    lib.oakgpu_net_main_precision.argtypes = [vp, C.POINTER(C.c_int)]
    lib.oakgpu_leaf_eval_dev.argtypes = [vp, vp, vp, vp, u32, vp, vp]
    lib.oakgpu_leaf_eval.argtypes = [vp, vp, vp, vp, u32, vp, vp]
    lib.oakgpu_leaf_eval_cached_dev.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp]
    lib.oakgpu_leaf_cache_last_count.argtypes = [vp, C.POINTER(u32)]
    lib.oakgpu_leaf_eval_policy_dev.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]
    lib.oakgpu_leaf_eval_policy.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise OakGpuError("liboakgpu: %s (code %d)" % (load().oakgpu_last_error().decode(), rc))
