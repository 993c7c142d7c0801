"""ctypes binding of libsrlx.so (C ABI declared in include/srlx.h).

The library is built in-tree by `python __graft_entry__.py` / `make -C simple_distributed_rl_amd/csrc`
and is the ONLY implementation of the device path: a missing library is an error, never a fallback.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRLX_LIB") or os.path.join(_HERE, "libsrlx.so")  # SRLX_LIB: another BUILD of libsrlx (A/B timing of kernel variants on one box)

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The engines synchronise their few streams with
# events many times per step, and an event wait that crosses hardware queues costs tens of microseconds: with 3+ queues the
# placement of the learner's streams decides between a 0.5 ms and a 1.4 ms update.  Two queues keep
# actors and learner concurrent and make the placement irrelevant.  Read by the runtime when it initialises, so it is set
# before the first device call of the process (setdefault: an explicit choice of the user wins).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

OK = 0
ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_UNIFORMS_EXHAUSTED, ERR_UNSUPPORTED = -1, -2, -3, -4, -5
PRIO_NONE, PRIO_F64, PRIO_F32, PRIO_RAW = 0, 1, 2, 3

c_i64 = ctypes.c_int64
c_f64 = ctypes.c_double
c_u64 = ctypes.c_uint64
c_int = ctypes.c_int
c_p = ctypes.c_void_p


class SrlxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libsrlx error {status}: {message}")
        self.status = status


class UniformsExhausted(SrlxError):
    pass


# name -> (restype, argtypes); mirrors include/srlx.h one to one (tests/test_abi.py checks the set)
SIGNATURES = {
    "srlx_last_error": (ctypes.c_char_p, []),
    "srlx_version": (c_int, []),
    "srlx_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "srlx_device_info": (c_int, [c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_i64)]),
    "srlx_per_create": (c_int, [ctypes.POINTER(c_p), c_i64, c_f64, c_f64, c_f64, c_int, c_f64, c_int]),
    "srlx_per_destroy": (c_int, [c_p]),
    "srlx_per_set_has_duplicate": (c_int, [c_p, c_int]),
    "srlx_per_set_update_counter": (c_int, [c_p, c_p]),
    "srlx_per_set_add_counters": (c_int, [c_p, c_p, c_p]),
    "srlx_per_clear": (c_int, [c_p, c_p]),
    "srlx_per_length": (c_i64, [c_p]),
    "srlx_per_capacity": (c_i64, [c_p]),
    "srlx_per_add": (c_int, [c_p, c_i64, c_p, c_int, c_int, c_p]),
    "srlx_per_sample": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_int, c_p]),
    "srlx_per_sample_after_adds": (c_int, [c_p, c_i64, c_p, c_int, c_i64, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "srlx_per_sample_after_adds_mt": (c_int, [c_p, c_i64, c_p, c_int, c_i64, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_per_sample_keyed": (c_int, [c_p, c_i64, c_p, c_u64, c_p, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "srlx_per_sample_gather_train": (c_int, [c_p, c_p, c_i64, c_p, c_u64, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_per_update": (c_int, [c_p, c_i64, c_p, c_p, c_int, c_int, c_p]),
    "srlx_per_set_range": (c_int, [c_p, c_i64, c_i64, c_p, c_int, c_int, c_p]),
    "srlx_per_backup": (c_int, [c_p, ctypes.POINTER(c_f64), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_p]),
    "srlx_per_restore": (c_int, [c_p, c_f64, c_i64, c_i64, c_p]),
    "srlx_per_restore_resized": (c_int, [c_p, c_i64, c_i64, c_p]),
    "srlx_per_tree_ptr": (c_int, [c_p, ctypes.POINTER(c_p), ctypes.POINTER(c_i64)]),
    "srlx_per_max_priority": (c_int, [c_p, c_p, c_p]),
    "srlx_per_state_ptr": (c_int, [c_p, ctypes.POINTER(c_p)]),
    "srlx_per_refresh": (c_int, [c_p, c_p]),
    "srlx_rng_uniform": (c_int, [ctypes.c_uint64, c_p, c_i64, c_p, c_p]),
    "srlx_rng_permutation": (c_int, [ctypes.c_uint64, c_p, c_i64, c_p, c_p]),
    "srlx_rng_permutations": (c_int, [ctypes.c_uint64, c_p, c_i64, c_int, c_p, c_p]),
    "srlx_store_actor_td": (c_int, [c_p, c_i64, c_i64, c_p, c_int, c_p, c_f64, c_f64, c_int, c_int, c_p, c_p]),
    "srlx_pack_frames": (c_int, [c_p, c_p, c_i64, c_i64, c_p, c_p, c_p]),
    "srlx_store_create": (c_int, [ctypes.POINTER(c_p), c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint64, c_int]),
    "srlx_store_destroy": (c_int, [c_p]),
    "srlx_store_item_len": (c_i64, [c_p]),
    "srlx_store_per_capacity": (c_i64, [c_p]),
    "srlx_store_reset_all": (c_int, [c_p, c_p, c_p]),
    "srlx_store_stack_current": (c_int, [c_p, c_p, c_p]),
    "srlx_store_commit_step": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_store_commit_step_ex": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_p, c_p]),
    "srlx_store_commit_step_at": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_store_set_item_slack": (c_int, [c_p, c_int]),
    "srlx_store_commit_step_packed": (c_int, [c_p, c_p, c_i64, c_i64, c_int, c_p, c_p, c_p, c_p, c_int, c_p]),
    "srlx_store_advance": (c_int, [c_p, c_p]),
    "srlx_store_views": (c_int, [c_p, ctypes.POINTER(c_p), ctypes.POINTER(c_p), ctypes.POINTER(c_p)]),
    "srlx_store_gather_nstep": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_store_locate": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "srlx_agent57_ucb_step": (c_int, [c_i64, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f64, c_f64, c_p]),
    "srlx_store_obs_base": (c_int, [c_p, ctypes.POINTER(c_p), ctypes.POINTER(c_i64)]),
    "srlx_store_frame_table_current": (c_int, [c_p, c_p, c_p]),
    "srlx_store_gather_items": (c_int, [c_p, c_i64, c_p, c_int, c_int, c_p, c_p, c_p, c_p, c_p]),
    "srlx_store_gather_train": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_store_gather_obs": (c_int, [c_p, c_i64, c_int, c_int, c_p, c_p]),
    "srlx_qnet_create": (c_int, [ctypes.POINTER(c_p), c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_int]),
    "srlx_qnet_destroy": (c_int, [c_p]),
    "srlx_qnet_bind": (c_int, [c_p, c_p]),
    "srlx_qnet_forward_u8": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p]),
    "srlx_qnet_forward_convs_multi_u8": (c_int, [c_p, c_int, c_i64, c_p, c_p, c_p]),
    "srlx_qnet_forward_dense_planes": (c_int, [c_p, c_i64, c_p, c_p]),
    "srlx_qnet_forward_convs_u8": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p]),
    "srlx_qnet_forward_u8_policy": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_u64, c_p, c_p, c_p, c_p, c_p]),
    "srlx_qnet_actor_sets_enable": (c_int, [c_p]),
    "srlx_qnet_actor_set_planes": (c_int, [c_p, c_int, ctypes.POINTER(c_p)]),
    "srlx_qnet_actor_set_select": (c_int, [c_p, c_int]),
    "srlx_qnet_publish": (c_int, [c_p, c_p, c_int, c_int, c_p, c_p]),
    "srlx_debug_stamp": (c_int, [c_p, c_int, c_p]),
    "srlx_stream_create": (c_int, [c_int, ctypes.POINTER(c_p)]),
    "srlx_stream_destroy": (c_int, [c_p]),
    "srlx_qnet_set_fc1_branch": (c_int, [c_p, c_int]),
    "srlx_qnet_set_stamp_buffer": (c_int, [c_p, c_p]),
    "srlx_qnet_set_td_event": (c_int, [c_p, c_p]),
    "srlx_qnet_set_priority_sink": (c_int, [c_p, c_p, c_i64, c_p, c_p, c_int]),
    "srlx_qnet_set_sink_wait": (c_int, [c_p, c_p]),
    "srlx_qnet_set_sink_done": (c_int, [c_p, c_p]),
    "srlx_qnet_set_sink_stream": (c_int, [c_p, c_p]),
    "srlx_qnet_set_main_first": (c_int, [c_p, c_int]),
    "srlx_qnet_set_dgrad_split": (c_int, [c_p, c_int]),
    "srlx_qnet_fuse_adam_fc1_planes": (c_int, [c_p, c_p]),
    "srlx_qnet_set_pack_sticky": (c_int, [c_p, c_int]),
    "srlx_qnet_set_fc1_neighbour": (c_int, [c_p, c_int]),
    "srlx_qnet_set_planes_small": (c_int, [c_p, c_int, c_p]),
    "srlx_qnet_weights_changed": (c_int, [c_p]),
    "srlx_qnet_enable_training": (c_int, [c_p, c_i64]),
    "srlx_qnet_set_probe": (c_int, [c_p, c_p, c_p]),
    "srlx_qnet_set_probe_fc1": (c_int, [c_p, c_p, c_p]),
    "srlx_qnet_set_fc1_span": (c_int, [c_p, c_p]),
    "srlx_qnet_enable_fc1_planes": (c_int, [c_p]),
    "srlx_qnet_refresh_fc1_planes": (c_int, [c_p, c_p, c_p, c_p]),
    "srlx_qnet_invalidate_fc1_planes": (c_int, [c_p]),
    "srlx_qnet_set_debug": (c_int, [c_p, c_p]),
    "srlx_qnet_range_flags": (c_int, [c_p, c_p]),
    "srlx_qnet_set_side_stream": (c_int, [c_p, c_p]),
    "srlx_qnet_fuse_adam_fc1": (c_int, [c_p, c_p, c_p, c_f64, c_f64, c_f64, c_f64, c_p]),
    "srlx_qnet_fuse_adam_rest": (c_int, [c_p, c_p, c_p, c_p]),
    "srlx_qnet_backward_u8": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "srlx_qnet_backward_convs_u8": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "srlx_qnet_backward_td_u8": (c_int, [c_p, c_i64, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f64, c_f64, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_qnet_forward_f32": (c_int, [c_p, c_i64, c_p, c_p, c_p]),
    "srlx_qnet_bind_noisy": (c_int, [c_p, c_p, c_u64]),
    "srlx_qnet_bind_noisy_grads": (c_int, [c_p, c_p]),
    "srlx_qnet_redraw_rows": (c_int, [c_p, c_i64, c_i64, c_p, c_p]),
    "srlx_qnet_noisy_effective": (c_int, [c_p, c_int, c_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_p]),
    "srlx_policy_epsilon_greedy": (c_int, [c_i64, c_int, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_image_preprocess": (c_int, [c_i64, c_int, c_int, c_int, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p, c_int, c_f64, c_p]),
    "srlx_episode_account": (c_int, [c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p]),
    "srlx_synth_env_step": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "srlx_synth_env_step_at": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p]),
    "srlx_nstep_td_huber_priority": (
        c_int,
        [c_i64, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f64, c_f64, c_int, c_int, c_p, c_p, c_p, c_p, c_p],
    ),
    "srlx_nstep_td_huber_priority_packed": (
        c_int,
        [c_i64, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f64, c_f64, c_int, c_int, c_p, c_p, c_p, c_p, c_p],
    ),
    "srlx_dqn_target": (c_int, [c_i64, c_int, c_p, c_p, c_p, c_p, c_p, c_f64, c_p, c_int, c_int, c_int, c_p, c_p]),
    "srlx_gae_scan": (c_int, [c_i64, c_i64, c_p, c_p, c_p, c_p, c_f64, c_f64, c_p, c_p]),
    "srlx_adam_step": (c_int, [c_int, c_p, c_p, c_p, c_p, c_p, c_f64, c_f64, c_f64, c_f64, c_p, c_p]),
    "srlx_rank_create": (c_int, [ctypes.POINTER(c_p), c_i64, c_int]),
    "srlx_rank_destroy": (c_int, [c_p]),
    "srlx_rank_set": (c_int, [c_p, c_i64, c_p, c_p, c_i64, c_p]),
    "srlx_rank_select": (c_int, [c_p, c_i64, c_i64, c_p, c_p, c_p]),
    "srlx_rank_priorities": (c_int, [c_p, ctypes.POINTER(c_p)]),
    "srlx_ppo_normal_act": (c_int, [c_i64, c_p, c_p, c_f64, c_f64, c_u64, c_p, c_int, c_p, c_p, c_p]),
    "srlx_ppo_loss_normal": (c_int, [c_i64, c_int, c_p, c_p, c_f64, c_f64, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_f64, c_int, c_f64, c_f64, c_f64,
                                     c_p, c_p, c_p, c_p, c_p]),
    "srlx_ppo_loss_logpi": (c_int, [c_i64, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_f64, c_int, c_f64, c_f64, c_f64, c_p, c_p, c_p, c_p]),
    "srlx_pendulum_step": (c_int, [c_i64, c_p, c_p, c_p, c_i64, c_u64, c_p, c_p, c_p, c_p, c_p]),
    "srlx_ppo_net_param_count": (c_int, [c_int, c_int]),
    "srlx_ppo_net_partials_floats": (c_int, [c_int, c_int]),
    "srlx_ppo_net_rollout_max_horizon": (c_int, [c_int]),
    "srlx_ppo_net_forward": (c_int, [c_i64, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_ppo_net_rollout": (c_int, [c_i64, c_i64, c_int, c_p, c_p, c_p, c_p, c_i64, c_u64, c_p, c_u64, c_p, c_f64, c_f64, c_f64, c_f64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                     c_p, c_p, c_p]),
    "srlx_ppo_net_minibatch": (c_int, [c_i64, c_p, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f64, c_f64, c_int, c_int, c_f64, c_int, c_f64, c_f64, c_f64, c_p, c_p, c_p,
                                       c_p]),
    "srlx_ppo_net_adam": (c_int, [c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_f64, c_f64, c_f64, c_f64, c_f64, c_f64, c_p]),
    "srlx_ngu_create": (c_int, [ctypes.POINTER(c_p), c_i64, c_int, c_i64, c_int, c_f64, c_f64, c_f64, c_int]),
    "srlx_ngu_destroy": (c_int, [c_p]),
    "srlx_ngu_reset": (c_int, [c_p, c_p]),
    "srlx_ngu_counts": (c_int, [c_p, ctypes.POINTER(c_p)]),
    "srlx_ngu_episodic_reward": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_ngu_lifelong_reward": (c_int, [c_i64, c_int, c_p, c_p, c_f64, c_p, c_p]),
    "srlx_agent57_seq_td": (c_int, [c_i64, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f64, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_qnet_bind_uvfa": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "srlx_qnet_set_uvfa_inputs": (c_int, [c_p, c_p, c_p, c_p, c_p]),
    "srlx_qnet_fuse_adam_uvfa": (c_int, [c_p, c_p, c_p, c_p]),
    "srlx_qnet_set_td_extras": (c_int, [c_p, c_p, c_p]),
    "srlx_qnet_set_head_mode": (c_int, [c_p, c_int, c_int, c_p, c_p, c_f64]),
    "srlx_agent57_policy": (c_int, [c_i64, c_int, c_p, c_p, c_p, c_p, c_p, c_f64, c_f64, c_u64, c_p, c_p, c_p, c_p]),
    "srlx_agent57_post_step": (c_int, [c_i64] + [c_p] * 16),
    "srlx_agent57_begin_episodes": (c_int, [c_i64, c_int, c_p, c_u64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_agent57_gather_inputs": (c_int, [c_i64, c_i64] + [c_p] * 21),
    "srlx_agent57_pack_record": (c_int, [c_i64] + [c_p] * 11),
    "srlx_agent57_unpack_fields": (c_int, [c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p]),
    "srlx_agent57_emb_tail": (c_int, [c_i64, c_int, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_f64, c_f64, c_f64, c_f64, c_f64, c_p, c_p, c_p, c_p]),
    "srlx_agent57_rnd_tail": (c_int, [c_i64, c_int, c_i64] + [c_p] * 12 + [c_f64] * 5 + [c_p, c_p, c_p, c_p]),
    "srlx_agent57_priority": (c_int, [c_i64, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
}
OBS_U8, OBS_F32 = 0, 1
PRIO_NONE_MASKED = 4
PRIO_EST_F32 = 5

_lib = None
_lock = threading.Lock()


def lib():
    """Loads libsrlx.so once.  Raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C simple_distributed_rl_amd/csrc`). "
                "simple_distributed_rl_amd has no CPU fallback for the device path."
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("SRLX_LIB") and not hasattr(L, name):
                continue  # an OLDER build selected for an A/B timing may lack the newest entry points (the in-tree library must export every one)
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
        return L


def check(status):
    if status == OK:
        return
    msg = lib().srlx_last_error().decode("utf-8", "replace")
    if status == ERR_UNIFORMS_EXHAUSTED:
        raise UniformsExhausted(status, msg)
    raise SrlxError(status, msg)


def device_count():
    n = c_int(0)
    check(lib().srlx_device_count(ctypes.byref(n)))
    return n.value


def device_info(device=0):
    name = ctypes.create_string_buffer(64)
    cu = c_int(0)
    mem = c_i64(0)
    check(lib().srlx_device_info(device, name, 64, ctypes.byref(cu), ctypes.byref(mem)))
    return dict(arch=name.value.decode(), cu_count=cu.value, hbm_bytes=mem.value)


def np_ptr(a):
    return a.ctypes.data_as(c_p)


def tptr(t):
    """device (or host) pointer of a torch tensor as c_void_p; None passes NULL."""
    return None if t is None else c_p(t.data_ptr())


def torch_stream_ptr():
    """hipStream_t of torch's current stream (so srlx kernels order with torch ops)."""
    import torch

    return c_p(torch.cuda.current_stream().cuda_stream)
