"""ctypes loader for libdrlgx.so (the C ABI of include/drlgx.h).  Fails loudly when the HIP library
has not been built — there is no Python or CPU fallback."""
import ctypes as C
import os

from .config import DrlgxConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.environ.get("DRLGX_LIB_DEV", os.path.join(_HERE, "libdrlgx.so"))  # DRLGX_LIB_DEV: kernel-variant experiments only
_lib = None

N_TIMERS = 8


class DrlgxError(RuntimeError):
    pass


class CsrCache(C.Structure):
    """struct drlgx_csr_cache (include/drlgx.h): device addresses of the per-graph normalisation / CSR / AX cache."""
    _fields_ = [(n, C.c_void_p) for n in ("deg", "selfw", "ax", "ptr_dst", "end_dst", "ptr_src", "end_src", "nbr_dst", "nbr_src", "wn_dst",
                                          "wn_src")]
    NODE_WORDS = dict(deg=1, selfw=1, ax=8, ptr_dst=1, end_dst=1, ptr_src=1, end_src=1)  # 4-byte words per node
    EDGE_WORDS = dict(nbr_dst=1, nbr_src=1, wn_dst=1, wn_src=1)                          # ... per edge


def lib_path():
    return _PATH


# every symbol declared in include/drlgx.h
SYMBOLS = [
    "drlgx_create", "drlgx_destroy", "drlgx_set_stream", "drlgx_synchronize", "drlgx_strerror", "drlgx_last_error",
    "drlgx_status_host", "drlgx_status_fetch_host", "drlgx_reset_host", "drlgx_step", "drlgx_utility", "drlgx_uncertainty_em", "drlgx_explored", "drlgx_metrics", "drlgx_cov_array",
    "drlgx_stage_reset_host", "drlgx_stage_set_prior_information_host", "drlgx_stage_set_prior_pose_host", "drlgx_stage_move", "drlgx_stage_measure", "drlgx_stage_add_measurements", "drlgx_stage_optimize",
    "drlgx_stage_update_map", "drlgx_set_planner_parameter", "drlgx_set_fixed_landmarks_host", "drlgx_fm2_update", "drlgx_step_plan", "drlgx_step_plans",
    "drlgx_line_plan", "drlgx_lookahead", "drlgx_lookahead_bounded", "drlgx_graph_capacity", "drlgx_graph", "drlgx_get_counts_host", "drlgx_counts",
    "drlgx_get_poses_host", "drlgx_get_landmarks_host", "drlgx_get_cov_traces_host", "drlgx_vm_shape",
    "drlgx_get_virtual_map_host", "drlgx_get_ground_truth_host", "drlgx_get_adjacency_host", "drlgx_get_factors_host",
    "drlgx_get_landmark_order_host", "drlgx_snapshot", "drlgx_restore", "drlgx_timing_enable",
    "drlgx_timing_read_host", "drlgx_debug_phase_clocks_host", "drlgx_debug_gemm_tile_rows", "drlgx_debug_map_form", "drlgx_inc_stats_host", "drlgx_gcn_workspace_bytes", "drlgx_gcn_forward", "drlgx_gcn_forward_batched", "drlgx_gcn_backward",
    "drlgx_replay_collate", "drlgx_replay_collate_pair", "drlgx_dqn_targets", "drlgx_dqn_loss_grad", "drlgx_dqn_arena_bytes", "drlgx_dqn_arena_views", "drlgx_dqn_prepare", "drlgx_dqn_forward_backward", "drlgx_replay_cache_csr", "drlgx_gcn_collate_csr", "drlgx_gcn_forward_prebuilt", "drlgx_adam_step", "drlgx_adam_step_scaled", "drlgx_normalise_rewards",
    "drlgx_segment_softmax", "drlgx_segment_softmax_backward", "drlgx_mean_pool", "drlgx_mean_pool_backward",
]


def lib():
    """Load libdrlgx.so and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own ROCm runtime: import it first so that libdrlgx.so binds to the SAME
    # libamdhip64 (device pointers and streams are shared with torch tensors)
    import torch  # noqa: F401
    if not os.path.exists(_PATH):
        raise DrlgxError(
            "libdrlgx.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % _PATH)
    L = C.CDLL(_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)
    L.drlgx_create.argtypes = [C.POINTER(DrlgxConfig), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.drlgx_destroy.argtypes = [vp]
    L.drlgx_set_stream.argtypes = [vp, vp]
    L.drlgx_synchronize.argtypes = [vp]
    L.drlgx_strerror.restype = C.c_char_p
    L.drlgx_strerror.argtypes = [C.c_int]
    L.drlgx_last_error.restype = C.c_char_p
    L.drlgx_last_error.argtypes = [vp]
    L.drlgx_status_host.argtypes = [vp]
    L.drlgx_status_fetch_host.argtypes = [vp, vp, C.c_size_t, vp]
    L.drlgx_reset_host.argtypes = [vp, C.c_int, ip, C.POINTER(C.c_uint32), dp]
    L.drlgx_step.argtypes = [vp, vp, vp]
    L.drlgx_step_plan.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.drlgx_step_plans.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.drlgx_stage_reset_host.argtypes = [vp, C.c_int, ip, C.POINTER(C.c_uint32), dp]
    L.drlgx_stage_set_prior_information_host.argtypes = [vp, C.c_int, dp]
    L.drlgx_stage_set_prior_pose_host.argtypes = [vp, C.c_int, dp]
    L.drlgx_stage_move.argtypes = [vp, vp, vp]
    L.drlgx_stage_measure.argtypes = [vp, vp, vp, vp, vp]
    L.drlgx_stage_add_measurements.argtypes = [vp, vp, vp, vp, vp]
    L.drlgx_stage_optimize.argtypes = [vp, vp]
    L.drlgx_stage_update_map.argtypes = [vp, vp, C.c_int]
    L.drlgx_set_planner_parameter.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
    L.drlgx_set_fixed_landmarks_host.argtypes = [vp, C.c_int, dp]
    L.drlgx_fm2_update.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp]
    L.drlgx_utility.argtypes = [vp, vp, vp]
    L.drlgx_uncertainty_em.argtypes = [vp, C.c_int, vp]
    L.drlgx_explored.argtypes = [vp, vp]
    L.drlgx_metrics.argtypes = [vp, C.c_double, vp]
    L.drlgx_cov_array.argtypes = [vp, vp, vp]
    L.drlgx_line_plan.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.drlgx_lookahead.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.drlgx_lookahead_bounded.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp]
    L.drlgx_graph_capacity.argtypes = [vp, ip, ip, ip]
    L.drlgx_graph.argtypes = [vp] + [vp] * 8
    L.drlgx_get_counts_host.argtypes = [vp, C.c_int, ip]
    L.drlgx_counts.argtypes = [vp, vp]
    L.drlgx_get_poses_host.argtypes = [vp, C.c_int, dp, dp]
    L.drlgx_get_landmarks_host.argtypes = [vp, C.c_int, ip, dp, dp]
    L.drlgx_get_cov_traces_host.argtypes = [vp, C.c_int, dp, dp]
    L.drlgx_vm_shape.argtypes = [vp, ip, ip]
    L.drlgx_get_virtual_map_host.argtypes = [vp, C.c_int, dp, dp, dp, C.POINTER(C.c_uint8)]
    L.drlgx_get_ground_truth_host.argtypes = [vp, C.c_int, dp, dp]
    L.drlgx_get_adjacency_host.argtypes = [vp, C.c_int, dp, dp]
    L.drlgx_get_factors_host.argtypes = [vp, C.c_int, ip, ip, dp, dp]
    L.drlgx_get_landmark_order_host.argtypes = [vp, ip]
    L.drlgx_snapshot.argtypes = [vp, C.c_int]
    L.drlgx_restore.argtypes = [vp, C.c_int]
    L.drlgx_timing_enable.argtypes = [vp, C.c_int]
    L.drlgx_timing_read_host.argtypes = [vp, dp, C.POINTER(C.c_int64)]
    L.drlgx_debug_phase_clocks_host.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
    L.drlgx_debug_gemm_tile_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.drlgx_debug_map_form.argtypes = [C.c_int]
    L.drlgx_inc_stats_host.argtypes = [vp, C.POINTER(C.c_int64), C.c_int]
    L.drlgx_gcn_workspace_bytes.restype = C.c_size_t
    L.drlgx_gcn_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.drlgx_gcn_forward.argtypes = [vp] + [C.c_int] * 5 + [vp] * 12
    L.drlgx_gcn_backward.argtypes = [vp] + [C.c_int] * 5 + [vp] * 15
    L.drlgx_replay_collate.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, C.c_int64, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.drlgx_replay_collate_pair.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, C.c_int64, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
    L.drlgx_gcn_forward_batched.argtypes = [vp] + [C.c_int] * 5 + [vp] * 12 + [C.c_int, vp, vp, C.c_int]
    L.drlgx_dqn_targets.argtypes = [vp, C.c_int, vp, vp, vp, C.c_double, C.c_int64, vp, vp]
    L.drlgx_dqn_loss_grad.argtypes = [vp, C.c_int, vp, vp, vp, C.c_double, vp, vp]
    i64 = C.c_int64
    L.drlgx_dqn_arena_bytes.restype = C.c_size_t
    L.drlgx_dqn_arena_bytes.argtypes = [C.c_int, i64, i64, i64, C.c_int, C.c_int, C.c_int]
    L.drlgx_dqn_arena_views.argtypes = [vp, C.c_int, i64, i64, i64, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.drlgx_dqn_prepare.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, i64, vp, vp, i64, i64, i64, vp, vp, C.c_double, vp, i64, i64, i64,
                                    C.c_int, C.c_int, C.POINTER(CsrCache)]
    L.drlgx_dqn_forward_backward.argtypes = [vp, C.c_int, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp), vp, C.c_double,
                                             C.POINTER(vp), vp, i64, i64, i64, C.c_int]
    L.drlgx_replay_cache_csr.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp, i64, vp, C.POINTER(CsrCache)]
    L.drlgx_gcn_collate_csr.argtypes = [vp, C.c_int, vp, C.POINTER(CsrCache), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    L.drlgx_gcn_forward_prebuilt.argtypes = [vp] + [C.c_int] * 5 + [vp] * 9
    L.drlgx_normalise_rewards.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.drlgx_segment_softmax.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.drlgx_segment_softmax_backward.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.drlgx_mean_pool.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp]
    L.drlgx_mean_pool_backward.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp]
    L.drlgx_adam_step.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64),
                                  C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_double]
    L.drlgx_adam_step_scaled.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64),
                                  C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_double, C.c_double]
    _lib = L
    return L


def stream_ptr(device):
    """Raw hipStream_t (int) of torch's current stream on `device`: what `torch.cuda.current_stream(device).cuda_stream`
    returns without building the Stream object (the update loop asks ~7 times per mini-batch)."""
    import torch
    idx = device.index if isinstance(device, torch.device) else device
    if idx is None:
        idx = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


def check(rc, handle=None):
    if rc != 0:
        L = lib()
        msg = L.drlgx_strerror(rc).decode()
        if handle is not None:
            extra = L.drlgx_last_error(handle).decode()
            if extra:
                msg += " (" + extra + ")"
        raise DrlgxError("drlgx error %d: %s" % (rc, msg))
