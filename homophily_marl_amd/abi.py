"""ctypes mirror of include/ssd_hip.h and the loader of the in-tree HIP library.

The native library is REQUIRED: there is no CPU fallback on the product path.  `load_library()` raises if
`libssd_hip.so` is missing or does not export every symbol the header declares.
"""
import ctypes as C
import os

ABI_VERSION = 7
MAX_AGENTS = 10
MAX_CELLS = 1024
MAX_SITES = 256

ENV_CLEANUP, ENV_HARVEST = 0, 1
RNG_TAPE, RNG_COUNTER = 0, 1
OBS_F32, OBS_BF16, OBS_U8, OBS_CODE = 0, 1, 2, 3
COLOR_SIMPLIFIED, COLOR_FULL = 0, 1
CODE_CLASS, CODE_CHANNEL_MASK = 0, 1
# ssd_policy_head.input_flags (include/ssd_hip.h, SSD_INPUT_*): the _build_inputs blocks in the reference's order
INPUT_LAST_ACTION, INPUT_AGENT_ID, INPUT_REWARD, INPUT_INC_REWARD, INPUT_DISTANCE, INPUT_AGENT_POS = 1, 2, 4, 8, 16, 32
INPUT_OTHERS_LAST_ACTION = 64     # ssd_build_inputs_flags only (the fused rollout heads do not build it)
INPUT_EXPLICIT = 0x80000000   # marks a given flag word: the empty set is INPUT_EXPLICIT alone, 0 means the shipped set
STREAM_UNIFORM, STREAM_MOVE, STREAM_WASTE, STREAM_SPAWN_ROT = 0, 1, 2, 3

SSD_OK, SSD_ERR_INVALID, SSD_ERR_DEVICE, SSD_ERR_NOMEM, SSD_ERR_UNSUPPORTED = 0, -1, -2, -3, -4


class SsdConfig(C.Structure):
    _fields_ = [
        ("env_kind", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("ascii_map", C.c_char_p),
        ("n_agents", C.c_int32), ("n_env", C.c_int32), ("view_size", C.c_int32), ("episode_limit", C.c_int32),
        ("random_spawn_point", C.c_int32), ("spawn_rotation", C.c_int32), ("obs_color", C.c_int32),
        ("rng_mode", C.c_int32), ("device", C.c_int32), ("env_id_base", C.c_uint32), ("seed", C.c_uint64),
        ("threshold_depletion", C.c_double), ("threshold_restoration", C.c_double),
        ("waste_spawn_prob", C.c_double), ("apple_respawn_prob", C.c_double),
        ("harvest_spawn_prob", C.c_double * 4),
    ]


class SsdTape(C.Structure):
    _fields_ = [
        ("move_order", C.c_void_p), ("uniforms", C.c_void_p), ("uniforms_stride", C.c_int32),
        ("waste_order", C.c_void_p), ("spawn_rot", C.c_void_p), ("spawn_order", C.c_void_p),
    ]


class SsdStepOut(C.Structure):
    _fields_ = [
        ("reward", C.c_void_p), ("clean_num", C.c_void_p), ("apple_den", C.c_void_p), ("terminated", C.c_void_p),
        ("collective_return", C.c_void_p), ("equality", C.c_void_p), ("n_draws", C.c_void_p),
    ]


class SsdObsOut(C.Structure):
    _fields_ = [
        ("obs", C.c_void_p), ("obs_format", C.c_int32), ("state", C.c_void_p), ("pos", C.c_void_p),
        ("orient", C.c_void_p), ("obs_env_stride", C.c_int64), ("obs_slot_stride", C.c_int64),
        ("obs_t_slots", C.c_int32), ("obs_code", C.c_void_p),
    ]


class SsdState(C.Structure):
    _fields_ = [
        ("grid", C.c_void_p), ("pos", C.c_void_p), ("orient", C.c_void_p), ("ep_reward", C.c_void_p),
        ("ep_step", C.c_void_p), ("epoch", C.c_void_p),
    ]


class SsdInfo(C.Structure):
    _fields_ = [
        ("n_actions", C.c_int32), ("n_apple_sites", C.c_int32), ("n_waste_sites", C.c_int32),
        ("n_spawn_points", C.c_int32), ("max_uniforms", C.c_int32), ("obs_edge", C.c_int32),
    ]


# name -> (restype, argtypes); the single source the symbol test and both loaders use.
def _sigs(prefix, with_stream):
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    st = [vp] if with_stream else []
    P = C.POINTER
    return {
        prefix + "abi_version": (C.c_int, []),
        prefix + "last_error": (C.c_char_p, []),
        prefix + "create": (C.c_int, [P(SsdConfig), P(vp)]),
        prefix + "destroy": (C.c_int, [vp]),
        prefix + "reset": (C.c_int, [vp, vp, P(SsdTape), P(SsdStepOut)] + st),
        prefix + "step": (C.c_int, [vp, vp, P(SsdTape), P(SsdStepOut)] + st),
        prefix + "observe": (C.c_int, [vp, P(SsdObsOut)] + st),
        prefix + "export_state": (C.c_int, [vp, P(SsdState)] + st),
        prefix + "import_state": (C.c_int, [vp, P(SsdState)] + st),
        prefix + "get_info": (C.c_int, [vp, P(SsdInfo)]),
        prefix + "build_inputs": (C.c_int, [i32, i32, i32, i32, vp, vp, vp, vp, f32, vp, i32, i32] + st),
        prefix + "incentive_transfer": (C.c_int, [i32, i32, i32, vp, vp, f32, f32, f32, f32,
                                                  vp, vp, vp, vp, vp, vp] + st),
    }


HIP_SIGNATURES = _sigs("ssd_", True)
HIP_SIGNATURES["ssd_step_observe"] = (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(SsdTape), C.POINTER(SsdStepOut),
                                                C.POINTER(SsdObsOut), C.c_void_p])
HIP_SIGNATURES["ssd_encoder"] = (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p])


class SsdStoreStep(C.Structure):
    _fields_ = [("t_index", C.c_void_p), ("n_env", C.c_int32), ("n_agents", C.c_int32), ("n_actions", C.c_int32), ("t_slots", C.c_int32),
                ("pos", C.c_void_p), ("orient", C.c_void_p), ("reward", C.c_void_p), ("clean_num", C.c_void_p), ("apple_den", C.c_void_p),
                ("terminated", C.c_void_p), ("actions", C.c_void_p), ("actions_inc", C.c_void_p),
                ("dst_pos", C.c_void_p), ("dst_orient", C.c_void_p), ("dst_reward", C.c_void_p), ("dst_clean_num", C.c_void_p),
                ("dst_apple_den", C.c_void_p), ("dst_actions_onehot", C.c_void_p), ("dst_terminated", C.c_void_p),
                ("dst_actions", C.c_void_p), ("dst_actions_inc", C.c_void_p),
                ("prev_actions", C.c_void_p), ("prev_reward", C.c_void_p), ("prev_actions_inc", C.c_void_p), ("ep_return", C.c_void_p),
                ("next_t_out", C.c_void_p), ("counter_inc", C.c_void_p)]


COPY_BLOCKS_MAX = 32
ADAM_MAX_JOBS = 64


class SsdRowGather(C.Structure):
    """ssd_row_gather (include/ssd_hip.h): one field of ssd_gather_rows."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_bytes", C.c_int64)]


class SsdAdamJob(C.Structure):
    """ssd_adam_job (include/ssd_hip.h): one parameter tensor of ssd_clip_adam_step."""
    _fields_ = [("param", C.c_void_p), ("offset", C.c_int64), ("numel", C.c_int32), ("segment", C.c_int32),
                ("exp_avg", C.c_void_p * 2), ("exp_avg_sq", C.c_void_p * 2), ("step", C.c_void_p * 2)]


class SsdClipAdamArgs(C.Structure):
    """ssd_clip_adam_args (include/ssd_hip.h)."""
    _fields_ = [("flat_grad", C.c_void_p), ("total", C.c_int64), ("jobs", C.c_void_p),
                ("n_jobs", C.c_int32), ("partials", C.c_void_p), ("lr_inc", C.c_float), ("lr_env", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("clip", C.c_float)]


class SsdBlockCopy(C.Structure):
    """ssd_block_copy (include/ssd_hip.h): one strided 2-D f32 block copy of ssd_copy_blocks."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("src_stride", C.c_int32),
                ("dst_stride", C.c_int32)]


class SsdBlockFill(C.Structure):
    """ssd_block_fill (include/ssd_hip.h): one 32-bit pattern fill of ssd_fill_blocks."""
    _fields_ = [("dst", C.c_void_p), ("bytes", C.c_int64), ("value", C.c_uint32), ("reserved", C.c_uint32)]


FILL_BLOCKS_MAX = 16


class SsdPolicyHead(C.Structure):
    """include/ssd_hip.h: ssd_policy_head (fused controller step, one launch per head)."""
    _fields_ = [("n_env", C.c_int32), ("n_agents", C.c_int32), ("n_actions", C.c_int32), ("input_shape", C.c_int32),
                ("pos_scale", C.c_float), ("seed", C.c_uint32),
                ("inputs", C.c_void_p), ("h", C.c_void_p), ("weights", C.c_void_p), ("avail", C.c_void_p), ("epsilon", C.c_void_p),
                ("step", C.c_void_p), ("prev_actions", C.c_void_p), ("prev_reward", C.c_void_p), ("prev_actions_inc", C.c_void_p),
                ("pos", C.c_void_p), ("actions", C.c_void_p), ("pos_pre", C.c_void_p), ("orient_pre", C.c_void_p),
                ("reward", C.c_void_p), ("clean_num", C.c_void_p), ("apple_den", C.c_void_p), ("out_actions", C.c_void_p),
                ("q_out", C.c_void_p), ("orient", C.c_void_p), ("out_actions_i32", C.c_void_p), ("pos_copy", C.c_void_p),
                ("orient_copy", C.c_void_p),
                ("t_index", C.c_void_p), ("t_slots", C.c_int32),
                ("dst_pos", C.c_void_p), ("dst_orient", C.c_void_p), ("dst_actions_onehot", C.c_void_p), ("dst_reward", C.c_void_p),
                ("dst_clean_num", C.c_void_p), ("dst_apple_den", C.c_void_p), ("dst_terminated", C.c_void_p), ("terminated", C.c_void_p),
                ("dst_actions", C.c_void_p), ("dst_actions_inc", C.c_void_p), ("prev_actions_out", C.c_void_p),
                ("prev_actions_inc_out", C.c_void_p), ("prev_reward_out", C.c_void_p), ("ep_return", C.c_void_p), ("next_t_out", C.c_void_p),
                ("precision", C.c_int32), ("env_id_base", C.c_uint32), ("feat_part", C.c_void_p), ("feat_bands", C.c_int32),
                ("lin_b", C.c_void_p), ("input_flags", C.c_uint32),
                ("next_step_out", C.c_void_p), ("t_copy_out", C.c_void_p), ("step_copy_out", C.c_void_p),
                ("recv_inc", C.c_void_p), ("recv_inc_out", C.c_void_p), ("avail_bits", C.c_uint32)]


class SsdPolicyHeadParams(C.Structure):
    """include/ssd_hip.h: ssd_policy_head_params (the reference-shaped f32 parameters of one head)."""
    _fields_ = [("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p), ("w_i", C.c_void_p * 3), ("w_h", C.c_void_p * 3), ("b_i", C.c_void_p * 3),
                ("b_h", C.c_void_p * 3), ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p), ("fc2_v_w", C.c_void_p), ("fc2_v_b", C.c_void_p),
                ("n_agents", C.c_int32), ("fc1_in", C.c_int32), ("fc2_in", C.c_int32), ("fc2_out", C.c_int32)]


class SsdPolicyEncodeArgs(C.Structure):
    """include/ssd_hip.h: ssd_policy_encode_args."""
    _fields_ = [("codes", C.c_void_p), ("code_bytes", C.c_int64), ("env_stride", C.c_int64), ("slot_stride", C.c_int64),
                ("agent_stride", C.c_int64), ("slot_t", C.c_void_p),
                ("rows", C.c_int32), ("view_edge", C.c_int32), ("n_agents", C.c_int32), ("agent_major", C.c_int32), ("precision", C.c_int32),
                ("conv_frags", C.c_void_p), ("lin_frags", C.c_void_p), ("conv_b", C.c_void_p), ("lin_b", C.c_void_p),
                ("out", C.c_void_p), ("out_stride", C.c_int32), ("part", C.c_void_p), ("slot_t_copy", C.c_void_p), ("counter_inc", C.c_void_p),
                ("alphabet", C.c_int32), ("act", C.c_void_p), ("slot_add", C.c_int32), ("layout", C.c_int32)]


class SsdTdLossArgs(C.Structure):
    """include/ssd_hip.h: ssd_td_loss_args."""
    _fields_ = [("batch", C.c_int32), ("t_slots", C.c_int32), ("n_agents", C.c_int32), ("n_actions", C.c_int32), ("sim_horizon", C.c_int32),
                ("double_q", C.c_int32),
                ("gamma_env", C.c_float), ("gamma_inc", C.c_float), ("reward_scale", C.c_float), ("incentive_ratio", C.c_float),
                ("incentive_cost", C.c_float), ("incentive", C.c_float), ("seq_len", C.c_float), ("sim_threshold", C.c_float),
                ("sim_loss_weight", C.c_float),
                ("q_env", C.c_void_p), ("q_inc", C.c_void_p), ("tq_env", C.c_void_p), ("tq_inc", C.c_void_p),
                ("actions", C.c_void_p), ("actions_inc", C.c_void_p), ("avail", C.c_void_p), ("reward", C.c_void_p), ("clean_num", C.c_void_p),
                ("terminated", C.c_void_p), ("filled", C.c_void_p), ("dens", C.c_void_p),
                ("dq_env", C.c_void_p), ("dq_inc", C.c_void_p), ("partials", C.c_void_p)]


TD_LOSS_PARTIALS = 16
POLICY_HEAD_FRAGS, POLICY_HEAD_TAIL_FLOATS = 58, 464 + 64


POLICY_TAIL_PIECES = 3


def policy_image_bytes(precision):
    """SSD_POLICY_IMAGE_BYTES: 14 K-steps of 4 * precision pieces + the resident chunk (fc2, tail, padding), 1 KiB pieces."""
    return (56 * precision + 8) * 1024


def policy_tail_piece(precision):
    """first 1 KiB piece of the f32 tail (biases, pair part of fc2) inside a head image"""
    return 56 * precision + 2 * precision


def policy_frag_piece(precision, F, term):
    """1 KiB piece of fragment F (the numbering of csrc: 2 ot + s fc1; 8 + 2 ot + s GRU input side, 12 tiles; 32 + 2 ot + s hidden
    side; 56 + s fc2) and split term `term` inside a head image (include/ssd_hip.h: consumption order)."""
    if F < 8:
        c, ot = F & 1, F >> 1
    elif F < 56:
        G = (F - 8) % 24
        c, ot = (8 if F >= 32 else 2) + 2 * (G >> 3) + (G & 1), (G & 7) >> 1
    else:
        return 56 * precision + 2 * term + (F - 56)
    return 4 * precision * c + 4 * term + ot


def policy_head_plan(n_env, n_agents, fused_with_encoder=False):
    """(workgroups per agent, compute waves per workgroup, tiles the busiest wave walks) of a head launch on the current device
    (ssd_policy_head_plan); tiles > 1 = the looped kernel instantiations."""
    lib = load_library()
    a, b, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    check(lib, lib.ssd_policy_head_plan(n_env, n_agents, int(bool(fused_with_encoder)), C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def encode_bands(V):
    return 3 if V == 31 else 1


ENCODE_LAYOUT_TOEPLITZ, ENCODE_LAYOUT_LUT = 0, 1
ENCODE_LUT_TABLE_BYTES = 3 * 64 * 6 * 4


def encode_lut_ksteps(V):
    """SSD_ENCODE_LUT_KSTEPS: K-steps of the Linear image in the class-LUT layout (4 output positions each, numbered through the bands)"""
    return 73 + 73 + 66 if V == 31 else 43


def encode_frag_bytes(V, precision, layout=ENCODE_LAYOUT_TOEPLITZ):
    """(conv image, Linear image) sizes in bytes (include/ssd_hip.h SSD_ENCODE_*): Toeplitz fragments, or the class-LUT table + the
    position-major Linear image."""
    if layout == ENCODE_LAYOUT_LUT:
        return ENCODE_LUT_TABLE_BYTES, encode_lut_ksteps(V) * 2 * precision * 1024
    units = 29 * 2 * 3 if V == 31 else 13 * 3
    return precision * 9 * 1024, units * 2 * precision * 1024


def code_agent_stride(V):
    return (V * V + 15) & ~15


HIP_SIGNATURES["ssd_policy_head_env"] = (C.c_int, [C.POINTER(SsdPolicyHead), C.c_void_p])
HIP_SIGNATURES["ssd_policy_head_inc"] = (C.c_int, [C.POINTER(SsdPolicyHead), C.c_void_p])
HIP_SIGNATURES["ssd_gru_seq_fwd"] = (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 3 + [C.c_void_p])
HIP_SIGNATURES["ssd_gru_seq_bwd"] = (C.c_int, [C.c_void_p] * 9 + [C.c_int32] * 3 + [C.c_void_p])
HIP_SIGNATURES["ssd_build_inputs_width"] = (C.c_int, [C.c_int32, C.c_int32, C.c_uint32])
HIP_SIGNATURES["ssd_build_inputs_flags"] = (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p])
HIP_SIGNATURES["ssd_clip_adam_step"] = (C.c_int, [C.POINTER(SsdClipAdamArgs), C.c_void_p])
HIP_SIGNATURES["ssd_dueling_q_fwd"] = (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 5 + [C.c_void_p])
HIP_SIGNATURES["ssd_dueling_q_bwd"] = (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 5 + [C.c_void_p])
HIP_SIGNATURES["ssd_gather_rows"] = (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p])
HIP_SIGNATURES["ssd_sample_ids"] = (C.c_int, [C.c_uint64, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p])
SAMPLE_IDS_MAX = 1024
HIP_SIGNATURES["ssd_copy_blocks"] = (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p])
HIP_SIGNATURES["ssd_fill_blocks"] = (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p])
HIP_SIGNATURES["ssd_runner_stats"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p])
HIP_SIGNATURES["ssd_gru_seq_fwd_parts"] = (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p])
HIP_SIGNATURES["ssd_gru_seq_bwd_parts"] = (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_void_p, C.c_int32] + [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_void_p])
HIP_SIGNATURES["ssd_bias_bmm_fwd"] = (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p])
HIP_SIGNATURES["ssd_bias_bmm_bwd"] = (C.c_int, [C.c_void_p] * 7 + [C.c_int32] * 4 + [C.c_void_p])
HIP_SIGNATURES["ssd_bias_bmm_leaky_fwd"] = (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p])
HIP_SIGNATURES["ssd_bias_bmm_leaky_bwd"] = (C.c_int, [C.c_void_p] * 8 + [C.c_int32] * 4 + [C.c_void_p])
HIP_SIGNATURES["ssd_bias_bmm2_fwd"] = (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 7 + [C.c_void_p])
HIP_SIGNATURES["ssd_bias_bmm2_bwd_w"] = (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 7 + [C.c_void_p])
HIP_SIGNATURES["ssd_bias_bmm_bwd_x"] = (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_int64, C.c_void_p])
HIP_SIGNATURES["ssd_dueling_head_fwd"] = (C.c_int, [C.c_void_p] * 2 + [C.c_int32] * 5 + [C.c_void_p])
HIP_SIGNATURES["ssd_dueling_head_bwd"] = (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 5 + [C.c_void_p])
HIP_SIGNATURES["ssd_unroll_other"] = (C.c_int, [C.c_void_p] * 6 + [C.c_float] + [C.c_int32] * 4 + [C.c_void_p] * 3)
HIP_SIGNATURES["ssd_bmm_reserve_scratch"] = (C.c_int, [C.c_void_p])
HIP_SIGNATURES["ssd_set_learner_precision"] = (C.c_int, [C.c_int32])
HIP_SIGNATURES["ssd_learner_precision"] = (C.c_int, [])
HIP_SIGNATURES["ssd_conv_wgrad_partial_rows"] = (C.c_int, [C.c_int32])
HIP_SIGNATURES["ssd_conv_wgrad_codes"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p])
HIP_SIGNATURES["ssd_policy_head_plan"] = (C.c_int, [C.c_int32] * 3 + [C.POINTER(C.c_int32)] * 3)
HIP_SIGNATURES["ssd_policy_encode"] = (C.c_int, [C.POINTER(SsdPolicyEncodeArgs), C.c_void_p])
HIP_SIGNATURES["ssd_policy_head_inc_encode"] = (C.c_int, [C.POINTER(SsdPolicyHead), C.POINTER(SsdPolicyEncodeArgs), C.c_void_p])
HIP_SIGNATURES["ssd_policy_pack_encoder_lut"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p])
HIP_SIGNATURES["ssd_policy_pack_encoder"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p])
HIP_SIGNATURES["ssd_numeric_status"] = (C.c_int, [C.POINTER(C.c_int32)])
ERRBIT_F16_RANGE = 32
HIP_SIGNATURES["ssd_column_sums"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p])
COLSUM_CHUNK = 64
HIP_SIGNATURES["ssd_td_sim_loss"] = (C.c_int, [C.POINTER(SsdTdLossArgs), C.c_int32, C.c_void_p])
HIP_SIGNATURES["ssd_policy_pack_head"] = (C.c_int, [C.POINTER(SsdPolicyHeadParams), C.c_int32, C.c_void_p, C.c_void_p])
HIP_SIGNATURES["ssd_conv_leaky"] = (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                             C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p])
HIP_SIGNATURES["ssd_store_step_launch"] = (C.c_int, [C.POINTER(SsdStoreStep), C.c_void_p])
HIP_SIGNATURES["ssd_gru_gates"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p])
HIP_SIGNATURES["ssd_gru_gates_fwd"] = (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_int32, C.c_void_p])
HIP_SIGNATURES["ssd_gru_gates_bwd"] = (C.c_int, [C.c_void_p] * 7 + [C.c_int32, C.c_int32, C.c_void_p])
HIP_SIGNATURES["ssd_dueling_pick"] = (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32,
                                               C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p])
HIP_SIGNATURES["ssd_poll_error"] = (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)])
CPU_SIGNATURES = _sigs("ssd_cpu_", False)

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# SSD_HIP_LIB_PATH: diagnostics only (tools/stamps.py loads a -DSSD_STAMPS build of the same sources)
HIP_LIB_PATH = os.environ.get("SSD_HIP_LIB_PATH", os.path.join(_PKG_DIR, "libssd_hip.so"))
_lib = None


def bind(lib, signatures):
    missing = []
    for name, (res, args) in signatures.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise ImportError("native library is missing symbols: " + ", ".join(missing))
    return lib


def load_library():
    """Load homophily_marl_amd/libssd_hip.so (built by __graft_entry__.build()).  Fails loudly when absent."""
    global _lib
    if _lib is None:
        # torch first: the PyTorch-ROCm wheel bundles its own libamdhip64.so (soname libamdhip64.so.7).  Loaded
        # first, it satisfies this library's NEEDED entry, so both share ONE HIP runtime (device pointers and
        # streams are only meaningful within one runtime).  Loaded second, /opt/rocm's copy would already be in
        # the process and torch would bring a second runtime.
        import torch  # noqa: F401
        if not os.path.exists(HIP_LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % HIP_LIB_PATH)
        lib = C.CDLL(HIP_LIB_PATH)
        bind(lib, HIP_SIGNATURES)
        ver = lib.ssd_abi_version()
        if ver != ABI_VERSION:
            raise ImportError("libssd_hip.so ABI version %d != %d" % (ver, ABI_VERSION))
        _lib = lib
    return _lib


class SsdError(RuntimeError):
    pass


def check(lib, rc, err_fn="ssd_last_error"):
    if rc != 0:
        msg = getattr(lib, err_fn)()
        raise SsdError("ssd error %d: %s" % (rc, msg.decode() if msg else "?"))
