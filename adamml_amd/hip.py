"""ctypes binding of libadamml_hip.so (include/adamml_hip.h).

The product path has NO fallback: if the shared library is missing or a launch fails the
caller gets a RuntimeError -- never a silent CPU / eager path."""
import ctypes
import os
from ctypes import c_void_p, c_int, c_int32, c_int64, c_float, c_double, c_size_t, c_char_p, POINTER, Structure, byref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADAMML_HIP_LIB: developer aid for A/B runs of two builds of the library in one session (tools/bench_*.py)
LIB_PATH = os.environ.get("ADAMML_HIP_LIB") or os.path.join(_HERE, "libadamml_hip.so")

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
STAT_SLOTS = 32          # ADAMML_STAT_SLOTS in include/adamml_hip.h


class ConvDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("N", "H", "W", "Cin", "OH", "OW", "Cout", "KH", "KW", "stride", "pad", "up", "act",
                                      "accumulate", "groups", "in_gstride")]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if self.groups < 1:
            self.groups = 1


_P, _I, _F, _D, _Z, _L = c_void_p, c_int, c_float, c_double, c_size_t, c_int64
_DESC = POINTER(ConvDesc)

# name -> argtypes (every entry point declared in include/adamml_hip.h, stream last)
SIGNATURES = {
    "adamml_conv_fwd": [_DESC, _P, _P, _P, _P, _P, _P, _P],
    "adamml_conv_bwd_data": [_DESC, _P, _P, _P, _I, _P],
    "adamml_conv_fwd_bn_add": [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "adamml_conv_fwd_bn_add_next": [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    "adamml_conv_fwd_bn_add_tpool": [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "adamml_temporal_pool_bwd_code": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "adamml_temporal_pool_bwd_code_prod": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _P],
    "adamml_copy2d": [_P, _Z, _P, _Z, _Z, _Z, _P],
    "adamml_gram_stats": [_P, _P, _P, _P, _I, _I, _I, _P],
    "adamml_gram_colsum": [_P, _P, _P, _I, _I, _P, _P, _Z, _I, _I, _P, _Z, _P],
    "adamml_conv_bwd_data_bn": [_DESC, _P, _P, _P, _P, _P, _I, _P, _P],
    "adamml_bn_bwd_affine": [_P, _P, _P, _I, _I, _P],
    "adamml_conv_bwd_data_dual": [_DESC, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P],
    "adamml_conv_bwd_weight_grouped": [_DESC, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _Z, _P],
    "adamml_lazy_colsum": [_P, _P, _P, _I, _I, _P, _Z, _I, _I, _P],
    "adamml_alg_sumfix": [_P, _P, _P, _P, _I, _I, _I, _P],
    "adamml_alg_pack": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "adamml_alg_wgrad_combine": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "adamml_conv_bwd_data_alg": [_DESC, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P],
    "adamml_conv_bwd_data_res": [_DESC, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P],
    "adamml_conv_bwd_data_res_prod": [_DESC, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P, _Z, _P],
    "adamml_bn_act_add_mask": [_P, _P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _Z, _I, _I, _P],
    "adamml_residual_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _P],
    "adamml_conv_bwd_weight": [_DESC, _P, _P, _P, _P, _P, _I, _P, _Z, _P],
    "adamml_pack_conv_weight": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "adamml_pack_conv_weights_batched": [_P, _I, _L, _P],
    "adamml_pack_stem_weight": [_P, _P, _I, _I, _P],
    "adamml_conv_stem_fwd": [_DESC, _P, _P, _P, _P, _P],
    "adamml_conv_stem_bwd_weight": [_DESC, _P, _P, _P, _I, _P, _Z, _P],
    "adamml_dwconv_fwd": [_DESC, _P, _P, _P, _P, _P, _P, _P],
    "adamml_conv_stem1_fwd": [_DESC, _P, _Z, _Z, _P, _P, _P, _P],
    "adamml_conv_stem1_bwd_weight": [_DESC, _P, _P, _Z, _Z, _P, _P, _Z, _P],
    "adamml_dwconv_bwd_data": [_DESC, _P, _P, _P, _I, _P],
    "adamml_dwconv_bwd_data_bn": [_DESC, _P, _P, _P, _P, _P, _I, _P, _P],
    "adamml_dwconv_bwd_weight": [_DESC, _P, _P, _P, _P, _P, _P, _Z, _P],
    "adamml_dwconv_bwd_fused": [_DESC, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _Z, _P],
    "adamml_stats_collapse": [_P, _P, _I, _I, _P],
    "adamml_bn_finalize": [_P, _I, _I, _D, _P, _P, _P, _P, _F, _F, _P, _I, _P],
    "adamml_bn_eval_affine": [_P, _P, _P, _P, _F, _P, _P, _I, _P],
    "adamml_bn_act_add": [_P, _P, _P, _I, _I, _P, _P, _P, _I, _P, _Z, _I, _I, _P],
    "adamml_act_bwd_from_output": [_P, _P, _I, _P, _Z, _P],
    "adamml_bn_bwd_reduce": [_P, _P, _P, _I, _P, _Z, _I, _I, _P],
    "adamml_bn_bwd_finalize": [_P, _I, _I, _D, _P, _P, _P, _P, _P, _I, _F, _P],
    "adamml_bn_bwd_finalize_affine": [_P, _I, _I, _D, _P, _P, _P, _P, _P, _P, _I, _F, _P],
    "adamml_bn_bwd_apply": [_P, _P, _P, _I, _P, _P, _Z, _I, _I, _P],
    "adamml_maxpool2d_fwd": [_P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "adamml_maxpool2d_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "adamml_maxpool2d_bwd_bn_reduce": [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "adamml_maxpool2d_bwd_bn_apply": [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "adamml_temporal_pool_fwd": [_P, _P, _P, _I, _I, _P, _I, _I, _Z, _I, _I, _I, _P],
    "adamml_temporal_pool_bwd": [_P, _P, _P, _P, _I, _I, _P, _I, _I, _Z, _I, _I, _I, _P],
    "adamml_temporal_pool_bwd_res": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "adamml_gap_fwd": [_P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _P],
    "adamml_gap_bwd": [_P, _P, _I, _I, _I, _P],
    "adamml_head_fwd": [_P, _P, _P, _I, _I, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "adamml_head_bwd": [_P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "adamml_colsum_f32": [_P, _P, _I, _I, _I, _P],
    "adamml_clip_to_nhwc": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "adamml_clip_u8_to_nhwc": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P],
    "adamml_clip_u8_rgbdiff_to_nhwc": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "adamml_gemm_f32": [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _I, _I, _P],
    "adamml_sgd_step": [_P, _P, _P, _Z, _F, _F, _F, _I, _I, _P],
    "adamml_adam_step": [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _I, _P],
    "adamml_policy_head_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "adamml_policy_head_bwd": [_P, _P, _P, _I, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "adamml_gumbel_gate_fwd": [_P, _P, _F, _P, _P, _I, _P],
    "adamml_gumbel_gate_bwd": [_P, _P, _F, _P, _I, _P],
    "adamml_fusion_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "adamml_fusion_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
}

_lib = None


def load():
    """Load libadamml_hip.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("adamml_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                           "there is no CPU fallback for the HIP hot path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argt in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.argtypes = argt
        fn.restype = c_int
    lib.adamml_conv_bwd_weight_workspace.argtypes = [_DESC, _I]
    lib.adamml_conv_bwd_weight_workspace.restype = c_size_t
    lib.adamml_conv_stem_bwd_weight_workspace.argtypes = [_DESC]
    lib.adamml_conv_stem_bwd_weight_workspace.restype = c_size_t
    lib.adamml_dwconv_bwd_weight_workspace.argtypes = [_DESC]
    lib.adamml_dwconv_bwd_weight_workspace.restype = c_size_t
    lib.adamml_conv_stem1_bwd_weight_workspace.argtypes = [_DESC]
    lib.adamml_conv_stem1_bwd_weight_workspace.restype = c_size_t
    lib.adamml_conv_stem1_supported.argtypes = [_DESC]
    lib.adamml_conv_stem1_supported.restype = c_int
    lib.adamml_gram_colsum_workspace.argtypes = [_Z, _I, _I]
    lib.adamml_gram_colsum_workspace.restype = c_size_t
    lib.adamml_gram_colsum_supported.argtypes = [_I]
    lib.adamml_gram_colsum_supported.restype = c_int
    lib.adamml_conv_bwd_data_res_prod_workspace.argtypes = [_DESC]
    lib.adamml_conv_bwd_data_res_prod_workspace.restype = c_size_t
    lib.adamml_conv_bwd_data_res_prod_supported.argtypes = [_DESC, _I]
    lib.adamml_conv_bwd_data_res_prod_supported.restype = c_int
    lib.adamml_conv_bwd_data_res_prod_streams.argtypes = [_DESC, _I]
    lib.adamml_conv_bwd_data_res_prod_streams.restype = c_int
    lib.adamml_conv_fwd_bn_add_streams.argtypes = [_DESC]
    lib.adamml_conv_fwd_bn_add_streams.restype = c_int
    lib.adamml_conv_bwd_data_res_streams.argtypes = [_DESC]
    lib.adamml_conv_bwd_data_res_streams.restype = c_int
    lib.adamml_conv_fwd_bn_add_tpool_streams.argtypes = [_DESC, _I]
    lib.adamml_conv_fwd_bn_add_tpool_streams.restype = c_int
    lib.adamml_conv_fused_input_supported.argtypes = [_DESC]
    lib.adamml_conv_fused_input_supported.restype = c_int
    lib.adamml_conv1x1_narrow_supported.argtypes = [_DESC, c_int]
    lib.adamml_conv1x1_narrow_supported.restype = c_int
    lib.adamml_conv1x1_wide_supported.argtypes = [_DESC, c_int]
    lib.adamml_conv1x1_wide_supported.restype = c_int
    lib.adamml_conv_fwd_bn_add_supported.argtypes = [_DESC]
    lib.adamml_conv_fwd_bn_add_supported.restype = c_int
    lib.adamml_conv_fwd_bn_add_next_supported.argtypes = [_DESC, c_int]
    lib.adamml_conv_fwd_bn_add_next_supported.restype = c_int
    lib.adamml_conv_fwd_bn_add_tpool_supported.argtypes = [_DESC, _I, _I, _I]
    lib.adamml_conv_fwd_bn_add_tpool_supported.restype = c_int
    lib.adamml_conv_bwd_data_dual_supported.argtypes = [_DESC]
    lib.adamml_conv_bwd_data_dual_supported.restype = c_int
    lib.adamml_conv_bwd_data_res_supported.argtypes = [_DESC]
    lib.adamml_conv_bwd_data_res_supported.restype = c_int
    lib.adamml_temporal_pool_bwd_res_supported.argtypes = [_I, _I, _I]
    lib.adamml_temporal_pool_bwd_res_supported.restype = c_int
    lib.adamml_dwconv_bwd_data_bn_supported.argtypes = [_DESC]
    lib.adamml_dwconv_bwd_data_bn_supported.restype = c_int
    lib.adamml_temporal_pool_bwd_code_prod_supported.argtypes = [c_int, c_int, c_int]
    lib.adamml_temporal_pool_bwd_code_prod_supported.restype = c_int
    lib.adamml_temporal_pool_bwd_code_prod_workspace.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int]
    lib.adamml_temporal_pool_bwd_code_prod_workspace.restype = c_size_t
    lib.adamml_dwconv_bwd_fused_supported.argtypes = [_DESC]
    lib.adamml_dwconv_bwd_fused_supported.restype = c_int
    lib.adamml_dwconv_bwd_fused_workspace.argtypes = [_DESC]
    lib.adamml_dwconv_bwd_fused_workspace.restype = c_size_t
    lib.adamml_conv_stem_supported.argtypes = [_DESC]
    lib.adamml_conv_stem_supported.restype = c_int
    lib.adamml_plan_run.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int]
    lib.adamml_plan_run.restype = c_int
    lib.adamml_plan_events_create.argtypes = [c_void_p, c_int]
    lib.adamml_plan_events_create.restype = c_int
    lib.adamml_plan_events_destroy.argtypes = [c_void_p, c_int]
    lib.adamml_plan_events_destroy.restype = c_int
    lib.adamml_plan_num_entry_points.restype = c_int
    lib.adamml_pack_block_elems.restype = c_int
    lib.adamml_version.restype = c_int
    lib.adamml_last_error_string.restype = c_char_p
    lib.adamml_set_deterministic.argtypes = [_I]
    lib.adamml_set_deterministic.restype = c_int
    lib.adamml_get_deterministic.restype = c_int
    _lib = lib
    return lib


def set_deterministic(on=True):
    """Every per-channel statistic is order-fixed and exact (csrc/common.h: reproducible reductions): two runs of the same step are
    bit-identical, always.  Kept for callers of earlier versions: True and False are both accepted and change nothing (False warns
    once: a `try: set_deterministic(True) ... finally: set_deterministic(False)` caller keeps working); `deterministic()` stays True.
    The ADAMML_DETERMINISTIC environment variable of rounds 2-3 is no longer read."""
    lib = _lib if _lib is not None else load()
    rc = lib.adamml_set_deterministic(1 if on else 0)
    if rc != 0:
        raise RuntimeError("adamml_set_deterministic failed (%d): %s" % (rc, lib.adamml_last_error_string().decode()))


def deterministic():
    return bool((_lib if _lib is not None else load()).adamml_get_deterministic())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw accessor: no Stream object per launch -- the Python-side
    cost of ~1000 launches per step is what bounds the step at the per-GPU batch of the reference recipe, B = 9)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


recorder = None          # plan.Recorder listening to the launches of one backbone call (plan.py), else None


def ptr(t):
    if t is None:
        return None
    if recorder is not None:
        recorder.keep.append(t)          # the plan owns every tensor whose address one of its records holds
    return t.data_ptr()


class LaunchProfiler:
    """Optional per-launch HIP-event timing (bench.py roofline leg): every C-ABI launch is bracketed by events on the
    stream it is enqueued on; `meta` = (algorithmic flops, algorithmic HBM bytes[, device kernel]) supplied by the caller:
    launches are grouped by the device kernel when the caller names it (one C-ABI entry point may dispatch to several
    kernels), by the entry point otherwise."""

    def __init__(self):
        self.records = []

    def summary(self, by_role=False):
        """by_role=False: launches grouped by device kernel (entry point where the caller does not name one).
        by_role=True: the MFMA launches grouped by their roofline role (meta[3], runtime.R_*); launches without a role are left out."""
        torch.cuda.synchronize()
        agg = {}
        for name, s, e, meta in self.records:
            if by_role:
                if len(meta) < 4 or not meta[3]:
                    continue
                name = meta[3]
            elif len(meta) > 2 and meta[2]:
                name = meta[2]
            a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            a["launches"] += 1
            a["ms"] += s.elapsed_time(e)
            a["flops"] += meta[0]
            a["bytes"] += meta[1]
        return agg


profiler = None
next_meta = (0.0, 0.0)


_fns = {}


def call(name, *args):
    global next_meta
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    if profiler is not None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args, _stream())
        e.record()
        profiler.records.append((name, s, e, next_meta))
        next_meta = (0.0, 0.0)
    else:
        st = _stream()
        rc = fn(*args, st)
        if recorder is not None:
            recorder.call(name, args, st)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, load().adamml_last_error_string().decode()))


_wgrad_ws = {}


def wgrad_workspace(desc, cin_true, device, depthwise=False, stem=False):
    """Persistent per-device scratch for the split weight-gradient partial tiles (grown on demand)."""
    if stem:
        need = load().adamml_conv_stem_bwd_weight_workspace(ctypes.byref(desc))
    elif depthwise:
        need = load().adamml_dwconv_bwd_weight_workspace(ctypes.byref(desc))
    else:
        need = load().adamml_conv_bwd_weight_workspace(ctypes.byref(desc), cin_true)
    key = (device, _stream())      # one scratch per stream: backbones run concurrently
    buf = _wgrad_ws.get(key)
    if buf is None or buf.numel() * 4 < need:
        buf = torch.empty(max(need // 4 + 1, 1 << 20), dtype=torch.float32, device=device)
        _wgrad_ws[key] = buf
    return buf


def scratch(nbytes, device):
    """The per-stream scratch buffer of wgrad_workspace(), grown to at least nbytes."""
    key = (device, _stream())
    buf = _wgrad_ws.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty(max(nbytes // 4 + 1, 1 << 20), dtype=torch.float32, device=device)
        _wgrad_ws[key] = buf
    return buf


def ptr_array(tensors):
    """HOST array of device pointers (const float* const* arguments); None entries become NULL."""
    return (c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("adamml_amd: tensors must live on an MI355X (got device %s); there is no CPU path" % t.device)
