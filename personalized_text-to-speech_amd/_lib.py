"""ctypes binding of libvitsmi.so (the C ABI declared in include/vitsmi.h).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C csrc``) and is the ONLY
implementation of the hot ops: there is no CPU or eager fallback behind it.  ``lib()`` raises
``RuntimeError`` if the shared object is missing or an expected symbol is not exported.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvitsmi.so")
ABI_VERSION = 16

c_int = ctypes.c_int
c_void_p = ctypes.c_void_p
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t

# name -> (restype, argtypes); kept in lock-step with include/vitsmi.h (tests/test_abi.py checks
# that every function declared in the header is listed here and exported by the .so).
SIGNATURES = {
    "vits_abi_version": (c_int, []),
    "vits_last_error": (ctypes.c_char_p, []),
    "vits_mas_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vits_relsoftmax": (c_int, [c_int] + [c_void_p] * 7 + [c_int] * 4 + [c_float, c_void_p]),
    "vits_relsoftmax_bwd": (c_int, [c_int] + [c_void_p] * 7 + [c_int] * 4 + [c_float, c_void_p]),
    "vits_rq_spline": (c_int, [c_int, c_void_p, c_void_p, c_int, c_float, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "vits_rq_spline_bwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_float, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vits_rowops_workspace": (c_size_t, [c_int, c_int, c_int]),
    "vits_ln_act_cl": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p]),
    "vits_ln_act_cl_bwd": (c_int, [c_int] + [c_void_p] * 8 + [c_size_t, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "vits_dwconv_cl": (c_int, [c_int] + [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "vits_dwconv_cl_bwd": (c_int, [c_int] + [c_void_p] * 8 + [c_size_t] + [c_int] * 6 + [c_void_p]),
    "vits_weight_prep": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vits_weight_prep_transpose": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "vits_weight_prep_bwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vits_convt_fold_cl": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "vits_convt_unfold_cl": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "vits_reduce_workspace": (c_size_t, [c_int]),
    "vits_absdiff_sum": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_float, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "vits_absdiff_bwd": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "vits_segsum_f32": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vits_colsum_workspace": (c_size_t, [c_int, c_int, c_int]),
    "vits_colsum": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vits_lrelu_mask_bwd": (c_int, [c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vits_conv1d_cl_wgrad_workspace": (c_size_t, [c_int] * 5),
    "vits_conv1d_cl_wgrad": (c_int, [c_void_p, c_void_p]),
    "vits_conv1d_cl_wgrad_deferred": (c_int, [c_void_p, c_void_p, c_void_p]),
    "vits_wgrad_reduce_pending": (c_int, [c_void_p, c_int, c_void_p]),
    "vits_conv1d_cl_wgrad_batch_plan": (c_int, [c_void_p, c_int, c_void_p]),
    "vits_conv1d_cl_wgrad_batch": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "vits_mas_f32_cpu": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vits_conv1d_cl": (c_int, [c_void_p, c_void_p]),
    "vits_conv1d_cl_multi": (c_int, [c_void_p, c_int, c_void_p]),
    "vits_wn_layer_fwd": (c_int, [c_void_p, c_void_p]),
    "vits_wn_layer_bwd": (c_int, [c_void_p, c_void_p]),
    "vits_wn_pack_bytes": (c_size_t, [c_int] * 6),
    "vits_wn_pack": (c_int, [c_void_p, c_int, c_void_p]),
    "vits_grouped_conv_fwd": (c_int, [c_int] + [c_void_p] * 4 + [c_int] * 8 + [c_float, c_void_p]),
    "vits_grouped_conv_dgrad": (c_int, [c_int] + [c_void_p] * 5 + [c_int] * 8 + [c_float, c_void_p]),
    "vits_grouped_conv_wgrad_workspace": (c_size_t, [c_int] * 5),
    "vits_grouped_conv_wgrad": (c_int, [c_int] + [c_void_p] * 5 + [c_size_t] + [c_int] * 9 + [c_void_p, c_void_p]),
    "vits_stft_mel_fwd": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_float, c_void_p]),
    "vits_stft_mel_bwd": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float, c_void_p]),
    "vits_disc_first_rows": (c_int, [c_int] * 5),
    "vits_disc_first_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_float, c_void_p]),
    "vits_disc_first_wgrad_workspace": (c_size_t, [c_int] * 7),
    "vits_disc_first_wgrad": (c_int, [c_int] + [c_void_p] * 5 + [c_size_t] + [c_int] * 8 + [c_void_p]),
    "vits_disc_first_dgrad": (c_int, [c_int] + [c_void_p] * 3 + [c_int] * 9 + [c_void_p]),
    "vits_disc_post_fwd": (c_int, [c_int] + [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "vits_disc_post_dgrad": (c_int, [c_int] + [c_void_p] * 5 + [c_int] * 6 + [c_float, c_void_p]),
    "vits_disc_post_wgrad_workspace": (c_size_t, [c_int] * 4),
    "vits_disc_post_wgrad": (c_int, [c_int] + [c_void_p] * 5 + [c_size_t] + [c_int] * 6 + [c_void_p]),
    "vits_adamw_blocks": (c_size_t, [c_void_p, c_int]),
    "vits_adamw": (c_int, [c_void_p] * 4 + [c_int, c_void_p] + [ctypes.c_double] * 4 + [c_void_p, c_size_t, c_void_p]),
    "vits_gradnorm_final": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_void_p]),
    "vits_coupling_tail": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "vits_coupling_tail_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "vits_flow_affine": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "vits_flow_affine_bwd": (c_int, [c_void_p] * 8 + [c_int] * 4 + [c_void_p]),
    "vits_flow_dequant_log": (c_int, [c_void_p] * 6 + [c_int, c_int, c_void_p]),
    "vits_flow_dequant_log_bwd": (c_int, [c_void_p] * 7 + [c_int, c_int, c_void_p]),
    "vits_flow_front": (c_int, [c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vits_flow_front_workspace": (c_size_t, [c_int, c_int]),
    "vits_flow_front_bwd": (c_int, [c_int, c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_size_t, c_int, c_int, c_void_p]),
    "vits_flow_spline": (c_int, [c_int, c_void_p, c_void_p, c_int, c_float, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "vits_flow_spline_bwd": (c_int, [c_int, c_void_p, c_void_p, c_int, c_float, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p, c_int, c_void_p]),
    "vits_feature_l1_workspace": (c_size_t, [c_int]),
    "vits_feature_l1": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vits_feature_l1_bwd": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "vits_lsgan_workspace": (c_size_t, [c_int]),
    "vits_lsgan_loss": (c_int, [c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vits_lsgan_loss_bwd": (c_int, [c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vits_neg_cent": (c_int, [c_int, c_void_p, ctypes.c_long, c_int, c_void_p, c_void_p, ctypes.c_long, c_void_p] + [c_int] * 4 + [c_void_p]),
    "vits_slice_segments": (c_int, [c_int, c_int, c_void_p, c_void_p, ctypes.c_long, c_void_p] + [c_int] * 5 + [c_void_p]),
    "vits_generate_path": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
}

class ConvDesc(ctypes.Structure):
    """vits_conv_desc of include/vitsmi.h"""
    _fields_ = [(n, ctypes.c_int32) for n in ("dtype", "b", "t", "c_in", "c_out", "k", "dil", "pad", "stride", "flags",
                                              "ldx", "ldy", "ldy2", "gate_h", "ldw", "in_div", "t_out_override", "groups")] + \
               [("w_batch_stride", ctypes.c_int64)] + \
               [(n, c_float) for n in ("in_slope", "mg_slope", "out_scale", "out_slope")] + \
               [(n, c_void_p) for n in ("x", "w", "bias", "bias_b", "res", "mg_src", "y", "y2", "lengths")]


class WnLayerDesc(ctypes.Structure):
    """vits_wn_layer_desc of include/vitsmi.h"""
    _fields_ = [(n, ctypes.c_int32) for n in ("dtype", "b", "t", "h", "k", "dil", "last", "accumulate", "ldx", "ldh", "ldskip", "ldacts", "ldpre")] + \
               [(n, c_void_p) for n in ("x", "w_in", "b_in", "cond", "w_rs", "b_rs", "pre", "acts", "h_out", "skip", "lengths")]


class WnPackSeg(ctypes.Structure):
    """vits_wn_pack_seg of include/vitsmi.h"""
    _fields_ = [("src", c_void_p), ("dst", c_void_p)] + [(n, ctypes.c_int32) for n in ("mode", "h", "rows", "rowbytes", "taps", "spt")]


class WnLayerBwdDesc(ctypes.Structure):
    """vits_wn_layer_bwd_desc of include/vitsmi.h"""
    _fields_ = [(n, ctypes.c_int32) for n in ("dtype", "b", "t", "h", "k", "dil", "last", "ld_dh", "ld_do", "ldpre", "lddpre", "ldout")] + \
               [(n, c_void_p) for n in ("d_h", "d_o", "pre", "w_rs_t", "w_in_t", "d_pre", "d_h_out", "lengths")]


class PrepEntry(ctypes.Structure):
    """vits_prep_entry of include/vitsmi.h"""
    _fields_ = [("v", c_void_p), ("g", c_void_p), ("off", ctypes.c_int64), ("off_dv", ctypes.c_int64), ("off_dg", ctypes.c_int64)] + \
               [(n, ctypes.c_int32) for n in ("layout", "c_out", "c_in", "k", "c_out_p", "c_in_p", "row_lo", "n_rows", "row0", "groups")]


class FeatItem(ctypes.Structure):
    """vits_feat_item of include/vitsmi.h"""
    _fields_ = [("h", c_void_p), ("dh", c_void_p), ("n", c_size_t), ("scale", c_float)]


class LsganItem(ctypes.Structure):
    """vits_lsgan_item of include/vitsmi.h"""
    _fields_ = [("y8", c_void_p), ("dy8", c_void_p), ("J", c_int), ("R", c_int)]


class AdamwEntry(ctypes.Structure):
    """vits_adamw_entry of include/vitsmi.h"""
    _fields_ = [("g", c_void_p), ("offset", ctypes.c_uint64), ("n", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class WgradPending(ctypes.Structure):
    """vits_wgrad_pending of include/vitsmi.h"""
    _fields_ = [("partial", c_void_p), ("dw", c_void_p), ("dbias", c_void_p), ("n", c_size_t), ("nb", c_size_t), ("slab", c_size_t),
                ("splits", ctypes.c_int32), ("accumulate", ctypes.c_int32)]


class WgradDesc(ctypes.Structure):
    """vits_wgrad_desc of include/vitsmi.h"""
    _fields_ = [(n, ctypes.c_int32) for n in ("dtype", "b", "t", "c_in", "c_out", "k", "dil", "pad", "stride", "flags",
                                              "ldx", "lddy")] + \
               [("in_slope", c_float), ("groups", ctypes.c_int32)] + \
               [("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t),
                ("lengths", c_void_p), ("dbias", c_void_p), ("counters", c_void_p), ("counters_len", c_size_t)]


_lib = None


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into lib/libvitsmi.so (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libvitsmi.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. "
                "The HIP kernels have no fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise RuntimeError(f"libvitsmi.so does not export {name}: rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        got = handle.vits_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"libvitsmi.so ABI {got} != binding ABI {ABI_VERSION}: rebuild it")
        _lib = handle
    return _lib


class VitsKernelError(RuntimeError):
    pass


_CODES = {-1: "VITS_E_BADARG", -2: "VITS_E_UNSUPPORTED", -3: "VITS_E_LAUNCH"}
E_UNSUPPORTED = -2


def check(rc, what):
    if rc != 0:
        detail = lib().vits_last_error().decode() if rc == -3 else ""
        raise VitsKernelError(f"{what}: {_CODES.get(rc, rc)} {detail}")


class KernelTimer:
    """Optional HIP-event timing of the C-ABI launches, per kernel name (bench.py's roofline leg).
    Events are recorded on the stream the kernel is launched on (torch's current stream)."""

    def __init__(self):
        self.enabled = False
        self.detail = os.environ.get("VITS_TIMER_DETAIL") == "1"     # key events by launch shape too (tools/time_shapes.py)
        self.events = {}          # name -> [(start, end, units)]

    def start(self, name):
        if not self.enabled:
            return None
        import torch
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0

    def stop(self, name, e0, units, shape=None):
        if e0 is None:
            return
        if self.detail and shape is not None:
            name = f"{name} {shape}"
        import torch
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.events.setdefault(name, []).append((e0, e1, units))

    def summary(self):
        """name -> dict(calls, total_ms, avg_ms, units_per_call); `units` may be a number or a tuple
        (summed component-wise).  Call after torch.cuda.synchronize()."""
        out = {}
        for name, ev in self.events.items():
            ms = [a.elapsed_time(b) for a, b, _ in ev]
            us = [u if isinstance(u, tuple) else (u,) for _, _, u in ev]
            tot = tuple(sum(col) for col in zip(*us))
            out[name] = dict(calls=len(ev), total_ms=sum(ms), avg_ms=sum(ms) / len(ms),
                             units_total=tot, units_per_call=tuple(t / len(ev) for t in tot))
        return out

    def reset(self):
        self.events = {}

    @staticmethod
    def pair_overhead_ms(n=200):
        """What an event pair with NOTHING between its two records measures on the current stream (median of n): the part of
        every sample above that is not kernel time (rocprofv3 times the dispatch itself and does not see it)."""
        import torch
        pairs = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
            pairs.append((a, b))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in pairs)
        return ms[len(ms) // 2]


timer = KernelTimer()


def stream_ptr():
    """The hipStream_t of torch's current stream, as an integer for the C ABI."""
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("this op runs only through the HIP kernels of libvitsmi.so: tensors must be on the GPU")
