"""ctypes binding of libtem_hip.so (C-ABI declared in include/tem_hip.h).

The reference has no native code; this is the binding a torch-em maintainer would add
(see INTEGRATION.md).  The product path has NO CPU fallback: if the shared library is
missing, every op raises.  `build()` compiles it in-tree with hipcc for gfx950.
"""
import ctypes
import threading
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# TEM_LIB: developer override for A/B builds of the same C-ABI (scripts/ab_lib.sh); the product path is the in-tree file
LIB_PATH = os.environ.get("TEM_LIB") or os.path.join(_HERE, "lib", "libtem_hip.so")
CSRC = os.path.join(_HERE, "csrc")

c_f32p = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_float = ctypes.c_float
c_double = ctypes.c_double
c_vp = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol of include/tem_hip.h
SIGNATURES = {
    "tem_last_error": (ctypes.c_char_p, []),
    "tem_version": (c_int, []),
    "tem_device_cus": (c_int, []),
    "tem_set_option": (c_int, [ctypes.c_char_p, c_i64]),
    "tem_get_option": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_i64)]),
    "tem_conv_packed_size": (c_i64, [c_int] * 5),
    "tem_conv_pack_weights": (c_int, [c_vp, c_vp] + [c_int] * 7 + [c_vp]),
    "tem_conv_pack_weights_batch": (c_int, [c_vp, c_int, c_i64, c_vp]),
    "tem_conv_pack_weights_tiles": (c_int, [c_vp, c_int, c_i64, c_vp]),
    "tem_conv_unpack_wgrad": (c_int, [c_vp, c_vp] + [c_int] * 5 + [c_vp]),
    "tem_conv3d_fwd_ws": (c_i64, [c_int] * 10),
    "tem_conv3d_fwd": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64]
                       + [c_int] * 11 + [c_vp]),
    "tem_conv3d_fwd_stat_blocks": (c_i64, [c_int] * 10),
    "tem_conv3d_fwd_kernel": (c_int, [c_int] * 10),
    "tem_conv3d_fwd_stats": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64]
                             + [c_int] * 11 + [c_vp, c_i64, c_vp]),
    "tem_conv3d_wgrad_ws": (c_i64, [c_int] * 10),
    "tem_conv3d_wgrad": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64] + [c_int] * 11 + [c_vp]),
    "tem_conv1x1_out_bwd_ok": (c_int, [c_int] * 2),
    "tem_conv1x1_out_bwd_ws": (c_i64, [c_int] * 2),
    "tem_conv1x1_out_bwd": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp]),
    "tem_conv3d_wgrad_gmax_ok": (c_int, [c_int] * 10),
    "tem_conv3d_wgrad_gmax": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64]
                              + [c_int] * 10 + [c_vp]),
    "tem_conv3d_wgrad_gscaled_ok": (c_int, [c_int] * 9),
    "tem_conv3d_wgrad_cs_ok": (c_int, [c_int] * 10 + [c_i64]),
    "tem_conv3d_wgrad_gscaled": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64]
                                 + [c_int] * 9 + [c_vp]),
    "tem_absmax": (c_int, [c_vp, c_i64, c_int, c_i64, c_vp, c_vp]),
    "tem_norm_bwd_from_partials": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_i64,
                                           c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    # (..., in_amax, ref_coef, TemByproducts*, stream) / (..., g_amax_in, g_amax_out, ws, ws_bytes, dims, use_mfma, TemByproducts*, stream)
    "tem_conv3d_fwd_ex": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64] + [c_int] * 11
                          + [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "tem_conv3d_wgrad_ex": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64]
                            + [c_int] * 10 + [c_i64, c_vp, c_vp]),
    "tem_conv3d_fwd_gscaled": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64] + [c_int] * 9 + [c_vp]),
    "tem_conv3d_fwd_refnorm": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64] + [c_int] * 10 + [c_vp]),
    "tem_norm_ws": (c_i64, [c_int, c_i64, c_int]),
    "tem_conv3d_wgrad_gnorm_ok": (c_int, [c_int] * 6),
    "tem_conv3d_wgrad_gnorm": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64]
                               + [c_int] * 10 + [c_vp]),
    "tem_conv3d_wgrad_sums_ok": (c_int, [c_int] * 10),
    "tem_conv3d_wgrad_sums": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64]
                              + [c_int] * 10 + [c_vp]),
    "tem_norm_bwd_from_sums": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int,
                                       c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "tem_norm_stats": (c_int, [c_vp, c_i64, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_float,
                               c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "tem_norm_finalize_partials": (c_int, [c_vp, c_i64, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_float,
                                           c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tem_norm_bwd": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int,
                             c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "tem_maxpool3d_fwd": (c_int, [c_vp, c_i64, c_vp, c_i64] + [c_int] * 8 + [c_vp]),
    "tem_maxpool3d_fwd_stat_blocks": (c_i64, [c_int] * 5),
    "tem_maxpool3d_fwd_stats": (c_int, [c_vp, c_i64, c_vp, c_i64] + [c_int] * 8 + [c_vp, c_i64, c_vp]),
    "tem_maxpool3d_bwd": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64] + [c_int] * 8 + [c_vp]),
    "tem_maxpool3d_bwd_norm": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64] + [c_int] * 8
                               + [c_vp, c_i64, c_vp, c_vp]),
    "tem_upsample_bwd_norm": (c_int, [c_vp, c_i64, c_vp, c_i64] + [c_int] * 8 + [c_vp, c_i64, c_vp, c_i64, c_vp]),
    "tem_norm_bwd_coef": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp,
                                  c_vp, c_vp, c_vp, c_i64, c_vp]),
    "tem_upsample_stats": (c_int, [c_vp, c_i64] + [c_int] * 8 + [c_vp, c_vp]),
    "tem_norm_finalize_partials2": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_float,
                                            c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tem_upsample_fwd": (c_int, [c_vp, c_i64, c_vp, c_i64] + [c_int] * 8 + [c_vp]),
    "tem_upsample_fwd_stats_ok": (c_int, [c_int] * 4),
    "tem_upsample_fwd_stats": (c_int, [c_vp, c_i64, c_vp, c_i64] + [c_int] * 8 + [c_vp, c_vp]),
    "tem_upsample_bwd": (c_int, [c_vp, c_i64, c_vp, c_i64] + [c_int] * 8 + [c_vp]),
    "tem_dice_ws": (c_i64, [c_int, c_i64, c_int]),
    "tem_dice_sums": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_int, c_int, c_i64,
                              c_vp, c_vp, c_i64, c_vp]),
    "tem_dice_finalize": (c_int, [c_vp, c_int, c_double, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "tem_dice_grad": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_int,
                              c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_i64, c_vp]),
    "tem_dice_sums2": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_i64, c_vp, c_vp, c_i64,
                               c_int, c_vp]),
    "tem_dice_finalize2": (c_int, [c_vp, c_int, c_int, c_double, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "tem_dice_grad2": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_int,
                               c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_i64, c_int, c_float, c_float, c_vp]),
    "tem_adamw_step": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_float, c_float, c_float, c_float, c_float, c_i64,
                               c_float, c_vp]),
    "tem_adamw_hyper": (c_int, [c_vp, c_float, c_float, c_float, c_float, c_float, c_i64, c_float]),
    "tem_adamw_step_dev": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "tem_adamw_step_tab": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "tem_amp_unscale_dev": (c_int, [c_vp, c_i64, c_vp, c_vp]),
    "tem_amp_update_dev": (c_int, [c_vp, c_float, c_float, c_int, c_vp]),
    "tem_ema_update": (c_int, [c_vp, c_vp, c_i64, c_float, c_vp]),
    "tem_amp_unscale": (c_int, [c_vp, c_i64, c_float, c_vp, c_vp]),
    "tem_boundary_target": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "tem_boundary_target_mode": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "tem_affinity_target": (c_int, [c_vp, c_vp, c_int, c_int, c_int, ctypes.POINTER(c_int), c_int, c_int, c_i64,
                                    c_int, c_int, c_int, c_vp]),
    "tem_nchw_to_nhwc": (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_i64, c_vp]),
    "tem_nhwc_to_nchw": (c_int, [c_vp, c_i64, c_vp, c_int, c_int, c_i64, c_vp]),
    "tem_standardize": (c_int, [c_vp, c_vp, c_int, c_i64, c_float, c_vp, c_i64, c_vp]),
    "tem_spoco_ws": (c_i64, [c_int, c_int, c_i64, c_int, c_int, c_int]),
    "tem_label_range": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "tem_spoco_cluster_means": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "tem_spoco_pull": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_float, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "tem_spoco_means_terms": (c_int, [c_vp, c_int, c_int, c_float, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "tem_spoco_instance_dice": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_float, c_float, c_vp,
                                        c_vp, c_i64, c_vp]),
    "tem_spoco_push": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_float, c_vp, c_float, c_vp, c_i64,
                               c_vp, c_vp, c_i64, c_vp]),
    "tem_spoco_embed_grad": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]
                             + [c_float] * 6 + [c_vp, c_i64, c_int, c_vp, c_i64, c_vp]),
    "tem_zero_count": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "tem_zero_select": (c_int, [c_vp, c_i64, c_vp, c_vp, c_int, c_vp, c_vp]),
    "tem_spoco_consistency": (c_int, [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_int, c_float, c_float, c_vp,
                                      c_float, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "tem_affinity_side": (c_int, [c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), c_int, c_float,
                                  c_float, c_vp, c_float, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "tem_flip3d": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "tem_affine_warp3d": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "tem_elastic_field": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_float, c_float, c_vp, c_vp]),
    "tem_elastic_warp2d": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp]),
    "tem_block_load_reflect": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int] + [ctypes.POINTER(c_int)] * 4 + [c_vp]),
    "tem_block_store_inner": (c_int, [c_vp, ctypes.POINTER(c_int), c_vp, c_int, c_int, c_int, c_int, c_vp]
                              + [ctypes.POINTER(c_int)] * 3 + [c_vp]),
    "tem_accumulate_channels": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_int, c_int, c_i64] + [c_int] * 5 + [c_vp]),
    "tem_act_bwd": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp]),
    # round 5: activation storage types (TEM_ST_*) as explicit arguments
    "tem_maxpool3d_fwd_st": (c_int, [c_vp, c_i64, c_vp, c_i64] + [c_int] * 8 + [c_vp, c_i64, c_int, c_vp]),
    "tem_maxpool3d_bwd_st": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_i64] + [c_int] * 8
                             + [c_vp, c_i64, c_vp, c_vp, c_int, c_vp]),
    "tem_upsample_fwd_st": (c_int, [c_vp, c_i64, c_vp, c_i64] + [c_int] * 8 + [c_vp, c_int, c_vp]),
    "tem_upsample_bwd_st": (c_int, [c_vp, c_i64, c_vp, c_i64] + [c_int] * 8 + [c_vp, c_i64, c_vp, c_i64, c_int, c_vp]),
    "tem_upsample_stats_st": (c_int, [c_vp, c_i64] + [c_int] * 8 + [c_vp, c_int, c_vp]),
    "tem_norm_stats_st": (c_int, [c_vp, c_i64, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_float,
                                  c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp]),
    "tem_norm_bwd_st": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_i64,
                                c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_vp]),
    "tem_conv3d_wgrad_gnorm_st": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64]
                                  + [c_int] * 12 + [c_vp]),
    "tem_conv1x1_out_bwd_st": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int,
                                       c_vp, c_int, c_int, c_vp]),
}

_lib = None


BP_OUT_AMAX, BP_NORM_COEF, BP_NORM_SUMS = 1, 2, 4


class Byproducts(ctypes.Structure):
    """include/tem_hip.h: TemByproducts -- the optional by-products of ONE call, passed to a *_ex entry point"""
    _fields_ = [("out_amax", c_vp), ("coef_G", c_int), ("coef_mean", c_vp), ("coef_rstd", c_vp), ("coef", c_vp),
                ("sums_x", c_vp), ("sums_x_ld", c_i64), ("sums_mean", c_vp), ("sums_rstd", c_vp), ("sums_G", c_int),
                ("sums_part", c_vp), ("sums_nblk", c_i64), ("delivered", ctypes.c_uint)]


class TemError(RuntimeError):
    """HIP launch / runtime failure inside libtem_hip.so."""


def build(verbose=False):
    """Compile libtem_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(os.cpu_count() or 4)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building libtem_hip.so failed (see output above)")
    return LIB_PATH


def load():
    """Load (once) and type the shared library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libtem_hip.so not found at {LIB_PATH}: the MI355X kernels are not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C torch_em_amd/csrc). "
            "There is no CPU fallback for this path."
        )
    import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first so both share one HIP runtime)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if os.environ.get("TEM_LIB"):   # an older A/B build of the C-ABI (developer override): the symbol stays unavailable
                continue
            raise RuntimeError(f"{LIB_PATH} does not export {name}: rebuild it (make -C torch_em_amd/csrc)")
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    # dispatch options of the library (tem_set_option) can be preset from the HOST environment as TEM_OPT_<NAME>=<int>;
    # the library itself never reads the environment
    for key, val in os.environ.items():
        if key.startswith("TEM_OPT_"):
            check(lib.tem_set_option(key[8:].lower().encode(), int(val)), key)
    return lib


def set_option(name, value):
    """Select a kernel variant (names: include/tem_hip.h, tem_set_option)."""
    check(load().tem_set_option(name.encode(), int(value)), "tem_set_option")


def get_option(name):
    out = c_i64(0)
    check(load().tem_get_option(name.encode(), ctypes.byref(out)), "tem_get_option")
    return int(out.value)


_tls = threading.local()


def launch_on(idx):
    """Make device `idx` current for the launch being assembled; `check()` -- which wraps every launch -- puts the
    caller's device back, so an op on a `cuda:1` tensor does not move the process's current device for good."""
    import torch
    cur = torch.cuda.current_device()
    if idx is not None and idx != cur:
        if getattr(_tls, "restore", None) is None:
            _tls.restore = cur
        torch.cuda.set_device(idx)


def check(rc, what=""):
    """Map a C-ABI return code to the exception the reference raises for the same mistake."""
    prev = getattr(_tls, "restore", None)
    if prev is not None:
        import torch
        _tls.restore = None
        torch.cuda.set_device(prev)
    if rc == 0:
        return
    msg = load().tem_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg or f"{what}: invalid argument")
    raise TemError(f"{what}: {msg} (rc={rc})")
