"""ctypes binding of libacimg.so (the C ABI declared in include/acimg.h).

The product path has no CPU fallback: if the HIP library is missing or a symbol is absent this
module raises, loudly.  Build it with ``__graft_entry__.build()`` (or ``make`` in ``csrc/``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACIMG_LIB") or os.path.join(_HERE, "lib", "libacimg.so")     # ACIMG_LIB: another build (A/B tools)

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


class ConvDesc(C.Structure):
    """Mirror of AcimgConvDesc (include/acimg.h)."""

    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "C", "ldx", "K", "ldy", "OH", "OW", "R", "S", "stride", "pad_t", "pad_l",
        "ldw", "act")]

    def __repr__(self):
        return "ConvDesc(" + ", ".join("%s=%d" % (n, getattr(self, n)) for n, _ in self._fields_) + ")"


class Config(C.Structure):
    """Mirror of AcimgConfig (include/acimg.h): the launch heuristics' tuning record."""

    _fields_ = [(n, C.c_int32) for n in (
        "splitk_cut", "splitk_target", "splitk_handoff", "wgrad_minpix", "wgrad_halo", "split3_tile_bm",
        "split3_tile_bn", "tail_split", "tail_s", "trunk_persistent", "trunk_bk", "trunk_stagger", "trunk_dma_pos",
        "trunk_ring", "trunk_ring_bm", "trunk_halo")]


class SequenceDims(C.Structure):
    """Mirror of AcimgSequenceDims (include/acimg.h)."""

    _fields_ = [(n, C.c_int64) for n in (
        "classes", "location", "audio_height", "audio_width", "audio_depth", "mics", "samples", "video_height",
        "video_width", "video_depth", "audio_image_steps", "audio_data_steps", "video_steps", "audio_data_values")]


_P = C.c_void_p
_I = C.c_int
_L = C.c_long
_F = C.c_float
_D = C.c_double
_SZ = C.c_size_t
_DP = C.POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol include/acimg.h declares
PROTOTYPES = {
    "acimg_version": (_I, []),
    "acimg_last_error": (_I, [C.c_char_p, _SZ]),
    "acimg_conv2d_fwd": (_I, [_DP, _P, _P, _P, _P, _P, _P, _I, _P, _P, _SZ, _P, _P]),
    "acimg_conv2d_stats_rows": (_I, [_DP]),
    "acimg_conv2d_fwd_tiling": (_I, [_DP, C.POINTER(C.c_int)]),
    "acimg_config_default": (_I, [C.POINTER(Config)]),
    "acimg_configure": (_I, [C.POINTER(Config)]),
    "acimg_conv2d_fwd_workspace": (_SZ, [_DP]),
    "acimg_conv2d_split3_weight_bytes": (_SZ, [_DP]),
    "acimg_conv2d_split3_prepare": (_I, [_DP, _P, _P, _P]),
    "acimg_conv2d_bf16_prepare": (_I, [_DP, _P, _P, _P]),
    "acimg_conv2d_fwd_split3_stats_rows": (_I, [_DP]),
    "acimg_conv2d_fwd_split3_tiling": (_I, [_DP, C.POINTER(C.c_int)]),
    "acimg_conv2d_fwd_split3p_stats_rows": (_I, [_DP]),
    "acimg_conv2d_fwd_split3": (_I, [_DP, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "acimg_conv2d_fwd_bf16": (_I, [_DP, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "acimg_conv2d_split3_dgrad_weight_bytes": (_SZ, [_DP]),
    "acimg_conv2d_split3_prepare_dgrad": (_I, [_DP, _P, _P, _P]),
    "acimg_conv2d_dgrad_split3": (_I, [_DP, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P]),
    "acimg_conv2d_dgrad_bf16": (_I, [_DP, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P]),
    "acimg_conv2d_split3_prepare_multi": (_I, [_I, _P, _P, _P, _P, _P]),
    "acimg_conv2d_fwd_split3p_workspace": (_SZ, [_DP]),
    "acimg_conv2d_fwd_split3p": (_I, [_DP, _P, _SZ, _P, _P, _P, _P, _SZ, _P]),
    "acimg_conv2d_fwd_split1p": (_I, [_DP, _P, _SZ, _P, _P, _P, _P, _SZ, _P]),
    "acimg_conv2d_fwd_split3p_stats": (_I, [_DP, _P, _SZ, _P, _P, _P, _SZ, _P]),
    "acimg_gram_stats_workspace": (_SZ, [_L, _I]),
    "acimg_gram_stats": (_I, [_P, _SZ, _L, _I, _P, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _SZ, _P]),
    "acimg_conv2d_fwd_split3p_tail": (_I, [_DP, _P, _SZ, _P, _P, _P, _P, _SZ, _P, _SZ, _P, _SZ, _P]),
    "acimg_conv2d_fwd_split3p_tail_proj": (_I, [_DP, _P, _SZ, _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _SZ, _P]),
    "acimg_split_plane_bytes": (_SZ, [_L, _I]),
    "acimg_bn_relu_split": (_I, [_P, _P, _P, _I, _P, _SZ, _L, _I, _P]),
    "acimg_bn_add_relu_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _SZ, _P, _SZ, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "acimg_bn_relu_maxpool_split": (_I, [_P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "acimg_conv2d_dgrad": (_I, [_DP, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _SZ, _P, _P]),
    "acimg_conv2d_dgrad_workspace": (_SZ, [_DP]),
    "acimg_conv2d_wgrad": (_I, [_DP, _P, _P, _I, _P, _P, _P, _SZ, _P]),
    "acimg_conv2d_wgrad_workspace": (_SZ, [_DP]),
    "acimg_conv2d_wgrad_split3": (_I, [_DP, _P, _P, _I, _P, _P, _P, _SZ, _P]),
    "acimg_conv2d_wgrad_bf16": (_I, [_DP, _P, _P, _I, _P, _P, _P, _SZ, _P]),
    "acimg_conv2d_affine_input_ok": (_I, [_DP, _I]),
    "acimg_conv2d_wgrad_affine": (_I, [_DP, _I, _P, _P, _P, _I, _P, _I, _P, _P, _P, _SZ, _P]),
    "acimg_deconv_fwd": (_I, [_DP, _P, _P, _P, _P, _P, _SZ, _P, _P]),
    "acimg_deconv_dgrad": (_I, [_DP, _P, _I, _P, _P, _P, _I, _P, _SZ, _P, _P]),
    "acimg_deconv_wgrad": (_I, [_DP, _P, _P, _I, _P, _P, _P, _SZ, _P]),
    "acimg_deconv_workspace": (_SZ, [_DP]),
    "acimg_bn_finalize": (_I, [_P, _I, _I, _I, _D, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P]),
    "acimg_bn_add_relu": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "acimg_bn_relu_maxpool": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "acimg_bn_relu": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "acimg_bn_relu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "acimg_pad_channels": (_I, [_P, _P, _L, _I, _I, _P]),
    "acimg_pad_image": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "acimg_tile_mfcc": (_I, [_P, _P, _I, _I, _I, _P]),
    "acimg_minmax_workspace": (_SZ, [_I, _I, _I]),
    "acimg_minmax_fwd": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _P, _SZ, _P]),
    "acimg_minmax_bwd": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "acimg_latent_fwd": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _P]),
    "acimg_latent_bwd": (_I, [_P, _P, _P, _P, _I, _F, _P, _I, _I, _P]),
    "acimg_softplus_fwd": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "acimg_softplus_bwd": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _P]),
    "acimg_latent_linear_fwd": (_I, [_P, _P, _P, _I, _P, _I, _I, _P]),
    "acimg_tapconv_stats_rows": (_I, [_DP]),
    "acimg_tapconv_pack": (_I, [_DP, _P, _P, _I, _P]),
    "acimg_tapconv_unpack": (_I, [_DP, _P, _I, _P, _F, _P, _P]),
    "acimg_tapconv_gather": (_I, [_DP, _P, _I, _P, _P, _P]),
    "acimg_tapconv_scatter": (_I, [_DP, _P, _I, _P, _I, _P]),
    "acimg_triplet_loss_workspace": (_SZ, [_I]),
    "acimg_triplet_loss_fwd": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _F, _I, _P, _SZ, _P, _P]),
    "acimg_triplet_loss_bwd": (_I, [_P, _I, _P, _I, _I, _I, _F, _P, _SZ, _P, _I, _P, _I, _I, _P]),
    "acimg_latent_linear_bwd": (_I, [_P, _P, _P, _I, _F, _P, _I, _I, _P]),
    "acimg_maxpool_fwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "acimg_maxpool_relu_bwd": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "acimg_spatial_sum": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "acimg_spatial_sum_relu_bwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "acimg_clip_softmax_ce": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "acimg_bn_bwd_workspace": (_SZ, [_L, _I]),
    "acimg_bn_bwd": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _L, _I, _P, _I, _P, _P, _P, _SZ, _P]),
    "acimg_loss_scratch_bytes": (_SZ, []),
    "acimg_recon_loss": (_I, [_P, _P, _P, _P, _L, _F, _F, _P, _SZ, _P]),
    "acimg_grad_slice": (_I, [_P, _I, _P, _I, _P, _I, _L, _I, _I, _P]),
    "acimg_loss_finalize": (_I, [_P, _P, _I, _D, _F, _F, _F, _F, _P, _P]),
    "acimg_randn": (_I, [_P, _L, C.c_uint64, C.c_uint64, _P]),
    "acimg_sqerr_channels": (_I, [_P, _P, _L, _I, _P, _P]),
    "acimg_zero": (_I, [_P, _SZ, _P]),
    "acimg_spin": (_I, [C.c_uint64, _P, _P]),
    "acimg_sumsq": (_I, [_P, _L, _P, _P, _SZ, _P]),
    "acimg_axpy": (_I, [_F, _P, _P, _L, _P]),
    "acimg_adam_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P]),
    "acimg_mfcc_frontend": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "acimg_mfcc_frontend_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "acimg_stft_mag": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "acimg_absmax": (_I, [_P, _I, _I, _P, _P]),
    "acimg_resize_bilinear": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "acimg_filtfilt": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _SZ, _P]),
    "acimg_filtfilt_workspace": (_SZ, [_I, _I]),
    "acimg_find_logen": (_I, [_P, _P, _P, _L, _P]),
    "acimg_mask_iou": (_I, [_P, _P, _I, _I, _P, _P]),
    "acimg_gzip_inflate": (_I, [_P, _SZ, _P, _SZ, C.POINTER(_SZ)]),
    "acimg_tfrecord_index": (_L, [_P, _SZ, _P, _P, _L, _I]),
    "acimg_sequence_example_decode": (_I, [_P, _SZ, C.POINTER(SequenceDims), _P, _SZ, _P, _SZ, _P, _SZ]),
    "acimg_crc32c": (C.c_uint32, [_P, _SZ, C.c_uint32]),
}

_lib = None


class AcimgError(RuntimeError):
    pass


def load():
    """Load libacimg.so and bind every prototype; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime (libamdhip64.so.7).  Import it FIRST so libacimg.so
    # binds to that same runtime instance: device pointers and hipStream_t handles are shared between
    # torch (allocator, streams) and our kernels, which only works inside one runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise AcimgError(
            "libacimg.so not found at %s: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise AcimgError("libacimg.so lacks symbol %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    configure_from_env()
    return lib


def configure(**kw):
    """acimg_configure with the compiled-in defaults overridden by `kw` (AcimgConfig field names)"""
    cfg = Config()
    check(_lib.acimg_config_default(C.byref(cfg)), "config_default")
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AcimgError("unknown tuning field %r" % k)
        setattr(cfg, k, int(v))
    check(_lib.acimg_configure(C.byref(cfg)), "configure")
    return cfg


def configure_from_env(env=None):
    """The experiment switches of tools/ (ACIMG_* environment variables) are read HERE, once, when the library is
    loaded, and handed to acimg_configure: the library itself never reads the environment."""
    env = os.environ if env is None else env
    kw = {}
    for var, field in (("ACIMG_SPLITK_CUT", "splitk_cut"), ("ACIMG_SPLITK_TARGET", "splitk_target"),
                       ("ACIMG_WGRAD_MINPIX", "wgrad_minpix"), ("ACIMG_TAIL_S", "tail_s"), ("ACIMG_TRUNK_BK", "trunk_bk"),
                       ("ACIMG_TRUNK_STAGGER", "trunk_stagger"), ("ACIMG_TRUNK_DMA_POS", "trunk_dma_pos"),
                       ("ACIMG_TRUNK_RING", "trunk_ring"), ("ACIMG_TRUNK_RING_BM", "trunk_ring_bm"),
                       ("ACIMG_TRUNK_HALO", "trunk_halo")):
        if env.get(var):
            kw[field] = int(env[var])
    for var, field in (("ACIMG_NO_SPLITK_HANDOFF", "splitk_handoff"), ("ACIMG_NO_WGRAD_HALO", "wgrad_halo"),
                       ("ACIMG_NO_TAIL_SPLIT", "tail_split"), ("ACIMG_NO_PERSISTENT", "trunk_persistent"),
                       ("ACIMG_NO_RING", "trunk_ring")):
        if env.get(var):
            kw[field] = 0
    if env.get("ACIMG_SPLIT3_TILE"):
        bm, bn = env["ACIMG_SPLIT3_TILE"].lower().split("x")
        kw["split3_tile_bm"], kw["split3_tile_bn"] = int(bm), int(bn)
    return configure(**kw) if kw else None


def last_error():
    buf = C.create_string_buffer(512)
    load().acimg_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc, what=""):
    if rc != 0:
        raise AcimgError("%s failed (rc=%d): %s" % (what or "acimg call", rc, last_error()))
