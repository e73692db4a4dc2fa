"""ctypes binding of ``csrc/libfie_hip.so`` (C ABI: include/fie.h) + thin tensor-level wrappers.

PyTorch-ROCm is plumbing here: it owns device memory (``tensor.data_ptr()``) and the current stream; every
multiply-add of the hot path runs in the hand-written HIP kernels behind these wrappers.  There is NO fallback:
if the library is missing or the device is not gfx950 the first call raises.
"""
import ctypes
import os
import subprocess

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("FIE_LIB_PATH") or os.path.join(_CSRC, "libfie_hip.so")      # FIE_LIB_PATH: A/B builds of the same sources (tools/)

ACT_NONE, ACT_SILU, ACT_GELU, ACT_QUICK_GELU, ACT_GEGLU = 0, 1, 2, 3, 4

_c = ctypes
_P, _I, _L, _F = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float

# name -> argtypes, mirrors include/fie.h one to one (tests/test_cabi_cpu.py checks every symbol is exported)
SIGNATURES = {
    "fie_version": [],
    "fie_ctx_create": [_I, _P, _c.POINTER(_P)],
    "fie_ctx_set_stream": [_P, _P],
    "fie_ctx_error_flag": [_P, _P],
    "fie_ctx_destroy": [_P],
    "fie_weights_register": [_P, _c.c_char_p, _P, _L, _L],
    "fie_weights_clear": [_P],
    "fie_vae_decode_workspace_bytes": [_P, _I, _I],
    "fie_vae_decode_f16": [_P, _P, _P, _P, _P, _L],
    "fie_vae_encode_workspace_bytes": [_P],
    "fie_vae_encode_f16": [_P, _P, _P, _P, _P, _L],
    "fie_clip_text_workspace_bytes": [_P],
    "fie_clip_text_forward_f16": [_P, _P, _c.c_char_p, _P, _P, _P, _P, _P, _L],
    "fie_unet_num_residuals": [_P],
    "fie_unet_workspace_bytes": [_P],
    "fie_unet_forward_f16": [_P, _P, _c.c_char_p, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L],
    "fie_controlnet_workspace_bytes": [_P],
    "fie_controlnet_forward_f16": [_P, _P, _c.c_char_p, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _L],
    "fie_add_f16": [_P, _P, _P, _P, _L],
    "fie_copy_rows_f16": [_P, _P, _L, _P, _L, _I, _I],
    "fie_program_begin": [_P, _c.POINTER(_P)],
    "fie_program_end": [_P],
    "fie_program_launches": [_P],
    "fie_program_run": [_P, _P],
    "fie_program_destroy": [_P, _P],
    "fie_graph_register": [_P, _c.c_char_p, _P],
    "fie_unet_forward": [_P],
    "fie_controlnet_forward": [_P],
    "fie_vae_encode": [_P],
    "fie_vae_decode": [_P],
    "fie_clip_text_forward": [_P],
    "fie_gemm_f16": [_P, _P, _L, _I, _P, _L, _P, _L, _P, _L, _I, _I, _I, _P, _P, _L, _I, _P, _L, _F, _I],
    "fie_gemm_ln_f16": [_P, _P, _L, _P, _L, _P, _F, _P, _L, _I, _I, _I, _I],
    "fie_conv3x3_nhwc_f16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P, _L, _I, _P, _P, _L, _P, _L, _F, _I],
    "fie_pack_rows_f8": [_P, _P, _L, _I, _I, _P, _L, _I, _P, _I],
    "fie_pack_conv3x3_f8": [_P, _P, _I, _I, _I, _P, _L, _I, _P],
    "fie_gemm_w8_f16": [_P, _P, _L, _I, _P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _P, _P, _L, _I, _P, _L, _F, _I],
    "fie_conv3x3_w8_nhwc_f16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P, _P, _L, _I, _P, _P, _L, _P, _L, _F, _I],
    "fie_gemm_x8_f16": [_P, _P, _L, _P, _L, _P, _F, _P, _L, _I, _I, _I, _P, _P, _L, _I, _P, _L, _F, _I, _I, _F],
    "fie_layernorm_f16_o8": [_P, _P, _L, _P, _L, _L, _I, _P, _P, _F, _F],
    "fie_attention_f16_o8": [_P, _P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _F, _I, _F],
    "fie_quantize_f8": [_P, _P, _L, _P, _L, _L, _I, _F],
    "fie_amax_f16": [_P, _P, _L, _L, _I, _P],
    "fie_groupnorm_coef_f16": [_P, _I, _I, _L, _I, _P, _P, _F, _P, _P, _I, _P],
    "fie_conv3x3_gn_ok": [_P, _I, _I, _I, _I, _I, _I],
    "fie_conv3x3_gn_nhwc_f16": [_P, _P, _I, _I, _I, _I, _P, _I, _P, _L, _P, _L, _I, _P, _P, _L],
    "fie_canny_rgb_device_begin_u8": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "fie_canny_rgb_device_finish_u8": [_P, _I, _I, _P, _P, _P, _P],
    "fie_weights_clear_prefix": [_P, _c.c_char_p],
    "fie_step_cache_bind": [_P, _c.c_char_p, _P, _L],
    "fie_step_cache_reset": [_P, _c.c_char_p],
    "fie_unet_step_cache_bytes": [_P, _I],
    "fie_groupnorm_nhwc_f16_o8": [_P, _P, _I, _P, _I, _P, _I, _L, _I, _P, _P, _F, _I, _P, _F],
    "fie_groupnorm_stats_nhwc_f16_o8": [_P, _P, _I, _P, _I, _L, _I, _P, _P, _F, _I, _P, _P, _I, _F],
    "fie_conv3x3_x8_nhwc_f16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P, _F, _P, _L, _I, _P, _P, _L, _P, _L, _F, _I],
    "fie_attention_f16": [_P, _P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _F, _I],
    "fie_groupnorm_workspace_bytes": [_I, _L, _I],
    "fie_groupnorm_nhwc_f16": [_P, _P, _I, _P, _I, _P, _I, _L, _I, _P, _P, _F, _I, _P],
    "fie_layernorm_f16": [_P, _P, _L, _P, _L, _L, _I, _P, _P, _F],
    "fie_sinusoid_f16": [_P, _P, _I, _I, _I, _P, _L, _I],
    "fie_time_embed_workspace_bytes": [_I],
    "fie_time_embed_f16": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _L, _P, _L, _P],
    "fie_clip_embed_f16": [_P, _P, _I, _I, _I, _P, _P, _P],
    "fie_pixels_in_u8_f16": [_P, _P, _I, _I, _I, _P, _I],
    "fie_pixels_out_f16_u8": [_P, _P, _L, _I, _I, _P],
    "fie_latent_prep": [_P, _P, _P, _P, _L, _F, _F, _F, _P, _P, _I],
    "fie_lcm_step": [_P, _P, _L, _I, _F, _P, _P, _L, _F, _F, _F, _F, _F, _F, _P, _I, _F, _P],
    "fie_pack_rows_f16": [_P, _P, _L, _I, _I, _P, _L, _I, _I],
    "fie_pack_conv3x3_f16": [_P, _P, _I, _I, _I, _P, _L, _I],
    "fie_canny_rgb_u8": [_P, _I, _I, _I, _I, _P],
    "fie_gemm_f32": [_P, _P, _L, _I, _P, _L, _P, _L, _I, _P, _L, _I, _I, _I, _P, _P, _L, _I, _P, _L, _F, _I, _I, _I, _L, _L, _L, _L, _L, _L],
    "fie_conv3x3_nhwc_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P, _L, _I, _P, _P, _L, _P, _L, _F, _I],
    "fie_softmax_rows_f32": [_P, _P, _L, _I, _L, _F, _I, _I],
    "fie_groupnorm_nhwc_f32": [_P, _P, _I, _P, _I, _P, _I, _L, _I, _P, _P, _F, _I, _P],
    "fie_layernorm_f32": [_P, _P, _L, _P, _L, _L, _I, _P, _P, _F],
    "fie_sinusoid_f32": [_P, _P, _I, _I, _I, _P, _L, _I],
    "fie_clip_embed_f32": [_P, _P, _I, _I, _I, _P, _P, _P],
    "fie_pixels_in_u8_f32": [_P, _P, _I, _I, _I, _P, _I],
    "fie_pixels_out_f32_u8": [_P, _P, _L, _I, _I, _P],
    "fie_latent_prep_f32": [_P, _P, _P, _P, _L, _F, _F, _F, _P, _P, _I],
    "fie_lcm_step_f32": [_P, _P, _L, _I, _F, _P, _P, _L, _F, _F, _F, _F, _F, _F, _P, _I, _F, _P],
    "fie_canny_workspace_bytes": [_I, _I],
    "fie_canny_rgb_device_u8": [_P, _P, _I, _I, _I, _I, _P, _P, _c.POINTER(_I)],
    "fie_resize_rgb_u8": [_P, _P, _I, _I, _P, _I, _I, _P, _P, _I, _P, _P, _I, _P],
    "fie_debug_force_tile": [_P, _I],
    "fie_debug_attn_variant": [_P, _I],
    "fie_debug_gn_onepass": [_P, _I],
    "fie_debug_tile_override": [_P, _c.c_char_p],
    "fie_debug_last_gemm_kernel": [_P],
    "fie_prefetch": [_P, _P, _L, _P, _I],
    "fie_conv3x3_plus_nhwc_f16": [_P, _P, _I, _I, _I, _I, _P, _L, _P, _L, _I, _P, _P, _L, _F, _I, _P, _L, _I, _P, _L, _I],
    "fie_conv_up2x_nhwc_f16": [_P, _P, _I, _I, _I, _I, _P, _L, _I, _P, _L, _I, _P, _P, _L, _F, _I],
    "fie_gn_stats_target": [_P, _P, _L, _I],
    "fie_gn_stats_bytes": [_I, _L, _I],
    "fie_groupnorm_stats_nhwc_f16": [_P, _P, _I, _P, _I, _L, _I, _P, _P, _F, _I, _P, _P, _I],
    "fie_gemm_autotune": [_P, _I],
    "fie_gemm_autotune_report": [_P, ctypes.c_char_p, _I],
    "fie_gemm_autotune_load": [_P, ctypes.c_char_p],
    "fie_debug_tune_exclude": [_P, ctypes.c_char_p],
    "fie_debug_gemm_probe": [_P, _I],
    "fie_debug_epilogue_prefetch": [_P, _I],
    "fie_debug_gemm_stamps": [_P, _P],
    "fie_splitk_workspace": [_P, _P, _L],
    "fie_debug_splitk": [_P, _I],
    "fie_debug_oplog": [_P, _I],
    "fie_debug_oplog_mark": [_P, _c.c_char_p],
    "fie_debug_oplog_read": [_P, _c.c_char_p, _L],
}

_lib = None


def build(force=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", _CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", _CSRC, "-j4"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libfie_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no non-HIP fallback for the hot path)")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = _L if name in ("fie_groupnorm_workspace_bytes", "fie_canny_workspace_bytes", "fie_time_embed_workspace_bytes", "fie_gn_stats_bytes", "fie_debug_oplog_read", "fie_vae_decode_workspace_bytes",
                                       "fie_vae_encode_workspace_bytes", "fie_clip_text_workspace_bytes", "fie_unet_workspace_bytes", "fie_controlnet_workspace_bytes",
                                       "fie_unet_step_cache_bytes") else _I
        _lib.fie_last_error.restype = ctypes.c_char_p
        _lib.fie_last_error.argtypes = []
        _lib.fie_debug_last_gemm_kernel.restype = ctypes.c_char_p
    return _lib


def last_gemm_kernel(ctx=None):
    """Kernel / tile the last GEMM or conv launch of `ctx` used (names the roofline kernel in bench.py)."""
    ctx = ctx or context(torch.cuda.current_device())
    return lib().fie_debug_last_gemm_kernel(ctx.h).decode()


class FieError(RuntimeError):
    pass


def fold_layernorm_tables(w, bias, gamma, beta, geglu=False):
    """LayerNorm folded into the Linear that consumes it:  LN(x) W^T + b = rstd * (x Wf^T - mean * S) + b'  with Wf = f16(W * gamma), S[n] = sum_k Wf[n, k]
    (over the ROUNDED folded weights, so that x Wf^T - mean * S is exactly (x - mean) Wf^T), b' = W beta + b in fp32.  Returns (Wf [N, K] f16,
    table [N, 2] fp32 = (S, b')), the table's rows in the packed column order (GEGLU: value / gate rows interleaved, as fie_pack_rows_f16 with interleave2)."""
    w = w.float()
    n = w.shape[0]
    wf = (w * gamma.float()[None, :]).to(torch.float16)
    b = w @ beta.float()
    if bias is not None:
        b = b + bias.float()
    tab = torch.stack([wf.float().sum(1), b], 1)
    if geglu:
        tab = torch.stack([tab[: n // 2], tab[n // 2:]], 1).reshape(n, 2)
    return wf, tab.contiguous()


def _chk(rc):
    if rc != 0:
        raise FieError(f"libfie_hip error {rc}: {lib().fie_last_error().decode()}")


def _p(t):
    return None if t is None else t.data_ptr()


GRAPH_NAMES = ("unet_forward", "controlnet_forward", "vae_encode", "vae_decode", "clip_text_forward")


class VaeConfig(ctypes.Structure):
    """include/fie.h: fie_vae_config (the C++ decoder walk, csrc/graphs.cpp)."""
    _fields_ = [("latent_h", _I), ("latent_w", _I), ("num_blocks", _I), ("block_out_channels", _I * 8), ("layers_per_block", _I),
                ("norm_num_groups", _I), ("norm_eps", _F), ("out_channels", _I), ("prefix", ctypes.c_char_p)]


class ClipConfig(ctypes.Structure):
    """include/fie.h: fie_clip_config."""
    _fields_ = [("batch", _I), ("tokens", _I), ("hidden", _I), ("heads", _I), ("layers", _I), ("intermediate", _I), ("projection_dim", _I),
                ("quick_gelu", _I), ("eps", _F)]


class UnetConfig(ctypes.Structure):
    """include/fie.h: fie_unet_config (UNet and ControlNet)."""
    _fields_ = [("batch", _I), ("latent_h", _I), ("latent_w", _I), ("text_len", _I), ("num_blocks", _I), ("block_out_channels", _I * 4),
                ("layers_per_block", _I), ("down_attn", (_I * 4) * 4), ("up_attn", (_I * 4) * 4), ("mid_attn", _I), ("mid_resnets", _I), ("head_dim", _I),
                ("norm_num_groups", _I), ("norm_eps", _F), ("cross_attention_dim", _I), ("addition_time_embed_dim", _I), ("pooled_dim", _I),
                ("num_cond_channels", _I), ("cond_channels", _I * 8)]


class Program:
    """A recorded launch program (include/fie.h, fie_program_*).  `keep` holds every tensor the op wrappers allocated while it
    was recorded: the program references their memory by raw pointer."""

    def __init__(self, ctx):
        self.ctx, self.h, self.keep, self.outputs = ctx, _P(), [], None

    def __len__(self):
        return lib().fie_program_launches(self.h)

    def run(self):
        """Re-issues the launches on torch's current stream.  Ordered against the program's own previous pass by the library (include/fie.h,
        ORDERING CONTRACT); the producers of the static inputs are the caller's to order (`stream.wait_stream(...)`)."""
        self.ctx.sync_stream()
        _chk(lib().fie_program_run(self.ctx.h, self.h))
        return self.outputs

    def register(self, name):
        """Bind to one of the graph-level C entries (fie_unet_forward, ...); `run_named(name)` then calls THAT entry."""
        assert name in GRAPH_NAMES, name
        _chk(lib().fie_graph_register(self.ctx.h, name.encode(), self.h))

    def close(self):
        if self.h:
            lib().fie_program_destroy(self.ctx.h, self.h)
            self.h = _P()


class W8:
    """A packed fp8 (e4m3) weight: bytes [Npad, Kpad] + one fp32 dequantisation scale per output channel (include/fie.h,
    fie_pack_*_f8).  `stride(0)` mirrors the packed fp16 tensors so call sites stay the same."""

    def __init__(self, q, scale):
        self.q, self.scale = q, scale

    def stride(self, d):
        return self.q.stride(d)

    def dequant(self):
        return self.q.view(torch.float8_e4m3fn).float() * self.scale[:, None]


class Context:
    """One fie_ctx per (process, device); launches go to torch's current stream on that device."""

    def __init__(self, device=0, dtype=torch.float16):
        if dtype not in (torch.float16, torch.float32):
            raise ValueError(f"dtype {dtype} not supported (float16 or float32)")
        self.dtype = dtype                           # storage / arithmetic type of every op issued through this context
        self.f32 = dtype == torch.float32
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the fie_amd hot path has no CPU fallback")
        self.device = torch.device("cuda", device)
        h = _P()
        _chk(lib().fie_ctx_create(device, None, ctypes.byref(h)))
        self.h = h
        # device error word (include/fie.h: FIE_DEVERR_*): kernels that must give up set it; check_device_errors() reads it
        self._err = torch.zeros(4, device=self.device, dtype=torch.int32)
        self._err_host = torch.zeros(4, dtype=torch.int32).pin_memory()
        _chk(lib().fie_ctx_error_flag(self.h, self._err.data_ptr()))
        self._stream = None
        self._gn_ws = {}
        self._gn_stats = {}
        self._gn_gen = 0
        self._sk_ws = {}               # split-K workspaces (fie_splitk_workspace), one per (stream, graph slot)
        self._sk_bound = None
        self._sk_pinned = False        # True while a program records: the program's OWN workspace stays bound (record())
        self._cap_stream = None
        self._prog_ws = []             # weak references to the live programs' workspaces (splitk_counters_clear)
        self.splitk_bytes = int(os.environ.get("FIE_SPLITK_MB", "96")) << 20      # 0: never split K
        self.gn_from_epilogue = os.environ.get("FIE_GN_FROM_EPILOGUE", "1") != "0"
        self.gn_quads = os.environ.get("FIE_GN_QUADS", "1") != "0"      # ... also for the UNet's 20 / 40-channel groups (quad partials); A/B switch
        self._epi_prefetch = True                 # C side default (fie_debug_epilogue_prefetch)
        self.up2x_parity = os.environ.get("FIE_UP2X_PARITY", "1") != "0"
        # transformer blocks: LayerNorm folded into the projection that consumes it (fie_gemm_ln_f16: no LayerNorm launch, no normalised tensor); A/B switch
        self.ln_fold = os.environ.get("FIE_LN_FOLD", "1") != "0"
        # ... also norm3 -> the GEGLU projection (256x320 tile): built and bit-checked, but the row sums add 11-15 us to a 49 us launch there (every wave of a
        # wave row sums the same fragments; dealing them out needs branches in the K loop, which cost as much): off, LayerNorm launch + plain FF1
        self.kv_group = os.environ.get("FIE_KV_GROUP", "1") != "0"      # text K/V of all transformer blocks of a net as one GEMM per image (nn.py: _finish_kv); A/B switch
        self.ln_fold_which = os.environ.get("FIE_LN_FOLD_WHICH", "qkv,q2")      # which of the two folds run (A/B)
        self.ln_fold_ff1 = os.environ.get("FIE_LN_FOLD_FF1", "0") != "0"
        self.conv_plus_shortcut = os.environ.get("FIE_CONV_PLUS", "1") != "0"   # resnet conv2 + 1x1 shortcut as one GEMM (fie_conv3x3_plus_nhwc_f16)      # 2x-upsampling convs as four 2x2 convs (fie_conv_up2x_nhwc_f16)
        self._resize_tables = {}       # (in, out) -> (taps, bounds, ksize) of the LANCZOS resample, on the device
        self.ws_tag = 0
        self._keep = None              # list collecting the tensors allocated while a program is being recorded (Context.record)
        self.w8 = False                # while True, pack_linear / pack_conv3x3 quantise eligible weights to fp8 e4m3 (see W8)
        # fp8 ACTIVATIONS for the transformer-block projections of an fp8-weight model (csrc/gemm_x8.hip): the producers (LayerNorm, attention,
        # the GEGLU epilogue) write e4m3 and the GEMMs run the block-scaled MFMA.  FIE_A8=0 keeps fp16 activations (round-2 behaviour: A/B)
        self.a8 = os.environ.get("FIE_A8", "1") != "0"
        # GroupNorm + SiLU applied by the consuming halo conv (conv3x3_gn): built, bit-exact, and 20-70 % SLOWER than the two launches (the transform's
        # exp / rcp do not fit the issue slots the MFMA segments leave: profiles/r04_gn_apply_in_the_halo_conv_negative.log) -- off unless FIE_GN_FUSE_CONV=1
        self.gn_fuse_conv = os.environ.get("FIE_GN_FUSE_CONV", "0") == "1"
        self.calib = False             # True during HipImg2ImgPipeline.calibrate_fp8: fp8-activation layers run f16 activations and record max |x|
        # Tile / split-K choices from a file (include/fie.h: fie_gemm_autotune_load): FIE_TUNE_TABLE=<report of an earlier process>.  With
        # FIE_TUNE_FROZEN=1 the pipelines never time anything new (shapes outside the table use the built-in rule): one choice per shape on
        # every box, which is what the test session pins (tests/conftest.py); direct ctx.autotune(1) calls still tune live.
        self.tune_frozen = os.environ.get("FIE_TUNE_FROZEN", "0") == "1"
        self.tune_table_entries = 0
        self.load_tune_table()

    def fetch_device_errors(self):
        """Queues the 16-byte D2H copy of the error word on the current stream (call before a synchronisation that happens anyway)."""
        self._err_host.copy_(self._err, non_blocking=True)

    def check_device_errors(self, fetch=True):
        """Raises FieError when a kernel reported a device-side failure since the last check (clears the word)."""
        if fetch:
            self.fetch_device_errors()
            torch.cuda.current_stream(self.device).synchronize()
        code = int(self._err_host[0])
        if code:
            self._err.zero_()
            self._err_host.zero_()
            what = {1: "fie_time_embed_f16: the in-launch barrier timed out (a workgroup never arrived); the timestep embedding of that "
                       "launch is invalid"}.get(code, "unknown device error")
            raise FieError(f"device error {code}: {what}")

    def sync_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._stream:
            _chk(lib().fie_ctx_set_stream(self.h, s))
            self._stream = s

    def load_tune_table(self, path=None):
        """Remembered tile / split-K choices from the text of an autotune report (FIE_TUNE_TABLE by default).  Returns the number loaded."""
        path = path or os.environ.get("FIE_TUNE_TABLE")
        if not path or self.f32:
            return 0
        with open(path) as f:
            n = lib().fie_gemm_autotune_load(self.h, f.read().encode())
        if n < 0:
            raise FieError(f"fie_gemm_autotune_load: {lib().fie_last_error().decode()}")
        self.tune_table_entries = n
        return n

    def _bind_splitk(self):
        """Split-K launches that may run concurrently must not share arrival counters / slabs: every (stream, graph slot) owns a
        workspace (zeroed once: the kernels leave the counters zero), bound to the C context before a GEMM / conv or a C++ graph walk is
        issued there.  While a program records, the program's own workspace stays bound instead (record())."""
        if self._sk_pinned or self.f32 or not self.splitk_bytes:
            return
        key = (self._stream, self.ws_tag)
        if key == self._sk_bound:
            return
        ws = self._sk_ws.get(key)
        if ws is None:
            # under stream capture torch.zeros would become a 96 MB memset node replayed with the graph: callers that capture bind the capture
            # stream's workspace BEFORE entering the capture (pipe._capture); a key first met inside a capture keeps working, only slower
            ws = self._sk_ws[key] = torch.zeros(self.splitk_bytes, device=self.device, dtype=torch.uint8)
        _chk(lib().fie_splitk_workspace(self.h, ws.data_ptr(), ws.numel()))
        self._sk_bound = key

    def capture_stream(self):
        """The stream this context's hipGraph captures run on (`torch.cuda.graph(g, stream=ctx.capture_stream())`), with the split-K workspace
        of (that stream, current ws_tag) bound BEFORE the capture begins: allocated inside a capture, its torch.zeros would be a 96 MB memset
        node that every replay re-runs."""
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._cap_stream):
            self.sync_stream()
            self._bind_splitk()
        torch.cuda.synchronize(self.device)
        return self._cap_stream

    def splitk_counters_clear(self):
        """True when every arrival counter of every split-K workspace of this context is zero (synchronises): what each launch must leave."""
        torch.cuda.synchronize(self.device)
        live = list(self._sk_ws.values()) + [w for w in (r() for r in self._prog_ws) if w is not None]
        return all(int(ws[:16384].view(torch.int32).abs().max()) == 0 for ws in live)

    # ------------------------------------------------------------------ launch programs / graph-level entries
    def record(self):
        """Context manager: `with ctx.record() as prog: out = model(...)` executes the ops as usual AND records every launch into
        `prog`.  Works on whatever torch stream is current; tensors allocated by the op wrappers meanwhile stay referenced by
        `prog.keep` (torch must not hand their memory to anybody else while the program lives)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prog = Program(self)
            self.sync_stream()
            # the program OWNS its split-K workspace (include/fie.h, ORDERING CONTRACT (2)): the recorded launches freeze its pointer, and no
            # stream's eager launches or other program ever share its arrival counters / slabs
            own = None
            if not self.f32 and self.splitk_bytes:
                own = torch.zeros(self.splitk_bytes, device=self.device, dtype=torch.uint8)
                prog.keep.append(own)
                import weakref
                self._prog_ws = [r for r in self._prog_ws if r() is not None] + [weakref.ref(own)]
                _chk(lib().fie_splitk_workspace(self.h, own.data_ptr(), own.numel()))
                self._sk_pinned = True
            _chk(lib().fie_program_begin(self.h, ctypes.byref(prog.h)))
            self._keep = prog.keep
            try:
                yield prog
            finally:
                self._keep = None
                _chk(lib().fie_program_end(self.h))
                if own is not None:
                    self._sk_pinned = False
                    self._sk_bound = None           # the next GEMM / conv re-binds its stream's workspace
        return cm()

    def run_named(self, name):
        """Run the program registered under `name` through its C-ABI graph entry (fie_unet_forward, fie_vae_decode, ...)."""
        self.sync_stream()
        _chk(getattr(lib(), "fie_" + name)(self.h))

    def _alloc(self, shape, dtype=None, zero=False):
        t = (torch.zeros if zero else torch.empty)(shape, device=self.device, dtype=dtype or self.dtype)
        if self._keep is not None:
            self._keep.append(t)
        return t

    # ------------------------------------------------------------------ tuning / test hooks (per ctx)
    def force_tile(self, code):
        """0 = heuristic; see include/fie.h for the codes.  An ineligible code makes the op raise FieError."""
        _chk(lib().fie_debug_force_tile(self.h, int(code)))

    def prefetch(self, t, stream=None, blocks=16):
        """Pull tensor t (a packed weight) into the Infinity Cache on `stream` (a torch stream; default: the context's)."""
        _chk(lib().fie_prefetch(self.h, t.data_ptr(), t.numel() * t.element_size(), stream.cuda_stream if stream is not None else None, int(blocks)))

    def autotune(self, on=True):
        """Per-shape tile autotune (include/fie.h: fie_gemm_autotune): first eager launch of a shape times the eligible tiles."""
        _chk(lib().fie_gemm_autotune(self.h, int(on)))                # 0 off, 1 tune new shapes, 2 remembered shapes only

    def tune_exclude(self, codes):
        """Tile codes ("63,96"; 10000 = all split-K variants) the tuner must not offer; forgets remembered choices (A/B tools)."""
        return lib().fie_debug_tune_exclude(self.h, (codes or "").encode())

    def autotune_report(self):
        buf = ctypes.create_string_buffer(1 << 20)
        n = lib().fie_gemm_autotune_report(self.h, buf, len(buf))
        return n, buf.value.decode()

    def splitk(self, on=True):
        """False: no launch of this context splits K (A/B switch; the tuner's remembered split choices fall back to the unsplit tile)."""
        _chk(lib().fie_debug_splitk(self.h, int(bool(on))))

    def oplog(self, on=True):
        """Launch log (include/fie.h: fie_debug_oplog): one line per kernel launch with the op's shape description."""
        _chk(lib().fie_debug_oplog(self.h, int(on)))
        self._oplog_on = bool(on)

    def oplog_mark(self, text):
        if getattr(self, "_oplog_on", False):
            _chk(lib().fie_debug_oplog_mark(self.h, text.encode()))

    def oplog_read(self):
        n = lib().fie_debug_oplog_read(self.h, None, 0)
        buf = ctypes.create_string_buffer(int(n) + 16)
        lib().fie_debug_oplog_read(self.h, buf, len(buf))
        return [l for l in buf.value.decode().split("\n") if l]

    def gemm_stamps(self, buf):
        """Device int32 tensor [tiles * waves * 8] the stamped ring kernels (tile codes 97 / 98) write their cycle sums to; None detaches."""
        _chk(lib().fie_debug_gemm_stamps(self.h, buf.data_ptr() if buf is not None else None))

    def gemm_probe(self, mode):
        """TIMING-ONLY probes of the LDS-DMA GEMM kernels (outputs are wrong): 0 off, 1 loads dropped, 2 all tiles load tile (0,0)."""
        _chk(lib().fie_debug_gemm_probe(self.h, int(mode)))

    @property
    def epi_prefetch(self):
        return self._epi_prefetch

    @epi_prefetch.setter
    def epi_prefetch(self, on):
        """A/B switch: bias row / residual tile of the ring GEMM and conv kernels loaded ahead of the K loop (default) or in the epilogue."""
        _chk(lib().fie_debug_epilogue_prefetch(self.h, int(bool(on))))
        self._epi_prefetch = bool(on)

    def tile_override(self, spec):
        """"mode,M,N,K=code;..." per-shape tile codes (None clears).  Returns the number of entries parsed."""
        return lib().fie_debug_tile_override(self.h, spec.encode() if spec else None)

    def close(self):
        if self.h:
            lib().fie_ctx_destroy(self.h)
            self.h = None

    # ------------------------------------------------------------------ weight packing
    def pack_linear(self, w, geglu=False, quant=True):
        """[N, K] f16 -> packed [Npad][Kpad] (zero padded); geglu interleaves (value, gate) rows.
        fp32 contexts keep plain [N, K] fp32 weights (rows interleaved for GEGLU)."""
        self.sync_stream()
        if self.f32:
            w = w.to(self.device, torch.float32)
            if geglu:
                n = w.shape[0]
                w = torch.stack([w[: n // 2], w[n // 2:]], 1).reshape(n, -1)
            return w.contiguous()
        w = w.to(self.device, torch.float16).contiguous()
        n, k = w.shape
        npad, kpad = (n + 127) // 128 * 128, (k + 63) // 64 * 64
        if self.w8 and quant and self.a8:
            kpad = (k + 127) // 128 * 128               # the fp8-activation kernels step K by 128
        if self.w8 and quant:
            q = torch.empty((npad, kpad), device=self.device, dtype=torch.uint8)
            scale = torch.empty((npad,), device=self.device, dtype=torch.float32)
            _chk(lib().fie_pack_rows_f8(self.h, _p(w), k, n, k, _p(q), kpad, npad, _p(scale), int(geglu)))
            return W8(q, scale)
        out = torch.empty((npad, kpad), device=self.device, dtype=torch.float16)
        _chk(lib().fie_pack_rows_f16(self.h, _p(w), k, n, k, _p(out), kpad, npad, int(geglu)))
        return out

    def pack_conv3x3(self, w, cin_pad=None, quant=True):
        """OIHW f16 -> packed [Npad][Kpad], k = (ky*3+kx)*cin_pad + ci.  fp32 contexts: [Cout][9*cin_pad] fp32."""
        self.sync_stream()
        if self.f32:
            co, ci = w.shape[:2]
            cp = cin_pad or (ci + 7) // 8 * 8
            out = torch.zeros((co, 3, 3, cp), device=self.device, dtype=torch.float32)
            out[..., :ci] = w.to(self.device, torch.float32).permute(0, 2, 3, 1)
            return out.reshape(co, 9 * cp)
        w = w.to(self.device, torch.float16).contiguous()
        co, ci = w.shape[:2]
        cin_pad = cin_pad or (ci + 7) // 8 * 8
        npad, kpad = (co + 127) // 128 * 128, (9 * cin_pad + 63) // 64 * 64
        if self.w8 and quant and cin_pad % 64 == 0:  # the fp8 kernels are the LDS-DMA ones: K-steps must not straddle a 3x3 tap
            q = torch.empty((npad, kpad), device=self.device, dtype=torch.uint8)
            scale = torch.empty((npad,), device=self.device, dtype=torch.float32)
            _chk(lib().fie_pack_conv3x3_f8(self.h, _p(w), co, ci, cin_pad, _p(q), kpad, npad, _p(scale)))
            return W8(q, scale)
        out = torch.empty((npad, kpad), device=self.device, dtype=torch.float16)
        _chk(lib().fie_pack_conv3x3_f16(self.h, _p(w), co, ci, cin_pad, _p(out), kpad, npad))
        return out

    # ------------------------------------------------------------------ ops
    # ---- GroupNorm statistics from the producing GEMM / conv (include/fie.h: fie_gn_stats_target)
    def _gn_stats_arm(self, rows_total, n, rows_per_image, groups):
        """Arms the next launch to write GroupNorm partial sums of its [rows_total, n] output; returns the tag groupnorm() looks for on
        the output tensor, or None when the shape is not eligible (channels per group not 4 / 8 / 16, rows not in 32-row granules)."""
        if not self.gn_from_epilogue or self.f32 or not groups or n % groups:
            return None
        if rows_per_image % 32 or rows_total % rows_per_image:
            return None
        cg, pgroups = n // groups, groups
        if cg not in (4, 8, 16):
            # the UNet's 20 / 40-channel groups: the producer writes one slot per 4-channel quad and the consumer sums the quads of a group.
            # Only where the three-kernel GroupNorm would run (64x64 latents and up): the small maps take the single-pass kernel, one launch
            if cg % 4 or rows_per_image < 4096 or not self.gn_quads:
                return None
            pgroups = n // 4
        b = rows_total // rows_per_image
        need = lib().fie_gn_stats_bytes(b, rows_per_image, pgroups)
        key = (self._stream, self.ws_tag)            # one buffer per stream and graph slot: a producer's sums are consumed before the next producer runs
        buf = self._gn_stats.get(key)
        if buf is None or buf.numel() < need:
            buf = self._gn_stats[key] = torch.empty(need, device=self.device, dtype=torch.uint8)
        if self._keep is not None:
            self._keep.append(buf)
        _chk(lib().fie_gn_stats_target(self.h, _p(buf), rows_per_image, pgroups))
        self._gn_gen += 1
        return (buf, groups, n, rows_per_image, b, self._gn_gen, key, pgroups)

    def gemm(self, a, wp, n, out=None, a2=None, bias=None, rowbias=None, rows_per_batch=0, residual=None, scale=1.0,
             act=ACT_NONE, k=None, gn_stats=None, a_scale=1.0, out_f8=False, out_inv_scale=1.0):
        """a: [M, K1] (last-dim contiguous, row stride free), optional a2: [M, K2]; wp packed weight; n logical N.
        a of dtype uint8 = e4m3 activations (fie_gemm_x8_f16; a_scale = their dequantisation scale); out_f8: the output is e4m3 bytes too."""
        self.sync_stream()
        self._bind_splitk()
        m, k1 = a.shape
        ktot = k1 + (a2.shape[1] if a2 is not None else 0)
        if k is not None:
            assert k == ktot
        nout = n // 2 if act == ACT_GEGLU else n
        if out is None:
            out = self._alloc((m, nout), torch.uint8 if out_f8 else None)
        assert a.stride(1) == 1 and out.stride(1) == 1
        if self.f32:
            _chk(lib().fie_gemm_f32(self.h, _p(a), a.stride(0), k1, _p(a2), a2.stride(0) if a2 is not None else 0, _p(wp),
                                    wp.stride(0), 0, _p(out), out.stride(0), m, n, ktot, _p(bias), _p(rowbias),
                                    rowbias.stride(0) if rowbias is not None else 0, rows_per_batch, _p(residual),
                                    residual.stride(0) if residual is not None else 0, float(scale), act, 1, 1, 0, 0, 0, 0, 0, 0))
            return out
        if a.dtype == torch.uint8:                      # e4m3 activations (written by a producer with out_f8) x e4m3 weights
            assert isinstance(wp, W8) and a2 is None and wp.stride(0) % 128 == 0, "fp8 activations need fp8 weights packed with Kpad % 128 == 0"
            if out_f8 and out.dtype != torch.uint8:
                raise ValueError("out_f8 needs a uint8 output tensor")
            _chk(lib().fie_gemm_x8_f16(self.h, _p(a), a.stride(0), _p(wp.q), wp.stride(0), _p(wp.scale), float(a_scale), _p(out), out.stride(0), m, n, ktot,
                                       _p(bias), _p(rowbias), rowbias.stride(0) if rowbias is not None else 0, rows_per_batch, _p(residual),
                                       residual.stride(0) if residual is not None else 0, float(scale), act, int(out_f8), float(out_inv_scale)))
            return out
        if isinstance(wp, W8):
            _chk(lib().fie_gemm_w8_f16(self.h, _p(a), a.stride(0), k1, _p(a2), a2.stride(0) if a2 is not None else 0,
                                       _p(wp.q), wp.stride(0), _p(wp.scale), _p(out), out.stride(0), m, n, ktot, _p(bias), _p(rowbias),
                                       rowbias.stride(0) if rowbias is not None else 0, rows_per_batch, _p(residual),
                                       residual.stride(0) if residual is not None else 0, float(scale), act))
            return out
        tag = self._gn_stats_arm(m, n, gn_stats[0], gn_stats[1]) if gn_stats and act != ACT_GEGLU and out.stride(0) == n else None
        _chk(lib().fie_gemm_f16(self.h, _p(a), a.stride(0), k1, _p(a2), a2.stride(0) if a2 is not None else 0,
                                _p(wp), wp.stride(0), _p(out), out.stride(0), m, n, ktot, _p(bias), _p(rowbias),
                                rowbias.stride(0) if rowbias is not None else 0, rows_per_batch, _p(residual),
                                residual.stride(0) if residual is not None else 0, float(scale), act))
        out._gn_tag = tag
        return out

    def fold_layernorm(self, w, bias, gamma, beta, geglu=False):
        """Prepares a Linear whose input is a LayerNorm for gemm_ln (include/fie.h: fie_gemm_ln_f16): the packed f16 matrix of W * gamma and the fp32 table
        [(sum_k Wf[n, k], (W beta)[n] + bias[n])] in the packed column order (fold_layernorm_tables below: the algebra, device-free)."""
        dev = lambda t: None if t is None else t.to(self.device)
        wf, tab = fold_layernorm_tables(dev(w), dev(bias), dev(gamma), dev(beta), geglu=geglu)
        return self.pack_linear(wf, geglu=geglu, quant=False), tab

    def gemm_ln(self, x, wp, n, tab, eps=1e-5, act=ACT_NONE, out=None):
        """out = act(LayerNorm(x) @ W^T + bias) with the LayerNorm folded into the GEMM (fold_layernorm prepared wp / tab); x: [M, K] un-normalised."""
        self.sync_stream()
        m, k = x.shape
        nout = n // 2 if act == ACT_GEGLU else n
        if out is None:
            out = self._alloc((m, nout))
        assert x.stride(1) == 1 and out.stride(1) == 1 and tab.dtype == torch.float32 and tab.shape == (n, 2) and not self.f32
        _chk(lib().fie_gemm_ln_f16(self.h, _p(x), x.stride(0), _p(wp), wp.stride(0), _p(tab), float(eps), _p(out), out.stride(0), m, n, k, act))
        out._gn_tag = None
        return out

    def conv3x3_plus(self, x, wp, cout, x2, x3=None, bias=None, rowbias=None, scale=1.0, act=ACT_NONE, gn_groups=None):
        """conv3x3(x) + [x2 | x3] @ W1x1^T in one GEMM (include/fie.h: fie_conv3x3_plus_nhwc_f16); x2 / x3: [B*H*W, C] views, last dim contiguous."""
        self.sync_stream()
        self._bind_splitk()
        b, h, w, cin = x.shape
        assert x.is_contiguous() and not self.f32 and x2.stride(1) == 1 and (x3 is None or x3.stride(1) == 1)
        out = self._alloc((b, h, w, cout))
        tag = self._gn_stats_arm(b * h * w, cout, h * w, gn_groups) if gn_groups else None
        _chk(lib().fie_conv3x3_plus_nhwc_f16(self.h, _p(x), b, h, w, cin, _p(wp), wp.stride(0), _p(out), out.stride(2), cout, _p(bias), _p(rowbias),
                                             rowbias.stride(0) if rowbias is not None else 0, float(scale), act, _p(x2), x2.stride(0), x2.shape[1],
                                             _p(x3), x3.stride(0) if x3 is not None else 0, x3.shape[1] if x3 is not None else 0))
        out._gn_tag = tag
        return out

    def pack_conv_up2x(self, w):
        """OIHW 3x3 weights -> the four parity matrices of fie_conv_up2x_nhwc_f16, [4][Npad][Kpad] f16: for output parity (py, px) the taps
        that fall on the same input pixel are summed (fp32, rounded once): rows {ky 0} | {ky 1, 2} for parity 0, {ky 0, 1} | {ky 2} for 1."""
        w = w.to(self.device, torch.float32)
        sets = (((0,), (1, 2)), ((0, 1), (2,)))
        mats = []
        for py in range(2):
            for px in range(2):
                taps = [sum(w[:, :, ky, kx] for ky in sets[py][a] for kx in sets[px][b]) for a in range(2) for b in range(2)]    # 4 x [Cout, Cin]
                mats.append(self.pack_linear(torch.stack(taps, 1).reshape(w.shape[0], -1).to(torch.float16), quant=False))
        return torch.stack(mats).contiguous()

    def conv_up2x(self, x, wp4, cout, bias=None, rowbias=None, scale=1.0, act=ACT_NONE, gn_groups=None):
        """conv3x3(nearest-2x(x)) from the four parity matrices of pack_conv_up2x: x [B, H, W, Cin] -> [B, 2H, 2W, cout]."""
        self.sync_stream()
        self._bind_splitk()
        b, h, w, cin = x.shape
        assert x.is_contiguous() and not self.f32 and cin % 64 == 0
        out = self._alloc((b, 2 * h, 2 * w, cout))
        tag = self._gn_stats_arm(b * 4 * h * w, cout, 4 * h * w, gn_groups) if gn_groups else None
        _chk(lib().fie_conv_up2x_nhwc_f16(self.h, _p(x), b, h, w, cin, _p(wp4), wp4.stride(1), wp4.shape[1], _p(out), out.stride(2), cout,
                                          _p(bias), _p(rowbias), rowbias.stride(0) if rowbias is not None else 0, float(scale), act))
        out._gn_tag = tag
        return out

    def conv3x3(self, x, wp, cout, out=None, stride=1, pad_mode=0, upsample=False, bias=None, rowbias=None,
                residual=None, scale=1.0, act=ACT_NONE, ldc=None, gn_groups=None, a_scale=1.0):
        """x: [B, H, W, Cin] f16 contiguous NHWC -> [B, OH, OW, ldc]."""
        self.sync_stream()
        self._bind_splitk()
        b, h, w, cin = x.shape
        assert x.is_contiguous()
        hin, win = (h * 2, w * 2) if upsample else (h, w)
        pads = 2 if pad_mode == 0 else 1
        oh, ow = (hin + pads - 3) // stride + 1, (win + pads - 3) // stride + 1
        ldc = ldc or cout
        if out is None:
            out = self._alloc((b, oh, ow, ldc), zero=ldc != cout)
        if x.dtype == torch.uint8:                      # e4m3 activations from a GroupNorm with out_f8 (fie_conv3x3_x8_nhwc_f16)
            assert isinstance(wp, W8) and cin % 128 == 0 and wp.stride(0) % 128 == 0
            tag = self._gn_stats_arm(b * oh * ow, cout, oh * ow, gn_groups) if gn_groups and ldc == cout else None
            _chk(lib().fie_conv3x3_x8_nhwc_f16(self.h, _p(x), b, h, w, cin, int(upsample), stride, pad_mode, _p(wp.q), wp.stride(0), _p(wp.scale), float(a_scale),
                                               _p(out), out.stride(2), cout, _p(bias), _p(rowbias), rowbias.stride(0) if rowbias is not None else 0,
                                               _p(residual), residual.stride(2) if residual is not None else 0, float(scale), act))
            out._gn_tag = tag
            return out
        if isinstance(wp, W8):
            _chk(lib().fie_conv3x3_w8_nhwc_f16(self.h, _p(x), b, h, w, cin, int(upsample), stride, pad_mode, _p(wp.q), wp.stride(0),
                                               _p(wp.scale), _p(out), out.stride(2), cout, _p(bias), _p(rowbias),
                                               rowbias.stride(0) if rowbias is not None else 0, _p(residual),
                                               residual.stride(2) if residual is not None else 0, float(scale), act))
            return out
        fn = lib().fie_conv3x3_nhwc_f32 if self.f32 else lib().fie_conv3x3_nhwc_f16
        tag = self._gn_stats_arm(b * oh * ow, cout, oh * ow, gn_groups) if gn_groups and ldc == cout else None
        _chk(fn(self.h, _p(x), b, h, w, cin, int(upsample), stride, pad_mode, _p(wp),
                                        wp.stride(0), _p(out), out.stride(2), cout, _p(bias), _p(rowbias),
                                        rowbias.stride(0) if rowbias is not None else 0, _p(residual),
                                        residual.stride(2) if residual is not None else 0, float(scale), act))
        out._gn_tag = tag
        return out

    def attention(self, q, k, v, heads, head_dim, tq, tk, batch, out=None, causal=False, scale=None, out_f8=False, out_inv_scale=1.0):
        """q: [B*Tq, >=H*D] view (row stride free); k, v: [B*Tk, ...]; returns [B*Tq, H*D] (out_f8: as e4m3 bytes, value * out_inv_scale)."""
        self.sync_stream()
        if out is None:
            out = self._alloc((batch * tq, heads * head_dim), torch.uint8 if out_f8 else None)
        scale = scale if scale is not None else head_dim ** -0.5
        if self.f32:
            # S = Q K^T (batched over image x head) -> row softmax -> O = P V; the fp32 scores are simply materialised
            s_ = self._alloc((batch * heads, tq, tk), torch.float32)
            d = head_dim
            _chk(lib().fie_gemm_f32(self.h, _p(q), q.stride(0), d, None, 0, _p(k), k.stride(0), 0, _p(s_), tk, tq, tk, d, None, None, 0,
                                    0, None, 0, 1.0, ACT_NONE, batch, heads, tq * q.stride(0), d, tk * k.stride(0), d,
                                    heads * tq * tk, tq * tk))
            _chk(lib().fie_softmax_rows_f32(self.h, _p(s_), batch * heads * tq, tk, tk, float(scale), int(causal), tq))
            _chk(lib().fie_gemm_f32(self.h, _p(s_), tk, tk, None, 0, _p(v), v.stride(0), 1, _p(out), out.stride(0), tq, d, tk, None,
                                    None, 0, 0, None, 0, 1.0, ACT_NONE, batch, heads, heads * tq * tk, tq * tk, tk * v.stride(0), d,
                                    tq * out.stride(0), d))
            return out
        if out_f8:
            _chk(lib().fie_attention_f16_o8(self.h, _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out), out.stride(0), batch, heads, tq, tk,
                                            head_dim, float(scale), int(causal), float(out_inv_scale)))
            return out
        _chk(lib().fie_attention_f16(self.h, _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out),
                                     out.stride(0), batch, heads, tq, tk, head_dim, float(scale), int(causal)))
        return out

    def groupnorm(self, x1, gamma, beta, groups, eps, silu, x2=None, out=None, out_f8=False, out_inv_scale=1.0):
        """x1: [B, rows, C1] (+ x2: [B, rows, C2]) NHWC-flattened, contiguous -> [B, rows, C1+C2] (out_f8: as e4m3 bytes, value * out_inv_scale)."""
        self.sync_stream()
        b, rows, c1 = x1.shape[0], x1[0].numel() // x1.shape[-1], x1.shape[-1]
        c2 = x2.shape[-1] if x2 is not None else 0
        assert x1.is_contiguous() and (x2 is None or x2.is_contiguous())
        if out is None:
            out = self._alloc(x1.shape[:-1] + (c1 + c2,), torch.uint8 if out_f8 else None)
        need = lib().fie_groupnorm_workspace_bytes(b, rows, groups)
        key = (self._stream, self.ws_tag)            # one scratch buffer per stream (and per in-flight graph slot)
        ws = self._gn_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._gn_ws[key] = torch.empty(need, device=self.device, dtype=torch.uint8)
        if self._keep is not None:
            self._keep.append(ws)
        tag = getattr(x1, "_gn_tag", None)
        if (tag is not None and x2 is None and tag[1:5] == (groups, c1, rows, b) and tag[5] == self._gn_gen and tag[6] == key):
            # the producer's epilogue left this tensor's partial sums (and nothing has overwritten them): one read of x instead of two
            if out_f8:
                _chk(lib().fie_groupnorm_stats_nhwc_f16_o8(self.h, _p(x1), c1, _p(out), b, rows, groups, _p(gamma), _p(beta), float(eps), int(silu),
                                                           _p(tag[0]), _p(ws), tag[7], float(out_inv_scale)))
                return out
            _chk(lib().fie_groupnorm_stats_nhwc_f16(self.h, _p(x1), c1, _p(out), b, rows, groups, _p(gamma), _p(beta), float(eps), int(silu),
                                                    _p(tag[0]), _p(ws), tag[7]))
            return out
        if out_f8:
            _chk(lib().fie_groupnorm_nhwc_f16_o8(self.h, _p(x1), c1, _p(x2), c2, _p(out), b, rows, groups, _p(gamma), _p(beta), float(eps), int(silu), _p(ws),
                                                 float(out_inv_scale)))
            return out
        _chk((lib().fie_groupnorm_nhwc_f32 if self.f32 else lib().fie_groupnorm_nhwc_f16)(self.h, _p(x1), c1, _p(x2), c2, _p(out), b, rows, groups, _p(gamma),
                                          _p(beta), float(eps), int(silu), _p(ws)))
        return out

    def conv3x3_gn_ok(self, x, cout, out_groups):
        """True when GroupNorm(+SiLU) -> conv3x3 exists as ONE launch for this input (include/fie.h: fie_conv3x3_gn_ok) and x carries its producer's sums."""
        b, h, w, cin = x.shape
        return bool(self.gn_fuse_conv and not self.f32 and getattr(x, "_gn_tag", None) is not None and out_groups
                    and lib().fie_conv3x3_gn_ok(self.h, b, h, w, cin, cout, out_groups))

    def groupnorm_coef(self, x, gamma, beta, groups, eps):
        """The GroupNorm of x (which carries its producer's partial sums) as per-(image, channel) coefficients [B, C, 2] f32 (fie_groupnorm_coef_f16), or
        None when the sums are not there any more."""
        self.sync_stream()
        b, rows, c = x.shape[0], x[0].numel() // x.shape[-1], x.shape[-1]
        tag = getattr(x, "_gn_tag", None)
        key = (self._stream, self.ws_tag)
        if not (tag is not None and tag[1:5] == (groups, c, rows, b) and tag[5] == self._gn_gen and tag[6] == key):
            return None
        need = lib().fie_groupnorm_workspace_bytes(b, rows, groups)
        ws = self._gn_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._gn_ws[key] = torch.empty(need, device=self.device, dtype=torch.uint8)
        if self._keep is not None:
            self._keep.append(ws)
        coef = self._alloc((b, c, 2), torch.float32)
        _chk(lib().fie_groupnorm_coef_f16(self.h, c, b, rows, groups, _p(gamma), _p(beta), float(eps), _p(tag[0]), _p(ws), tag[7], _p(coef)))
        return coef

    def conv3x3_gn(self, x, coef, silu, wp, cout, bias=None, residual=None, gn_groups=None):
        """conv3x3(silu(x * sc + sh)) in one launch (fie_conv3x3_gn_nhwc_f16); the output's own GroupNorm sums are armed as for conv3x3."""
        self.sync_stream()
        self._bind_splitk()
        b, h, w, cin = x.shape
        assert x.is_contiguous() and coef.dtype == torch.float32 and coef.shape == (b, cin, 2)
        out = self._alloc((b, h, w, cout))
        tag = self._gn_stats_arm(b * h * w, cout, h * w, gn_groups)
        assert tag is not None, "the fused GroupNorm -> conv needs the output's sums armed (conv3x3_gn_ok)"
        _chk(lib().fie_conv3x3_gn_nhwc_f16(self.h, _p(x), b, h, w, cin, _p(coef), int(bool(silu)), _p(wp), wp.stride(0), _p(out), out.stride(2), cout, _p(bias),
                                           _p(residual), residual.stride(2) if residual is not None else 0))
        out._gn_tag = tag
        return out

    def quantize_f8(self, x, inv_scale=1.0):
        """[rows, C] f16 -> e4m3 bytes (value * inv_scale, saturated): the plain conversion (tests; producers without a fused form)."""
        self.sync_stream()
        rows, c = x.shape
        out = self._alloc((rows, c), torch.uint8)
        _chk(lib().fie_quantize_f8(self.h, _p(x), x.stride(0), _p(out), out.stride(0), rows, c, float(inv_scale)))
        return out

    def amax_into(self, x, slot):
        """Folds max |x| of a [..., C] f16 tensor (last dim contiguous, uniform row stride) into the device float `slot` (fp8 calibration)."""
        self.sync_stream()
        x2 = x.reshape(-1, x.shape[-1]) if x.is_contiguous() else x
        assert x2.dim() == 2 and x2.stride(1) == 1 and x2.dtype == torch.float16 and slot.dtype == torch.float32
        _chk(lib().fie_amax_f16(self.h, _p(x2), x2.stride(0), x2.shape[0], x2.shape[1], _p(slot)))

    def layernorm(self, x, gamma, beta, eps=1e-5, out=None, out_f8=False, out_inv_scale=1.0):
        self.sync_stream()
        rows, c = x.shape
        if out_f8:
            out = self._alloc((rows, c), torch.uint8) if out is None else out
            _chk(lib().fie_layernorm_f16_o8(self.h, _p(x), x.stride(0), _p(out), out.stride(0), rows, c, _p(gamma), _p(beta), float(eps), float(out_inv_scale)))
            return out
        if out is None:
            out = self._alloc((rows, c))
        _chk((lib().fie_layernorm_f32 if self.f32 else lib().fie_layernorm_f16)(self.h, _p(x), x.stride(0), _p(out), out.stride(0), rows, c, _p(gamma), _p(beta),
                                     float(eps)))
        return out

    def sinusoid(self, vals, dim, out, col0=0):
        """vals: f32 [B, nvals] on device; writes [cos|sin] blocks of width dim into out[:, col0:]."""
        self.sync_stream()
        b, nv = vals.shape
        _chk((lib().fie_sinusoid_f32 if self.f32 else lib().fie_sinusoid_f16)(self.h, _p(vals), b, nv, dim, _p(out), out.stride(0), col0))
        return out

    def time_embed_workspace(self, e):
        """Zero-filled workspace of the fused timestep-embedding kernel (one per model: launches on one stream reuse it)."""
        return torch.zeros(lib().fie_time_embed_workspace_bytes(e), device=self.device, dtype=torch.uint8)

    def time_embed(self, t, w1, b1, w2, b2, ws, add=None):
        """K7 fused: silu(W2 silu(W1 sinusoid(t) + b1) + b2 + add) for the B <= 4 rows of one step; w1 [E, C0], w2 [E, E] plain f16."""
        self.sync_stream()
        b, (e, c0) = t.numel(), w1.shape
        out = self._alloc((b, e), torch.float16)
        _chk(lib().fie_time_embed_f16(self.h, _p(t), b, c0, e, _p(w1), _p(b1), _p(w2), _p(b2), _p(add),
                                      add.stride(0) if add is not None else 0, _p(out), out.stride(0), _p(ws)))
        return out

    def clip_embed(self, ids, tok, pos):
        self.sync_stream()
        b, t = ids.shape
        c = tok.shape[1]
        out = self._alloc((b * t, c))
        _chk((lib().fie_clip_embed_f32 if self.f32 else lib().fie_clip_embed_f16)(self.h, _p(ids), b, t, c, _p(tok), _p(pos), _p(out)))
        return out

    def pixels_in(self, u8_hwc, normalize, copies=1):
        self.sync_stream()
        h, w, _ = u8_hwc.shape
        out = self._alloc((copies, h, w, 8))
        _chk((lib().fie_pixels_in_u8_f32 if self.f32 else lib().fie_pixels_in_u8_f16)(self.h, _p(u8_hwc), h, w, int(normalize), _p(out), copies))
        return out

    def pixels_out(self, x_nhwc):
        self.sync_stream()
        _, h, w, ld = x_nhwc.shape
        out = self._alloc((h, w, 3), torch.uint8)
        _chk((lib().fie_pixels_out_f32_u8 if self.f32 else lib().fie_pixels_out_f16_u8)(self.h, _p(x_nhwc), ld, h, w, _p(out)))
        return out

    def resize_lanczos(self, rgb_u8, out_h, out_w):
        """u8 [H, W, 3] device tensor -> u8 [out_h, out_w, 3], bit-exact with PIL's `resize(..., Image.LANCZOS)`."""
        from . import resize
        self.sync_stream()
        h, w, _ = rgb_u8.shape
        assert rgb_u8.dtype == torch.uint8 and rgb_u8.is_contiguous()
        tabs = []
        for n_in, n_out in ((w, out_w), (h, out_h)):
            key = (n_in, n_out)
            if n_in != n_out and key not in self._resize_tables:
                kk, bounds, ks = resize.coefficients(n_in, n_out)
                self._resize_tables[key] = (torch.from_numpy(kk).to(self.device), torch.from_numpy(bounds).to(self.device), ks)
            tabs.append(self._resize_tables.get(key) if n_in != n_out else (None, None, 0))
        out = torch.empty((out_h, out_w, 3), device=rgb_u8.device, dtype=torch.uint8)
        tmp = torch.empty((h, out_w, 3), device=rgb_u8.device, dtype=torch.uint8) if (h != out_h and w != out_w) else None
        (kx, bx, ksx), (ky, by, ksy) = tabs
        _chk(lib().fie_resize_rgb_u8(self.h, _p(rgb_u8), h, w, _p(out), out_h, out_w, _p(kx), _p(bx), ksx, _p(ky), _p(by), ksy, _p(tmp)))
        return out

    def canny_device(self, rgb_u8, low=100, high=200):
        """u8 [H, W, 3] device tensor -> u8 [H, W, 3] edge map on the device (integer exact; synchronises the stream)."""
        self.sync_stream()
        h, w, _ = rgb_u8.shape
        assert rgb_u8.is_contiguous() and rgb_u8.dtype == torch.uint8
        ws = torch.empty(lib().fie_canny_workspace_bytes(h, w), device=self.device, dtype=torch.uint8)
        out = torch.empty((h, w, 3), device=self.device, dtype=torch.uint8)
        it = _I(0)
        _chk(lib().fie_canny_rgb_device_u8(self.h, _p(rgb_u8), h, w, int(low), int(high), _p(ws), _p(out), ctypes.byref(it)))
        self.canny_passes = it.value
        return out

    def canny_begin(self, rgb_u8, low=100, high=200, rounds=1):
        """First half of canny_device() (include/fie.h: fie_canny_rgb_device_begin_u8): launches NMS + `rounds` rounds of four hysteresis passes + the
        edge map and returns (edge map tensor, state) without waiting; canny_finish(state) waits and tells whether that map was final."""
        self.sync_stream()
        h, w, _ = rgb_u8.shape
        assert rgb_u8.is_contiguous() and rgb_u8.dtype == torch.uint8
        ws = torch.empty(lib().fie_canny_workspace_bytes(h, w), device=self.device, dtype=torch.uint8)
        out = torch.empty((h, w, 3), device=self.device, dtype=torch.uint8)
        cache = self.__dict__.setdefault("_canny_flags", {})           # pinned flag words, one set per stream (edits in flight run on their own)
        flags = cache.get(self._stream)
        if flags is None:
            flags = cache[self._stream] = torch.zeros(4, dtype=torch.int32).pin_memory()
        _chk(lib().fie_canny_rgb_device_begin_u8(self.h, _p(rgb_u8), h, w, int(low), int(high), int(rounds), _p(ws), _p(out), flags.data_ptr()))
        return out, (h, w, ws, out, flags, rgb_u8, int(rounds))

    def canny_finish(self, state):
        """Waits for canny_begin's work; runs the further hysteresis rounds if its last pass had still changed something (the edge map tensor is then
        rewritten).  Returns the tensor; `self.canny_more` = passes launched here (0: the map was final when begin's kernels ended)."""
        h, w, ws, out, flags, _src, rounds = state
        self.sync_stream()
        it = _I(0)
        _chk(lib().fie_canny_rgb_device_finish_u8(self.h, h, w, _p(ws), _p(out), flags.data_ptr(), ctypes.byref(it)))
        self.canny_more = it.value
        self.canny_passes = 4 * rounds + it.value
        return out

    def latent_prep(self, moments, eps_post, noise, hw, sf, sqrt_ab, sqrt_1mab, latents, model_in):
        self.sync_stream()
        _chk((lib().fie_latent_prep_f32 if self.f32 else lib().fie_latent_prep)(self.h, _p(moments), _p(eps_post), _p(noise), hw, float(sf), float(sqrt_ab),
                                   float(sqrt_1mab), _p(latents), _p(model_in), model_in.shape[0]))

    def lcm_step(self, eps, nb, guidance, latents, noise, hw, sab_t, s1mab_t, c_skip, c_out, sab_p, s1mab_p, model_in,
                 inv_sf, decode_in):
        self.sync_stream()
        _chk((lib().fie_lcm_step_f32 if self.f32 else lib().fie_lcm_step)(self.h, _p(eps), eps.shape[-1], nb, float(guidance), _p(latents), _p(noise), hw,
                                float(sab_t), float(s1mab_t), float(c_skip), float(c_out), float(sab_p),
                                float(s1mab_p), _p(model_in), model_in.shape[0] if model_in is not None else 0,
                                float(inv_sf), _p(decode_in)))


def canny_rgb(rgb_u8, low=100, high=200):
    """Host Canny through the C ABI (numpy uint8 HxWx3 in and out)."""
    import numpy as np
    a = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    out = np.empty_like(a)
    _chk(lib().fie_canny_rgb_u8(a.ctypes.data, a.shape[0], a.shape[1], int(low), int(high), out.ctypes.data))
    return out


_ctx = {}


def context(device=0, dtype=torch.float16):
    key = (device, dtype)
    if key not in _ctx:
        _ctx[key] = Context(device, dtype)
    return _ctx[key]
