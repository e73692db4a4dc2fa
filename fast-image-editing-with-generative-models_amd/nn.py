"""Host-side graph sequencing for the UNet / ControlNet: packs diffusers-named weights
into the kernel layouts once, then issues the HIP kernels of csrc/ through the C ABI in graph order.

Activations are NHWC fp16 ([B, H, W, C]; token views [B*H*W, C] share the same memory).  What is fused where:
  * resnet:        GN+SiLU kernel -> conv1 (+bias, +time-embedding row bias) -> GN+SiLU -> conv2 (+bias, +shortcut/identity residual)
  * up-block cat:  never materialised -- GroupNorm and the 1x1 shortcut GEMM read the two halves directly
  * up/downsample: nearest-2x gather / stride 2 folded into the conv's im2col addressing
  * transformer:   fused QKV GEMM; attention reads q/k/v as strided column views; out-proj/FF2 GEMMs add the residual;
                   FF1 is one GEMM with the GEGLU epilogue; cross-attention K/V of the (step-invariant) text are cached
  * time-embedding projections of ALL resnets are one GEMM per step (their weights are concatenated at load)
  * ControlNet zero-convs scale by conditioning_scale and add straight into the UNet skip tensors (GEMM epilogue)
"""
import torch

from . import hip
from .presets import time_embed_dim

F16 = torch.float16


def _dev(ctx, t):
    return t.to(ctx.device, ctx.dtype).contiguous()


class Linear:
    def __init__(self, ctx, sd, name, geglu=False, w=None, b=None, quant=True):
        """quant=False keeps the weight fp16 even while the context packs fp8 weights (`ctx.w8`): the M = batch-rows embedding
        MLPs, whose outputs condition every layer."""
        w = sd[name + ".weight"] if w is None else w
        b = sd.get(name + ".bias") if b is None and name else b
        w = w.reshape(w.shape[0], -1)
        self.n, self.k = w.shape
        self.act = hip.ACT_GEGLU if geglu else hip.ACT_NONE
        self.wp = ctx.pack_linear(w, geglu=geglu, quant=quant)
        if b is not None:
            b = _dev(ctx, b)
            if geglu:
                b = torch.stack([b[: self.n // 2], b[self.n // 2:]], 1).reshape(-1).contiguous()
        self.b = b
        # fp8 activations (csrc/gemm_x8.hip): possible when the weight is fp8 with Kpad % 128 == 0 and K % 16 == 0; the caller hands in e4m3 bytes
        self.a8 = isinstance(self.wp, hip.W8) and ctx.a8 and self.wp.stride(0) % 128 == 0 and self.k % 16 == 0

    def __call__(self, ctx, a, a2=None, residual=None, act=None, scale=1.0, out=None, rowbias=None, rows_per_batch=0, gn_stats=None,
                 a_scale=1.0, out_f8=False, out_inv_scale=1.0):
        return ctx.gemm(a, self.wp, self.n, a2=a2, bias=self.b, residual=residual, scale=scale, out=out,
                        act=self.act if act is None else act, rowbias=rowbias, rows_per_batch=rows_per_batch, gn_stats=gn_stats,
                        a_scale=a_scale, out_f8=out_f8, out_inv_scale=out_inv_scale)


class Conv3:
    def __init__(self, ctx, sd, name, cin_pad=None, cout_pad=None):
        w = sd[name + ".weight"]
        self.cout, self.cin = w.shape[:2]
        self.cin_pad = cin_pad or (self.cin + 7) // 8 * 8
        self.ldc = cout_pad or self.cout
        self.n = (self.cout + 3) // 4 * 4          # kernel writes whole 4-channel groups (extra ones are zeros)
        # fp8-weight models (config 5).  A conv whose input comes from a GroupNorm and has Cin % 128 == 0 can read e4m3 ACTIVATIONS (csrc/gemm_x8.hip,
        # conv view, split-K included).  The other 1280-wide convs of the 32x32-latent level stay fp16: M = 2048 rows fill the chip only with split-K,
        # which the fp16 ring kernels have and the fp8-weight / fp16-activation conv kernel has not (79 vs 109 us for 1280 -> 1280)
        a8_ok = bool(ctx.w8 and ctx.a8 and getattr(ctx, "a8_conv", True) and self.cin_pad == self.cin and self.cin % 128 == 0
                     and name.endswith((".conv1", ".conv2")))                # the resnet convs: their input is a GroupNorm output
        self.wp = ctx.pack_conv3x3(w, self.cin_pad, quant=self.cout < 1280 or a8_ok)
        self.a8 = a8_ok and isinstance(self.wp, hip.W8) and self.wp.stride(0) % 128 == 0      # the caller hands in e4m3 (GroupNorm with out_f8)
        # up-sampler convs (diffusers: "...upsamplers.0.conv") run as four 2x2 parity convs (hip.py: pack_conv_up2x).  Packed HERE, not at
        # the first call: a first call inside a launch-program recording would record the pack launches instead of running them.
        self.wp4 = None
        if ("upsamplers" in name and self.cin % 64 == 0 and self.cin_pad == self.cin and ctx.up2x_parity and not ctx.f32
                and torch.is_tensor(self.wp) and self.ldc <= self.n == self.cout):
            self.wp4 = ctx.pack_conv_up2x(w)
        b = sd.get(name + ".bias")
        if b is not None:
            bb = torch.zeros(self.n, dtype=ctx.dtype, device=ctx.device)
            bb[: self.cout] = b.to(ctx.device, ctx.dtype)
            b = bb
        self.b = b

    def __call__(self, ctx, x, stride=1, pad_mode=0, upsample=False, rowbias=None, residual=None, act=hip.ACT_NONE, gn_groups=None, a_scale=1.0):
        """gn_groups: the output feeds a GroupNorm with that many groups -- have the epilogue leave its partial sums (hip.py: _gn_stats_arm).
        a_scale: the dequantisation scale of e4m3 input activations (1 / the producer's out_inv_scale)."""
        if upsample and self.wp4 is not None and ctx.up2x_parity and residual is None and stride == 1 and pad_mode == 0:
            return ctx.conv_up2x(x, self.wp4, self.n, bias=self.b, rowbias=rowbias, act=act, gn_groups=gn_groups)
        return ctx.conv3x3(x, self.wp, self.n, stride=stride, pad_mode=pad_mode, upsample=upsample, bias=self.b,
                           rowbias=rowbias, residual=residual, act=act, ldc=max(self.ldc, self.n), gn_groups=gn_groups, a_scale=a_scale)


class Norm:
    def __init__(self, ctx, sd, name):
        self.g, self.b = _dev(ctx, sd[name + ".weight"]), _dev(ctx, sd[name + ".bias"])


class Resnet:
    def __init__(self, ctx, sd, p, groups, eps, temb_slot=None):
        self.groups, self.eps = groups, eps
        self.n1, self.n2 = Norm(ctx, sd, p + "norm1"), Norm(ctx, sd, p + "norm2")
        self.c1, self.c2 = Conv3(ctx, sd, p + "conv1"), Conv3(ctx, sd, p + "conv2")
        self.sc = Linear(ctx, sd, p + "conv_shortcut") if p + "conv_shortcut.weight" in sd else None
        self.temb_slot = temb_slot              # (col0, col1) into the fused time-embedding projection
        # fp8 activations: the dequantisation scales of the two GroupNorm outputs the convs read as e4m3 (value = byte * s8[i]); 1 until
        # HipImg2ImgPipeline.calibrate_fp8 measured them (powers of two).  amax: the device floats a calibration pass folds max |x| into
        self.s8, self.amax = [1.0, 1.0], None
        # conv2 + the 1x1 shortcut as ONE GEMM (fie_conv3x3_plus_nhwc_f16): conv2's packed matrix with the shortcut's columns appended
        self.wp_plus = None
        if self.sc is not None and torch.is_tensor(self.c2.wp) and torch.is_tensor(self.sc.wp) and self.c2.wp.dtype == torch.float16:
            k2, ksc = 9 * self.c2.cin_pad, sd[p + "conv_shortcut.weight"].shape[1]
            if self.c2.cin % 64 == 0 and self.c2.cin_pad == self.c2.cin and ksc % 64 == 0 and self.c2.n == self.c2.cout == self.sc.n and self.c2.ldc <= self.c2.n:
                self.wp_plus = torch.cat([self.c2.wp[:, :k2], self.sc.wp[:, :ksc]], 1).contiguous()
                zero = torch.zeros(self.c2.n, dtype=self.c2.wp.dtype, device=self.c2.wp.device)
                self.b_plus = (self.c2.b if self.c2.b is not None else zero) + (self.sc.b if self.sc.b is not None else zero)

    def _gn_conv(self, ctx, x, norm, conv, residual=None):
        """GroupNorm + SiLU + conv3x3 as ONE launch where the halo-resident kernel has that form (include/fie.h: fie_conv3x3_gn_nhwc_f16: one image, 16-aligned
        map, 128..1024 input channels, x carrying its producer's sums): the conv normalises its input in LDS and the apply kernel's read + write of the
        tensor disappear.  None: not built for this conv -- the caller takes the two launches."""
        if not (torch.is_tensor(conv.wp) and conv.wp.dtype == torch.float16 and conv.cin_pad == conv.cin == x.shape[-1] and conv.ldc <= conv.n == conv.cout
                and ctx.conv3x3_gn_ok(x, conv.n, self.groups)):
            return None
        coef = ctx.groupnorm_coef(x, norm.g, norm.b, self.groups, self.eps)
        if coef is None:
            return None
        return ctx.conv3x3_gn(x, coef, True, conv.wp, conv.n, bias=conv.b, residual=residual, gn_groups=self.groups)

    def __call__(self, ctx, x, temb_all=None, skip=None):
        b, h, w, _ = x.shape
        if skip is None and self.temb_slot is None and not ctx.calib:      # the VAE's resnets: both convs may take their GroupNorm along
            y = self._gn_conv(ctx, x, self.n1, self.c1)
            if y is not None:
                if self.sc is None:
                    out = self._gn_conv(ctx, y, self.n2, self.c2, residual=x)
                    if out is not None:
                        return out
                return self._tail(ctx, x, y, skip)
        calib = ctx.calib and self.amax is not None             # calibration pass: f16 activations (the fp8-weight kernels take them), max |x| recorded
        q1, q2 = self.c1.a8 and not calib, self.c2.a8 and self.wp_plus is None and not calib
        y = ctx.groupnorm(x, self.n1.g, self.n1.b, self.groups, self.eps, True, x2=skip, out_f8=q1, out_inv_scale=1.0 / self.s8[0])   # fp8 model: e4m3 for an fp8-activation conv
        if calib and self.c1.a8:
            ctx.amax_into(y, self.amax[0:1])
        rb = temb_all[:, self.temb_slot[0]:self.temb_slot[1]] if self.temb_slot is not None else None
        y = self.c1(ctx, y, rowbias=rb, gn_groups=self.groups, a_scale=self.s8[0] if q1 else 1.0)     # norm2's first pass rides on conv1's epilogue where the group width allows
        return self._tail(ctx, x, y, skip, calib, q2)

    def _tail(self, ctx, x, y, skip, calib=False, q2=False):
        """norm2 -> SiLU -> conv2 (+ shortcut | identity) on conv1's output y."""
        b, h, w, _ = x.shape
        y = ctx.groupnorm(y, self.n2.g, self.n2.b, self.groups, self.eps, True, out_f8=q2, out_inv_scale=1.0 / self.s8[1])
        if calib and self.c2.a8 and self.wp_plus is None:
            ctx.amax_into(y, self.amax[1:2])
        if self.wp_plus is not None and ctx.conv_plus_shortcut and x.shape[-1] % 64 == 0 and (skip is None or skip.shape[-1] % 64 == 0):
            return ctx.conv3x3_plus(y, self.wp_plus, self.c2.n, x.view(b * h * w, -1), None if skip is None else skip.view(b * h * w, -1),
                                    bias=self.b_plus, gn_groups=self.groups)
        if self.sc is not None:
            res = self.sc(ctx, x.view(b * h * w, -1), a2=None if skip is None else skip.view(b * h * w, -1))
            res = res.view(b, h, w, -1)
        else:
            res = x
        return self.c2(ctx, y, residual=res, gn_groups=self.groups, a_scale=self.s8[1] if q2 else 1.0)      # ... and the next block's norm on conv2's


class TBlock:
    def __init__(self, ctx, sd, p, c, head_dim):
        self.c, self.heads, self.hd = c, c // head_dim, head_dim
        self.ln = [Norm(ctx, sd, p + f"norm{i}") for i in (1, 2, 3)]
        wqkv = torch.cat([sd[p + f"attn1.to_{n}.weight"] for n in "qkv"], 0)
        self.qkv = Linear(ctx, None, None, w=wqkv)
        self.o1 = Linear(ctx, sd, p + "attn1.to_out.0")
        self.q2 = Linear(ctx, sd, p + "attn2.to_q")
        self._kv_raw = torch.cat([sd[p + "attn2.to_k.weight"], sd[p + "attn2.to_v.weight"]], 0)
        self.kv2 = Linear(ctx, None, None, w=self._kv_raw)
        # the net's grouped text K/V projection (_CondNet._finish_kv): this block's columns of the one [2 x 77, sum 2c] result; None: the block projects its own
        self.kv_net, self.kv_slot = None, None
        self.o2 = Linear(ctx, sd, p + "attn2.to_out.0")
        self.ff1 = Linear(ctx, sd, p + "ff.net.0.proj", geglu=True)
        self.ff2 = Linear(ctx, sd, p + "ff.net.2")
        self.kv_cache = None
        # LayerNorm folded into its consumer (include/fie.h: fie_gemm_ln_f16): norm1 -> qkv, norm2 -> to_q, norm3 -> GEGLU projection read the UN-normalised
        # stream; (W * gamma, column sums, W beta + bias) prepared here.  f16 weights only (the fp8 configuration's LayerNorm writes e4m3), K % 64 == 0,
        # GEGLU on the 256x320 tile (N = 8 c % 320 == 0).  ctx.ln_fold (FIE_LN_FOLD=0) is the A/B switch: the unfolded matrices stay loaded
        self.folded = None
        if (not ctx.f32 and not ctx.w8 and c % 64 == 0 and (8 * c) % 320 == 0 and all(torch.is_tensor(l.wp) for l in (self.qkv, self.q2, self.ff1))):
            ln = [(sd[p + f"norm{i}.weight"], sd[p + f"norm{i}.bias"]) for i in (1, 2, 3)]
            self.folded = (ctx.fold_layernorm(wqkv, None, *ln[0]),
                           ctx.fold_layernorm(sd[p + "attn2.to_q.weight"], sd.get(p + "attn2.to_q.bias"), *ln[1]),
                           ctx.fold_layernorm(sd[p + "ff.net.0.proj.weight"], sd.get(p + "ff.net.0.proj.bias"), *ln[2], geglu=True))
        # BASELINE config 5: every projection whose input is produced by LayerNorm / attention / the GEGLU epilogue reads e4m3 activations
        self.a8 = all(l.a8 for l in (self.qkv, self.o1, self.q2, self.o2, self.ff1, self.ff2)) and head_dim == 64
        # dequantisation scales of the six e4m3 tensors of the block (LN1, attention 1, LN2, attention 2, LN3, GEGLU output): value = byte * s8[i].
        # 1 until HipImg2ImgPipeline.calibrate_fp8 measured them (powers of two: exact both ways); amax: the calibration pass's device floats
        self.s8, self.amax = [1.0] * 6, None

    def _text_kv(self, ctx, text):
        """Cross-attention K | V of the text, [B * 77, 2 c]: invariant over the denoising steps of one image, so computed once per image -- and for ALL blocks
        of the net in ONE GEMM (M = B * 77 rows against the concatenated to_k / to_v matrices: one pass over the weights at HBM rate instead of 20-36 launches
        of 16 us at 0.04 of the MFMA peak); a block takes its column range of the result."""
        if self.kv_cache is None:
            net = self.kv_net
            if net is not None and ctx.kv_group and not ctx.calib:
                if net._kv_out is None:
                    net._kv_out = net.kv_all(ctx, text)
                self.kv_cache = net._kv_out[:, self.kv_slot[0]:self.kv_slot[1]]
            else:
                self.kv_cache = self.kv2(ctx, text)
        return self.kv_cache

    def _call_a8(self, ctx, h, text, batch, tokens, text_len):
        """The block with fp8 activations: LayerNorm, attention and the GEGLU epilogue WRITE e4m3 (value / s8[i], saturating RNE; s8 = 1 gives the
        values the fp8-weight kernels of round 2 converted per fragment), the six projections run the block-scaled fp8 MFMA with a_scale = s8[i]
        folded into the per-channel weight scale; h, q, k, v stay fp16."""
        c, s = self.c, self.s8
        y = ctx.layernorm(h, self.ln[0].g, self.ln[0].b, out_f8=True, out_inv_scale=1.0 / s[0])
        qkv = self.qkv(ctx, y, a_scale=s[0])
        a = ctx.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], self.heads, self.hd, tokens, tokens, batch, out_f8=True, out_inv_scale=1.0 / s[1])
        h = self.o1(ctx, a, residual=h, a_scale=s[1])
        y = ctx.layernorm(h, self.ln[1].g, self.ln[1].b, out_f8=True, out_inv_scale=1.0 / s[2])
        q = self.q2(ctx, y, a_scale=s[2])
        kv = self._text_kv(ctx, text)
        a = ctx.attention(q, kv[:, :c], kv[:, c:], self.heads, self.hd, tokens, text_len, batch, out_f8=True, out_inv_scale=1.0 / s[3])
        h = self.o2(ctx, a, residual=h, a_scale=s[3])
        y = ctx.layernorm(h, self.ln[2].g, self.ln[2].b, out_f8=True, out_inv_scale=1.0 / s[4])
        f = self.ff1(ctx, y, a_scale=s[4], out_f8=True, out_inv_scale=1.0 / s[5])
        return self.ff2(ctx, f, residual=h, a_scale=s[5])

    def __call__(self, ctx, h, text, batch, tokens, text_len):
        am = self.amax if ctx.calib and self.a8 else None      # calibration pass of an fp8 model: the f16-activation path below, max |x| of the six tensors recorded
        if self.a8 and am is None:
            return self._call_a8(ctx, h, text, batch, tokens, text_len)
        rec = (lambda t, i: ctx.amax_into(t, am[i:i + 1])) if am is not None else (lambda t, i: None)
        c = self.c
        if self.folded is not None and ctx.ln_fold and am is None:
            (wq, tq), (w2, t2), (wf, tf) = self.folded
            which = ctx.ln_fold_which
            qkv = ctx.gemm_ln(h, wq, 3 * c, tq) if "qkv" in which else self.qkv(ctx, ctx.layernorm(h, self.ln[0].g, self.ln[0].b))
            a = ctx.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], self.heads, self.hd, tokens, tokens, batch)
            h = self.o1(ctx, a, residual=h)
            q = ctx.gemm_ln(h, w2, c, t2) if "q2" in which else self.q2(ctx, ctx.layernorm(h, self.ln[1].g, self.ln[1].b))
            kv = self._text_kv(ctx, text)
            a = ctx.attention(q, kv[:, :c], kv[:, c:], self.heads, self.hd, tokens, text_len, batch)
            h = self.o2(ctx, a, residual=h)
            f = ctx.gemm_ln(h, wf, 8 * c, tf, act=hip.ACT_GEGLU) if ctx.ln_fold_ff1 else self.ff1(ctx, ctx.layernorm(h, self.ln[2].g, self.ln[2].b))
            return self.ff2(ctx, f, residual=h)
        y = ctx.layernorm(h, self.ln[0].g, self.ln[0].b)
        rec(y, 0)
        qkv = self.qkv(ctx, y)
        a = ctx.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], self.heads, self.hd, tokens, tokens, batch)
        rec(a, 1)
        h = self.o1(ctx, a, residual=h)
        y = ctx.layernorm(h, self.ln[1].g, self.ln[1].b)
        rec(y, 2)
        q = self.q2(ctx, y)
        kv = self._text_kv(ctx, text)
        a = ctx.attention(q, kv[:, :c], kv[:, c:], self.heads, self.hd, tokens, text_len, batch)
        rec(a, 3)
        h = self.o2(ctx, a, residual=h)
        y = ctx.layernorm(h, self.ln[2].g, self.ln[2].b)
        rec(y, 4)
        f = self.ff1(ctx, y)
        rec(f, 5)
        return self.ff2(ctx, f, residual=h)


class Transformer2D:
    def __init__(self, ctx, sd, p, c, depth, head_dim, groups):
        self.groups = groups
        self.norm = Norm(ctx, sd, p + "norm")
        self.pin, self.pout = Linear(ctx, sd, p + "proj_in"), Linear(ctx, sd, p + "proj_out")
        self.blocks = [TBlock(ctx, sd, f"{p}transformer_blocks.{k}.", c, head_dim) for k in range(depth)]

    def __call__(self, ctx, x, text, text_len):
        b, hh, ww, c = x.shape
        xt = x.view(b * hh * ww, c)
        h = ctx.groupnorm(x, self.norm.g, self.norm.b, self.groups, 1e-6, False).view(b * hh * ww, c)
        h = self.pin(ctx, h)
        for blk in self.blocks:
            h = blk(ctx, h, text, b, hh * ww, text_len)
        o = self.pout(ctx, h, residual=xt, gn_stats=(hh * ww, self.groups))     # the next resnet's norm1 reads this tensor: its sums ride on the epilogue
        y = o.view(b, hh, ww, c)
        y._gn_tag = getattr(o, "_gn_tag", None)            # the view is a new tensor object: carry the producer's GroupNorm sums along
        return y

    def reset(self):
        for blk in self.blocks:
            blk.kv_cache = None


class _CondNet:
    """conv_in + time/text-time embedding + down blocks + mid block: the half shared by UNet and ControlNet."""

    def __init__(self, ctx, cfg, sd):
        self.ctx, self.cfg = ctx, cfg
        g, eps, hd = cfg["norm_num_groups"], cfg["norm_eps"], cfg["head_dim"]
        chans = cfg["block_out_channels"]
        self.temb_names = []                      # resnet prefixes in fused-projection order
        self.conv_in = Conv3(ctx, sd, "conv_in", cin_pad=8)
        self.t1, self.t2 = Linear(ctx, sd, "time_embedding.linear_1", quant=False), Linear(ctx, sd, "time_embedding.linear_2", quant=False)
        self.a1, self.a2 = Linear(ctx, sd, "add_embedding.linear_1", quant=False), Linear(ctx, sd, "add_embedding.linear_2", quant=False)
        # fp16 path: the per-step timestep embedding runs in ONE fused kernel (fie_time_embed_f16) on the plain weights
        te_dim, ch0 = time_embed_dim(cfg), chans[0]
        self.te = None
        if not ctx.f32 and te_dim % 64 == 0 and ch0 % 32 == 0 and ch0 <= 512:
            self.te = tuple(_dev(ctx, sd[k]) for k in ("time_embedding.linear_1.weight", "time_embedding.linear_1.bias",
                                                       "time_embedding.linear_2.weight", "time_embedding.linear_2.bias"))
            self.te_ws = {}                       # per (stream, graph slot): edits in flight must not share the barrier counters
        self.down = []
        for i in range(len(chans)):
            layers = []
            for j in range(cfg["layers_per_block"]):
                r = self._resnet(sd, f"down_blocks.{i}.resnets.{j}.", g, eps)
                d = cfg["down_attn"][i][j]
                t = Transformer2D(ctx, sd, f"down_blocks.{i}.attentions.{j}.", chans[i], d, hd, g) if d else None
                layers.append((r, t))
            ds = Conv3(ctx, sd, f"down_blocks.{i}.downsamplers.0.conv") if i != len(chans) - 1 else None
            self.down.append((layers, ds))
        self.mid = [(self._resnet(sd, "mid_block.resnets.0.", g, eps), None)]
        for k in range(1, cfg["mid_resnets"]):
            t = Transformer2D(ctx, sd, f"mid_block.attentions.{k - 1}.", chans[-1], cfg["mid_attn"], hd, g) \
                if cfg["mid_attn"] else None
            self.mid.append((self._resnet(sd, f"mid_block.resnets.{k}.", g, eps), t))
        self._sd = sd

    def _resnet(self, sd, p, g, eps):
        r = Resnet(self.ctx, sd, p, g, eps)
        self.temb_names.append((p, r))
        return r

    def _finish_temb(self, sd):
        """Concatenate every resnet's time_emb_proj into one [sum Cout, temb] GEMM."""
        ws, bs, col = [], [], 0
        for p, r in self.temb_names:
            w = sd[p + "time_emb_proj.weight"]
            ws.append(w)
            bs.append(sd[p + "time_emb_proj.bias"])
            r.temb_slot = (col, col + w.shape[0])
            col += w.shape[0]
        self.temb_proj = Linear(self.ctx, None, None, w=torch.cat(ws, 0), b=torch.cat(bs, 0), quant=False)
        self._sd = None
        self._finish_kv()

    def _finish_kv(self):
        """The to_k / to_v matrices of every transformer block of the net, concatenated: TBlock._text_kv.  f16 weights only (ctx.kv_group: A/B switch)."""
        ctx = self.ctx
        blocks = [b for t in self.transformers() for b in t.blocks]
        self.kv_all, self._kv_out = None, None
        if len(blocks) > 1 and not ctx.f32 and not ctx.w8 and all(b._kv_raw is not None for b in blocks):
            self.kv_all = Linear(ctx, None, None, w=torch.cat([b._kv_raw for b in blocks], 0))
            off = 0
            for b in blocks:
                b.kv_net, b.kv_slot = self, (off, off + 2 * b.c)
                off += 2 * b.c
        for b in blocks:
            b._kv_raw = None

    def transformers(self):
        for layers, _ in self.down:
            for _, t in layers:
                if t:
                    yield t
        for _, t in self.mid:
            if t:
                yield t

    def begin_image(self, pooled, time_ids):
        """Per-image invariants: the text-time addition embedding; drops the cross-attention K/V caches."""
        ctx, cfg = self.ctx, self.cfg
        b = pooled.shape[0]
        ad = cfg["addition_time_embed_dim"]
        add_in = ctx._alloc((b, cfg["projection_class_embeddings_input_dim"]))
        add_in[:, : pooled.shape[1]] = pooled        # a torch copy: begin_image stays OUTSIDE recorded launch programs (hip.Context.record)
        ctx.sinusoid(time_ids, ad, add_in, col0=pooled.shape[1])
        self.add_emb = self.a2(ctx, self.a1(ctx, add_in, act=hip.ACT_SILU))
        for t in self.transformers():
            t.reset()
        self._kv_out = None

    def time_rowbias(self, t_dev):
        """silu(time_emb + add_emb) -> all resnets' time projections in one GEMM.  t_dev: f32 [B, 1] on device."""
        ctx = self.ctx
        ch0 = self.cfg["block_out_channels"][0]
        if self.te is not None and t_dev.shape[0] <= 4:
            ctx.sync_stream()
            key = (ctx._stream, ctx.ws_tag)       # launches that may run concurrently (other stream, other graph slot) never share the barrier counters
            ws = self.te_ws.get(key)
            if ws is None:
                ws = self.te_ws[key] = ctx.time_embed_workspace(self.te[0].shape[0])
            return self.temb_proj(ctx, ctx.time_embed(t_dev, *self.te, ws, add=self.add_emb))
        s = ctx._alloc((t_dev.shape[0], ch0))
        ctx.sinusoid(t_dev, ch0, s)
        # emb = time_emb + add_emb; resnets consume Linear(SiLU(emb)): add_emb rides in as a per-row bias so the
        # second MLP GEMM's epilogue emits SiLU(emb) directly
        semb = self.t2(ctx, self.t1(ctx, s, act=hip.ACT_SILU), rowbias=self.add_emb, rows_per_batch=1, act=hip.ACT_SILU)
        return self.temb_proj(ctx, semb)

    def encode(self, x, temb_all, text, text_len):
        ctx = self.ctx
        skips = [x]
        for layers, ds in self.down:
            for r, t in layers:
                x = r(ctx, x, temb_all)
                if t:
                    x = t(ctx, x, text, text_len)
                skips.append(x)
            if ds is not None:
                x = ds(ctx, x, stride=2, gn_groups=self.cfg["norm_num_groups"])
                skips.append(x)
        for r, t in self.mid:
            if t:
                x = t(ctx, x, text, text_len)
            x = r(ctx, x, temb_all)
        return skips, x


class UNet(_CondNet):
    def __init__(self, ctx, cfg, sd):
        super().__init__(ctx, cfg, sd)
        g, eps, hd = cfg["norm_num_groups"], cfg["norm_eps"], cfg["head_dim"]
        rev = list(reversed(cfg["block_out_channels"]))
        self.up = []
        for i in range(len(rev)):
            layers = []
            for j in range(cfg["layers_per_block"] + 1):
                r = self._resnet(sd, f"up_blocks.{i}.resnets.{j}.", g, eps)
                d = cfg["up_attn"][i][j]
                t = Transformer2D(ctx, sd, f"up_blocks.{i}.attentions.{j}.", rev[i], d, hd, g) if d else None
                layers.append((r, t))
            us = Conv3(ctx, sd, f"up_blocks.{i}.upsamplers.0.conv") if i != len(rev) - 1 else None
            self.up.append((layers, us))
        self.norm_out = Norm(ctx, sd, "conv_norm_out")
        self.conv_out = Conv3(ctx, sd, "conv_out")
        self._finish_temb(sd)

    def transformers(self):
        yield from super().transformers()
        for layers, _ in self.up:
            for _, t in layers:
                if t:
                    yield t

    def decode(self, x, skips, temb_all, text, text_len):
        ctx, cfg = self.ctx, self.cfg
        skips = list(skips)
        for layers, us in self.up:
            for r, t in layers:
                x = r(ctx, x, temb_all, skip=skips.pop())
                if t:
                    x = t(ctx, x, text, text_len)
            if us is not None:
                x = us(ctx, x, upsample=True)
        y = ctx.groupnorm(x, self.norm_out.g, self.norm_out.b, cfg["norm_num_groups"], cfg["norm_eps"], True)
        return self.conv_out(ctx, y)              # [B, H, W, 4]


class ControlNet(_CondNet):
    def __init__(self, ctx, cfg, sd):
        super().__init__(ctx, cfg, sd)
        p = "controlnet_cond_embedding."
        nb = len(cfg["conditioning_embedding_out_channels"]) - 1
        self.ce_in = Conv3(ctx, sd, p + "conv_in", cin_pad=8)
        self.ce_blocks = [(Conv3(ctx, sd, f"{p}blocks.{2 * i}"), Conv3(ctx, sd, f"{p}blocks.{2 * i + 1}")) for i in range(nb)]
        self.ce_out = Conv3(ctx, sd, p + "conv_out")
        n_skip = 1 + sum(cfg["layers_per_block"] + (1 if i != len(cfg["block_out_channels"]) - 1 else 0)
                         for i in range(len(cfg["block_out_channels"])))
        self.zero = [Linear(ctx, sd, f"controlnet_down_blocks.{i}") for i in range(n_skip)]
        self.zero_mid = Linear(ctx, sd, "controlnet_mid_block")
        self._finish_temb(sd)

    def cond_embedding(self, cond):
        """The edge-map embedding does not depend on the timestep: computed once per image. cond: [B,H,W,8] in [0,1]."""
        ctx = self.ctx
        c = self.ce_in(ctx, cond, act=hip.ACT_SILU)
        for a, b in self.ce_blocks:
            c = a(ctx, c, act=hip.ACT_SILU)
            c = b(ctx, c, stride=2, act=hip.ACT_SILU)
        return self.ce_out(ctx, c)

    def encode_cond(self, x_in, cond_emb, temb_all, text, text_len):
        """ControlNet trunk: sample = conv_in(x) + cond_embedding, then the shared down/mid path."""
        x = self.conv_in(self.ctx, x_in, residual=cond_emb)
        return self.encode(x, temb_all, text, text_len)

    def add_residuals(self, skips, mid, scale, unet_skips, unet_mid):
        """Zero-conv epilogues write `unet_skip + scale * zero_conv(controlnet_skip)` directly (fused scale + add)."""
        ctx = self.ctx
        out = []
        for z, s, u in zip(self.zero, skips, unet_skips):
            b, h, w, c = s.shape
            out.append(z(ctx, s.view(-1, c), scale=scale, residual=u.view(-1, c)).view(b, h, w, c))
        b, h, w, c = mid.shape
        m = self.zero_mid(ctx, mid.view(-1, c), scale=scale, residual=unet_mid.view(-1, c)).view(b, h, w, c)
        return out, m
