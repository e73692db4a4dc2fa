"""AutoencoderKL encode/decode on HIP kernels (upstream models/autoencoders/vae.py).

NHWC fp16; image and latent tensors are stored 8-channel zero-padded so every conv/GEMM operand row is 16-byte
aligned.  The asymmetric (0,1,0,1) pad + stride-2 of the encoder downsamplers and the nearest-2x of the decoder
upsamplers are folded into the conv kernel's addressing; the mid-block attention is the d=512 single-head
instance of the flash kernel.  Stays fp16 end to end (fp16-fix VAE semantics: no upcast)."""
import torch

from . import hip
from .nn import Conv3, Linear, Norm, Resnet, F16


class _MidAttn:
    def __init__(self, ctx, sd, p, c, groups, eps):
        self.c, self.groups, self.eps = c, groups, eps
        self.norm = Norm(ctx, sd, p + "group_norm")
        w = torch.cat([sd[p + f"to_{n}.weight"] for n in "qkv"], 0)
        b = torch.cat([sd[p + f"to_{n}.bias"] for n in "qkv"], 0)
        self.qkv = Linear(ctx, None, None, w=w, b=b)
        self.out = Linear(ctx, sd, p + "to_out.0")

    def __call__(self, ctx, x):
        b, h, w, c = x.shape
        y = ctx.groupnorm(x, self.norm.g, self.norm.b, self.groups, self.eps, False).view(b * h * w, c)
        qkv = self.qkv(ctx, y)
        a = ctx.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], 1, c, h * w, h * w, b)
        o = self.out(ctx, a, residual=x.view(b * h * w, c), gn_stats=(h * w, self.groups))
        y = o.view(b, h, w, c)
        y._gn_tag = getattr(o, "_gn_tag", None)        # the view is a new tensor object: carry the producer's GroupNorm sums along
        return y


class VAE:
    def __init__(self, ctx, cfg, sd):
        self.ctx, self.cfg = ctx, cfg
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        ch, L, lc = cfg["block_out_channels"], cfg["layers_per_block"], cfg["latent_channels"]
        R = lambda p: Resnet(ctx, sd, p, g, eps)
        # encoder
        self.e_in = Conv3(ctx, sd, "encoder.conv_in", cin_pad=8)
        self.e_down = []
        for i in range(len(ch)):
            rs = [R(f"encoder.down_blocks.{i}.resnets.{j}.") for j in range(L)]
            ds = Conv3(ctx, sd, f"encoder.down_blocks.{i}.downsamplers.0.conv") if i != len(ch) - 1 else None
            self.e_down.append((rs, ds))
        self.e_mid = (R("encoder.mid_block.resnets.0."), _MidAttn(ctx, sd, "encoder.mid_block.attentions.0.", ch[-1], g, eps),
                      R("encoder.mid_block.resnets.1."))
        self.e_norm = Norm(ctx, sd, "encoder.conv_norm_out")
        self.e_out = Conv3(ctx, sd, "encoder.conv_out")                   # -> 2*lc = 8 channels
        self.quant = Linear(ctx, sd, "quant_conv")
        # decoder; post_quant_conv sees the 8-channel padded latent: pad its K with zero columns
        wq = sd["post_quant_conv.weight"].reshape(lc, lc)
        wq8 = torch.zeros((8, 8), dtype=wq.dtype, device=wq.device)
        wq8[:lc, :lc] = wq
        bq8 = torch.zeros(8, dtype=wq.dtype, device=wq.device)
        bq8[:lc] = sd["post_quant_conv.bias"]
        self.post_quant = Linear(ctx, None, None, w=wq8, b=bq8)
        self.d_in = Conv3(ctx, sd, "decoder.conv_in", cin_pad=8)
        self.d_mid = (R("decoder.mid_block.resnets.0."), _MidAttn(ctx, sd, "decoder.mid_block.attentions.0.", ch[-1], g, eps),
                      R("decoder.mid_block.resnets.1."))
        rev = list(reversed(ch))
        self.d_up = []
        for i in range(len(rev)):
            rs = [R(f"decoder.up_blocks.{i}.resnets.{j}.") for j in range(L + 1)]
            us = Conv3(ctx, sd, f"decoder.up_blocks.{i}.upsamplers.0.conv") if i != len(rev) - 1 else None
            self.d_up.append((rs, us))
        self.d_norm = Norm(ctx, sd, "decoder.conv_norm_out")
        self.d_out = Conv3(ctx, sd, "decoder.conv_out")                   # 3 channels, written as 4

    def encode_moments(self, x):
        """x: [1, H, W, 8] f16 in [-1, 1] (3 real channels) -> moments [H/8*W/8, 8] (mean | logvar)."""
        ctx, cfg = self.ctx, self.cfg
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        x = self.e_in(ctx, x, gn_groups=g)
        for rs, ds in self.e_down:
            for r in rs:
                x = r(ctx, x)
            if ds is not None:
                x = ds(ctx, x, stride=2, pad_mode=1, gn_groups=g)
        r0, at, r1 = self.e_mid
        x = r1(ctx, at(ctx, r0(ctx, x)))
        x = ctx.groupnorm(x, self.e_norm.g, self.e_norm.b, g, eps, True)
        x = self.e_out(ctx, x)
        b, h, w, c = x.shape
        return self.quant(ctx, x.view(b * h * w, c)), (h, w)

    def decode(self, z):
        """z: [1, h, w, 8] f16 (already divided by the scaling factor) -> [1, 8h, 8w, 4] f16 (3 real channels)."""
        ctx, cfg = self.ctx, self.cfg
        g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        b, h, w, c = z.shape
        x = self.post_quant(ctx, z.view(b * h * w, c)).view(b, h, w, 8)
        x = self.d_in(ctx, x, gn_groups=g)
        r0, at, r1 = self.d_mid
        x = r1(ctx, at(ctx, r0(ctx, x)))
        for rs, us in self.d_up:
            for r in rs:
                x = r(ctx, x)
            if us is not None:
                x = us(ctx, x, upsample=True, gn_groups=g)
        x = ctx.groupnorm(x, self.d_norm.g, self.d_norm.b, g, eps, True)
        return self.d_out(ctx, x)
