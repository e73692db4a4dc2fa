"""The graph-level C-ABI forwards (include/fie.h: fie_*_forward_f16, csrc/graphs.cpp) driven from Python: what a non-Python host would do.

`register_*` hands a model's packed weights to the library under their diffusers parameter names (fie_weights_register; the tensors stay owned
by the Python objects), the `*_forward` functions call the C++ walks with raw tensor pointers.  The C++ walks take the same in-model fusions as
the Python walks (GroupNorm sums from the producing epilogue, conv2 + 1x1 shortcut as one launch, 2x2-parity up-samplers); what only the
product path (pipe.py) has is what needs state across calls or crosses two entries (the text K/V cached over the steps, the one-launch
timestep embedding, zero-conv epilogues adding into the UNet's skips, the two-stream fork).  tests/test_cabi_graphs_gpu.py pins the C++ walks
against the Python ones."""
import ctypes

import torch

from . import hip
from .presets import time_embed_dim


class _Reg:
    def __init__(self, ctx, prefix):
        self.ctx, self.prefix, self.keep = ctx, prefix, []

    def raw(self, name, t, n, ld):
        assert torch.is_tensor(t) and t.dtype == torch.float16 and t.is_cuda, f"{name}: the C++ walks take f16 weights only"
        self.keep.append(t)
        hip._chk(hip.lib().fie_weights_register(self.ctx.h, (self.prefix + name).encode(), t.data_ptr(), int(n), int(ld)))

    def vec(self, name, t):
        self.raw(name, t, t.numel(), 0)

    def lin(self, name, l):
        self.raw(name + ".weight", l.wp, l.n, l.wp.stride(0))
        if l.b is not None:
            self.vec(name + ".bias", l.b)

    def conv(self, name, c):
        self.raw(name + ".weight", c.wp, c.cout, c.wp.stride(0))
        if c.b is not None:
            self.vec(name + ".bias", c.b)
        if getattr(c, "wp4", None) is not None and self.ctx.up2x_parity:          # an up-sampler's four 2x2 parity matrices: the walk takes that form
            self.raw(name + ".weight4", c.wp4, c.wp4.shape[1], c.wp4.stride(1))

    def norm(self, name, nm):
        self.vec(name + ".weight", nm.g)
        self.vec(name + ".bias", nm.b)

    def resnet(self, p, r):
        self.norm(p + "norm1", r.n1); self.conv(p + "conv1", r.c1); self.norm(p + "norm2", r.n2); self.conv(p + "conv2", r.c2)
        if r.sc is not None:
            self.lin(p + "conv_shortcut", r.sc)
        if r.wp_plus is not None and self.ctx.conv_plus_shortcut:                  # conv2 + 1x1 shortcut as one launch
            self.raw(p + "conv2_plus.weight", r.wp_plus, r.c2.n, r.wp_plus.stride(0))
            self.vec(p + "conv2_plus.bias", r.b_plus)


def _keep(obj, reg):
    obj._cabi_keep = getattr(obj, "_cabi_keep", []) + reg.keep


# ---------------------------------------------------------------------------------------------------------------- VAE
def register_vae(vae, prefix=""):
    """Encoder + decoder of an fie_amd.vae.VAE under "<prefix>encoder...." / "decoder...." / "quant_conv" / "post_quant_conv"; vae_encode / vae_decode
    below pass the same prefix in the config."""
    vae._cabi_prefix = prefix
    r = _Reg(vae.ctx, prefix)
    r.conv("encoder.conv_in", vae.e_in)
    for i, (rs, ds) in enumerate(vae.e_down):
        for j, rn in enumerate(rs):
            r.resnet(f"encoder.down_blocks.{i}.resnets.{j}.", rn)
        if ds is not None:
            r.conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", ds)
    for side, mid in (("encoder", vae.e_mid), ("decoder", vae.d_mid)):
        r0, at, r1 = mid
        r.resnet(f"{side}.mid_block.resnets.0.", r0)
        r.resnet(f"{side}.mid_block.resnets.1.", r1)
        r.norm(f"{side}.mid_block.attentions.0.group_norm", at.norm)
        r.lin(f"{side}.mid_block.attentions.0.to_qkv", at.qkv)
        r.lin(f"{side}.mid_block.attentions.0.to_out.0", at.out)
    r.norm("encoder.conv_norm_out", vae.e_norm)
    r.conv("encoder.conv_out", vae.e_out)
    r.lin("quant_conv", vae.quant)
    r.lin("post_quant_conv", vae.post_quant)
    r.conv("decoder.conv_in", vae.d_in)
    for i, (rs, us) in enumerate(vae.d_up):
        for j, rn in enumerate(rs):
            r.resnet(f"decoder.up_blocks.{i}.resnets.{j}.", rn)
        if us is not None:
            r.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", us)
    r.norm("decoder.conv_norm_out", vae.d_norm)
    r.conv("decoder.conv_out", vae.d_out)
    _keep(vae, r)


def vae_config(cfg, h, w, prefix=""):
    ch = cfg["block_out_channels"]
    return hip.VaeConfig(h, w, len(ch), (ctypes.c_int * 8)(*ch), cfg["layers_per_block"], cfg["norm_num_groups"], cfg["norm_eps"], cfg["out_channels"],
                         prefix.encode())


def vae_encode(vae, x):
    """x: [1, H, W, 8] f16 -> moments [H/8 * W/8, 8] through fie_vae_encode_f16."""
    ctx = vae.ctx
    ctx.sync_stream()
    ctx._bind_splitk()              # the walk's GEMMs / convs split K on the ctx-bound workspace: this stream's own (include/fie.h)
    _, hh, ww, _ = x.shape
    down = 2 ** (len(vae.cfg["block_out_channels"]) - 1)
    vc = vae_config(vae.cfg, hh // down, ww // down, getattr(vae, "_cabi_prefix", ""))
    need = hip.lib().fie_vae_encode_workspace_bytes(ctypes.byref(vc))
    assert need > 0
    ws = torch.empty(need, device=x.device, dtype=torch.uint8)
    out = torch.zeros(((hh // down) * (ww // down), 8), device=x.device, dtype=torch.float16)
    hip._chk(hip.lib().fie_vae_encode_f16(ctx.h, ctypes.byref(vc), x.data_ptr(), out.data_ptr(), ws.data_ptr(), need))
    return out


def vae_decode(vae, z):
    """z: [1, h, w, 8] f16 -> [1, 8h, 8w, 4] f16 through fie_vae_decode_f16."""
    ctx = vae.ctx
    ctx.sync_stream()
    ctx._bind_splitk()              # the walk's GEMMs / convs split K on the ctx-bound workspace: this stream's own (include/fie.h)
    _, h, w, _ = z.shape
    vc = vae_config(vae.cfg, h, w, getattr(vae, "_cabi_prefix", ""))
    need = hip.lib().fie_vae_decode_workspace_bytes(ctypes.byref(vc), h, w)
    assert need > 0
    ws = torch.empty(need, device=z.device, dtype=torch.uint8)
    up = 2 ** (len(vae.cfg["block_out_channels"]) - 1)
    out = torch.empty((1, h * up, w * up, 4), device=z.device, dtype=torch.float16)      # conv_out writes all four channels (the fourth: zeros)
    hip._chk(hip.lib().fie_vae_decode_f16(ctx.h, ctypes.byref(vc), z.data_ptr(), out.data_ptr(), ws.data_ptr(), need))
    return out


# ---------------------------------------------------------------------------------------------------------------- CLIP
def register_clip(clip, prefix):
    r = _Reg(clip.ctx, prefix)
    r.vec("text_model.embeddings.token_embedding.weight", clip.tok)
    r.vec("text_model.embeddings.position_embedding.weight", clip.pos)
    for i, L in enumerate(clip.layers):
        p = f"text_model.encoder.layers.{i}."
        r.norm(p + "layer_norm1", L["ln1"]); r.lin(p + "self_attn.qkv_proj", L["qkv"]); r.lin(p + "self_attn.out_proj", L["out"])
        r.norm(p + "layer_norm2", L["ln2"]); r.lin(p + "mlp.fc1", L["fc1"]); r.lin(p + "mlp.fc2", L["fc2"])
    r.norm("text_model.final_layer_norm", clip.final_ln)
    if clip.proj is not None:
        r.lin("text_projection", clip.proj)
    r.vec("zero_row", clip.zero_row)
    _keep(clip, r)


def clip_forward(clip, prefix, ids, eos_rows=None):
    """ids: int32 [B, T] on the device; eos_rows: int32 [B] row indices (needed with a projection) -> (penultimate [B*T, C], pooled [B, P] | None)."""
    ctx, cfg = clip.ctx, clip.cfg
    ctx.sync_stream()
    ctx._bind_splitk()              # the walk's GEMMs / convs split K on the ctx-bound workspace: this stream's own (include/fie.h)
    b, t = ids.shape
    cc = hip.ClipConfig(b, t, cfg["hidden"], cfg["heads"], cfg["layers"], cfg["intermediate"], cfg["projection_dim"] or 0,
                        1 if cfg["act"] == "quick_gelu" else 0, cfg["eps"])
    need = hip.lib().fie_clip_text_workspace_bytes(ctypes.byref(cc))
    assert need > 0
    ws = torch.empty(need, device=ids.device, dtype=torch.uint8)
    pen = torch.empty((b * t, cfg["hidden"]), device=ids.device, dtype=torch.float16)
    pooled = torch.empty((b, cfg["projection_dim"]), device=ids.device, dtype=torch.float16) if cfg["projection_dim"] else None
    hip._chk(hip.lib().fie_clip_text_forward_f16(ctx.h, ctypes.byref(cc), prefix.encode(), ids.data_ptr(), eos_rows.data_ptr() if eos_rows is not None else None,
                                                 pen.data_ptr(), pooled.data_ptr() if pooled is not None else None, ws.data_ptr(), need))
    return pen, pooled


# ---------------------------------------------------------------------------------------------------------------- UNet / ControlNet
def _register_cond_half(r, net):
    r.conv("conv_in", net.conv_in)
    for n, l in (("time_embedding.linear_1", net.t1), ("time_embedding.linear_2", net.t2), ("add_embedding.linear_1", net.a1), ("add_embedding.linear_2", net.a2)):
        r.lin(n, l)
    r.lin("time_emb_proj_all", net.temb_proj)

    def transformer(p, t):
        r.norm(p + "norm", t.norm); r.lin(p + "proj_in", t.pin); r.lin(p + "proj_out", t.pout)
        for k, blk in enumerate(t.blocks):
            q = f"{p}transformer_blocks.{k}."
            for i in (1, 2, 3):
                r.norm(q + f"norm{i}", blk.ln[i - 1])
            r.lin(q + "attn1.to_qkv", blk.qkv); r.lin(q + "attn1.to_out.0", blk.o1); r.lin(q + "attn2.to_q", blk.q2); r.lin(q + "attn2.to_kv", blk.kv2)
            r.lin(q + "attn2.to_out.0", blk.o2); r.lin(q + "ff.net.0.proj", blk.ff1); r.lin(q + "ff.net.2", blk.ff2)

    for i, (layers, ds) in enumerate(net.down):
        for j, (rn, t) in enumerate(layers):
            r.resnet(f"down_blocks.{i}.resnets.{j}.", rn)
            if t:
                transformer(f"down_blocks.{i}.attentions.{j}.", t)
        if ds is not None:
            r.conv(f"down_blocks.{i}.downsamplers.0.conv", ds)
    for k, (rn, t) in enumerate(net.mid):
        r.resnet(f"mid_block.resnets.{k}.", rn)
        if t:
            transformer(f"mid_block.attentions.{k - 1}.", t)
    return transformer


def register_unet(unet, prefix="unet."):
    r = _Reg(unet.ctx, prefix)
    transformer = _register_cond_half(r, unet)
    for i, (layers, us) in enumerate(unet.up):
        for j, (rn, t) in enumerate(layers):
            r.resnet(f"up_blocks.{i}.resnets.{j}.", rn)
            if t:
                transformer(f"up_blocks.{i}.attentions.{j}.", t)
        if us is not None:
            r.conv(f"up_blocks.{i}.upsamplers.0.conv", us)
    r.norm("conv_norm_out", unet.norm_out)
    r.conv("conv_out", unet.conv_out)
    _keep(unet, r)


def register_controlnet(cn, prefix="controlnet."):
    r = _Reg(cn.ctx, prefix)
    _register_cond_half(r, cn)
    p = "controlnet_cond_embedding."
    r.conv(p + "conv_in", cn.ce_in)
    for i, (a, b) in enumerate(cn.ce_blocks):
        r.conv(f"{p}blocks.{2 * i}", a)
        r.conv(f"{p}blocks.{2 * i + 1}", b)
    r.conv(p + "conv_out", cn.ce_out)
    for i, z in enumerate(cn.zero):
        r.lin(f"controlnet_down_blocks.{i}", z)
    r.lin("controlnet_mid_block", cn.zero_mid)
    _keep(cn, r)


def unet_config(cfg, batch, h, w, text_len):
    ch = list(cfg["block_out_channels"])
    n = len(ch)
    assert time_embed_dim(cfg) == 4 * ch[0]
    c = hip.UnetConfig()
    c.batch, c.latent_h, c.latent_w, c.text_len, c.num_blocks, c.layers_per_block = batch, h, w, text_len, n, cfg["layers_per_block"]
    for i in range(n):
        c.block_out_channels[i] = ch[i]
        for j, d in enumerate(cfg["down_attn"][i]):
            c.down_attn[i][j] = int(d)
        if "up_attn" in cfg:
            for j, d in enumerate(cfg["up_attn"][i]):
                c.up_attn[i][j] = int(d)
    c.mid_attn, c.mid_resnets, c.head_dim, c.norm_num_groups, c.norm_eps = int(cfg["mid_attn"]), int(cfg["mid_resnets"]), cfg["head_dim"], cfg["norm_num_groups"], cfg["norm_eps"]
    c.cross_attention_dim, c.addition_time_embed_dim = cfg["cross_attention_dim"], cfg["addition_time_embed_dim"]
    c.pooled_dim = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
    cc = cfg.get("conditioning_embedding_out_channels")
    if cc:
        c.num_cond_channels = len(cc)
        for i, v in enumerate(cc):
            c.cond_channels[i] = int(v)
    return c


def skip_shapes(cfg, batch, h, w):
    """[B, H, W, C] of every skip tensor, in order (= the ControlNet's down residuals)."""
    ch = list(cfg["block_out_channels"])
    out = [(batch, h, w, ch[0])]
    for i in range(len(ch)):
        for _ in range(cfg["layers_per_block"]):
            out.append((batch, h, w, ch[i]))
        if i != len(ch) - 1:
            h, w = h // 2, w // 2
            out.append((batch, h, w, ch[i]))
    return out, (batch, h, w, ch[-1])


def controlnet_forward(cn, prefix, x, t, text, pooled, time_ids, cond, scale):
    """x [B, h, w, 8] f16, t f32 [B], text [B*T, X] f16, pooled [B, P] f16, time_ids f32 [B, 6], cond [B, 8h, 8w, 8] f16 -> (down residuals, mid residual)."""
    ctx = cn.ctx
    ctx.sync_stream()
    ctx._bind_splitk()              # the walk's GEMMs / convs split K on the ctx-bound workspace: this stream's own (include/fie.h)
    b, h, w, _ = x.shape
    uc = unet_config(cn.cfg, b, h, w, text.shape[0] // b)
    need = hip.lib().fie_controlnet_workspace_bytes(ctypes.byref(uc))
    assert need > 0, "fie_controlnet_workspace_bytes refused the config"
    ws = torch.empty(need, device=x.device, dtype=torch.uint8)
    shapes, mid_shape = skip_shapes(cn.cfg, b, h, w)
    assert len(shapes) == hip.lib().fie_unet_num_residuals(ctypes.byref(uc))
    downs = [torch.empty(s, device=x.device, dtype=torch.float16) for s in shapes]
    mid = torch.empty(mid_shape, device=x.device, dtype=torch.float16)
    ptrs = (ctypes.c_void_p * len(downs))(*[d.data_ptr() for d in downs])
    hip._chk(hip.lib().fie_controlnet_forward_f16(ctx.h, ctypes.byref(uc), prefix.encode(), x.data_ptr(), t.data_ptr(), text.data_ptr(), pooled.data_ptr(),
                                                  time_ids.data_ptr(), cond.data_ptr(), float(scale), ptrs, mid.data_ptr(), ws.data_ptr(), need))
    return downs, mid


def unet_forward(unet, prefix, x, t, text, pooled, time_ids, down_residuals=None, mid_residual=None):
    """-> eps [B, h, w, 4] f16 through fie_unet_forward_f16."""
    ctx = unet.ctx
    ctx.sync_stream()
    ctx._bind_splitk()              # the walk's GEMMs / convs split K on the ctx-bound workspace: this stream's own (include/fie.h)
    b, h, w, _ = x.shape
    uc = unet_config(unet.cfg, b, h, w, text.shape[0] // b)
    need = hip.lib().fie_unet_workspace_bytes(ctypes.byref(uc))
    assert need > 0, "fie_unet_workspace_bytes refused the config"
    ws = torch.empty(need, device=x.device, dtype=torch.uint8)
    out = torch.empty((b, h, w, 4), device=x.device, dtype=torch.float16)
    ptrs = None
    if down_residuals is not None:
        ptrs = (ctypes.c_void_p * len(down_residuals))(*[d.data_ptr() for d in down_residuals])
    hip._chk(hip.lib().fie_unet_forward_f16(ctx.h, ctypes.byref(uc), prefix.encode(), x.data_ptr(), t.data_ptr(), text.data_ptr(), pooled.data_ptr(),
                                            time_ids.data_ptr(), ptrs, mid_residual.data_ptr() if mid_residual is not None else None, out.data_ptr(),
                                            ws.data_ptr(), need))
    return out


def step_cache_begin(net, prefix, batch, h, w, text_len, controlnet):
    """A new image for the model under `prefix`: bind (first time / larger shape) or reset its step cache (include/fie.h: fie_step_cache_bind) -- the
    next forward fills it with the text K / V of every transformer block (ControlNet: and the conditioning embedding), later ones read it."""
    uc = unet_config(net.cfg, batch, h, w, text_len)
    need = hip.lib().fie_unet_step_cache_bytes(ctypes.byref(uc), int(controlnet))
    assert need > 0, "fie_unet_step_cache_bytes refused the config"
    buf = getattr(net, "_cabi_step_cache", None)
    if buf is None or buf.numel() < need:
        buf = net._cabi_step_cache = torch.empty(need, device=net.ctx.device, dtype=torch.uint8)
    hip._chk(hip.lib().fie_step_cache_bind(net.ctx.h, prefix.encode(), buf.data_ptr(), buf.numel()))     # a (re)bind marks it empty


def step_cache_end(net, prefix):
    """Unbind: forwards under `prefix` compute everything per call again (another model, another text)."""
    hip._chk(hip.lib().fie_step_cache_bind(net.ctx.h, prefix.encode(), None, 0))


# ---------------------------------------------------------------------------------------------------------------- a whole edit
def register_pipeline(pipe):
    """Every model of an fie_amd.pipe.HipImg2ImgPipeline under the prefixes of a diffusers pipeline directory ("text_encoder.", "unet.", ...), behind the
    pipeline's own prefix (pipe.weight_prefix: several pipelines share one context's registry; the product path has registered its VAE and text encoders
    there already)."""
    pre = getattr(pipe, "weight_prefix", "")
    if not getattr(pipe, "cpp_walks", False):
        register_clip(pipe.clip_l, pre + "text_encoder.")
        register_clip(pipe.clip_g, pre + "text_encoder_2.")
        register_vae(pipe.vae, pre)
    register_unet(pipe.unet, pre + "unet.")
    register_controlnet(pipe.controlnet, pre + "controlnet.")


@torch.no_grad()
def run_edit(pipe, job, step_cache=True):
    """The device side of one edit (pipe.run_device's job, one image) with EVERY model call going through a C-ABI forward: the call sequence a
    non-Python host would issue for the pipeline call at /root/reference/src/pipeline.py:261-272.  Returns the u8 HWC image on the device."""
    ctx = pipe.ctx
    dev = ctx.device
    h, w = job["hw"]
    nb, steps = job["nb"], job["steps"]
    lh, lw = h // 8, w // 8
    hw = lh * lw
    pre = getattr(pipe, "weight_prefix", "")
    pl, _ = clip_forward(pipe.clip_l, pre + "text_encoder.", job["ids_l"])
    pg, pooled = clip_forward(pipe.clip_g, pre + "text_encoder_2.", job["ids_g"], job["eos_rows"])
    text = torch.cat([pl, pg], dim=1)
    moments = vae_encode(pipe.vae, ctx.pixels_in(job["img_u8"], True))
    sf = pipe.cfgs["vae"]["scaling_factor"]
    latents = torch.empty((hw, 4), device=dev, dtype=torch.float32)
    model_in = torch.empty((nb, lh, lw, 8), device=dev, dtype=ctx.dtype)
    ctx.latent_prep(moments, job["noises"][0], job["noises"][1], hw, sf, steps[0]["sqrt_ab"], steps[0]["sqrt_1mab"], latents, model_in)
    cond = ctx.pixels_in(job["ctl_u8"], False).repeat(nb, 1, 1, 1).contiguous()       # upstream runs the ControlNet on the duplicated control image
    decode_in = torch.empty((1, lh, lw, 8), device=dev, dtype=ctx.dtype)
    if step_cache:                      # what does not change over the steps (text K / V, conditioning embedding) is computed by the first forward only
        step_cache_begin(pipe.unet, pre + "unet.", nb, lh, lw, text.shape[0] // nb, False)
        step_cache_begin(pipe.controlnet, pre + "controlnet.", nb, lh, lw, text.shape[0] // nb, True)
    next_noise = 2
    for st, t_dev in zip(steps, job["t_dev"]):
        downs, mid = controlnet_forward(pipe.controlnet, pre + "controlnet.", model_in, t_dev, text, pooled, job["time_ids"], cond, job["cn_scale"])
        eps = unet_forward(pipe.unet, pre + "unet.", model_in, t_dev, text, pooled, job["time_ids"], downs, mid)
        z = None if st["last"] else job["noises"][next_noise]
        ctx.lcm_step(eps, nb, job["guidance"], latents, z, hw, st["sqrt_ab"], st["sqrt_1mab"], st["c_skip"], st["c_out"], st["sqrt_ab_prev"],
                     st["sqrt_1mab_prev"], model_in, 1.0 / sf, decode_in)
        next_noise += 1
    if step_cache:
        step_cache_end(pipe.unet, pre + "unet.")
        step_cache_end(pipe.controlnet, pre + "controlnet.")
    return ctx.pixels_out(vae_decode(pipe.vae, decode_in))
