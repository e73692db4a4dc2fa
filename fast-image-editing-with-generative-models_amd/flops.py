"""Analytic FLOP model (2*MAC) of the hot path, regenerated from the active presets.
Contractions only (convs, linears, attention matmuls); element-wise / norm work is counted as bytes elsewhere.
`python -c "import fie_amd.flops as f; f.report()"` prints the table quoted in BASELINE.md section 2."""
from .presets import time_embed_dim


def _conv(hw, cin, cout, k=3):
    return 2 * hw * k * k * cin * cout


def _resnet(hw, cin, cout, temb, batch_rows=1):
    f = _conv(hw, cin, cout) + _conv(hw, cout, cout)
    if cin != cout:
        f += _conv(hw, cin, cout, 1)
    if temb:
        f += 2 * temb * cout
    return f


def _tblock(n, c, xdim, text=77):
    lin = 2 * n * c * (3 * c) + 2 * n * c * c            # qkv, out
    lin += 2 * n * c * c + 2 * text * xdim * 2 * c + 2 * n * c * c   # q2, kv2 (text), out2
    lin += 2 * n * c * 8 * c + 2 * n * 4 * c * c         # ff1 (GEGLU 8C), ff2
    self_attn = 4 * n * n * c
    cross = 4 * n * text * c
    return dict(linear=lin, self_attn=self_attn, cross_attn=cross)


def _t2d(n, c, depth, xdim):
    out = dict(linear=4 * n * c * c, self_attn=0, cross_attn=0)
    for _ in range(depth):
        for k, v in _tblock(n, c, xdim).items():
            out[k] += v
    return out


def _acc(total, part):
    for k, v in part.items():
        total[k] = total.get(k, 0) + v


def unet_like_flops(cfg, latent_hw=(128, 128), with_up=True):
    """FLOPs of one forward at batch 1."""
    h, w = latent_hw
    chans = cfg["block_out_channels"]
    te = time_embed_dim(cfg)
    xd = cfg["cross_attention_dim"]
    tot = dict(conv=0, linear=0, self_attn=0, cross_attn=0)
    tot["conv"] += _conv(h * w, cfg["in_channels"], chans[0])
    tot["linear"] += 2 * (chans[0] * te + te * te + cfg["projection_class_embeddings_input_dim"] * te + te * te)
    hw = h * w
    cin = chans[0]
    skips = [(chans[0], hw)]
    for i, cout in enumerate(chans):
        for j in range(cfg["layers_per_block"]):
            tot["conv"] += _resnet(hw, cin, cout, te)
            cin = cout
            d = cfg["down_attn"][i][j]
            if d:
                _acc(tot, _t2d(hw, cout, d, xd))
            skips.append((cout, hw))
        if i != len(chans) - 1:
            hw //= 4
            tot["conv"] += _conv(hw, cout, cout)
            skips.append((cout, hw))
    c = chans[-1]
    tot["conv"] += _resnet(hw, c, c, te)
    for _ in range(1, cfg["mid_resnets"]):
        if cfg["mid_attn"]:
            _acc(tot, _t2d(hw, c, cfg["mid_attn"], xd))
        tot["conv"] += _resnet(hw, c, c, te)
    if cfg["kind"] == "controlnet":
        emb = cfg["conditioning_embedding_out_channels"]
        chw = h * w * 64
        tot["conv"] += _conv(chw, cfg["conditioning_channels"], emb[0])
        for i in range(len(emb) - 1):
            tot["conv"] += _conv(chw, emb[i], emb[i])
            chw //= 4
            tot["conv"] += _conv(chw, emb[i], emb[i + 1])
        tot["conv"] += _conv(chw, emb[-1], chans[0])
        for cc, shw in skips:
            tot["linear"] += _conv(shw, cc, cc, 1)
        tot["linear"] += _conv(hw, c, c, 1)
    elif with_up:
        rev = list(reversed(chans))
        prev = rev[0]
        for i, cout in enumerate(rev):
            for j in range(cfg["layers_per_block"] + 1):
                sc, _ = skips.pop()
                tot["conv"] += _resnet(hw, prev + sc, cout, te)
                prev = cout
                d = cfg["up_attn"][i][j]
                if d:
                    _acc(tot, _t2d(hw, cout, d, xd))
            if i != len(rev) - 1:
                hw *= 4
                tot["conv"] += _conv(hw, cout, cout)
        tot["conv"] += _conv(hw, chans[0], cfg["out_channels"])
    tot["total"] = sum(tot.values())
    return tot


def vae_flops(cfg, image_hw=(1024, 1024)):
    h, w = image_hw
    ch, L, lc = cfg["block_out_channels"], cfg["layers_per_block"], cfg["latent_channels"]

    def mid(hw, c):
        return 2 * _resnet(hw, c, c, 0) + 2 * hw * c * c * 4 + 4 * hw * hw * c

    hw = h * w
    enc = _conv(hw, cfg["in_channels"], ch[0])
    cin = ch[0]
    for i, cout in enumerate(ch):
        for _ in range(L):
            enc += _resnet(hw, cin, cout, 0)
            cin = cout
        if i != len(ch) - 1:
            hw //= 4
            enc += _conv(hw, cout, cout)
    enc += mid(hw, ch[-1]) + _conv(hw, ch[-1], 2 * lc) + _conv(hw, 2 * lc, 2 * lc, 1)
    dec = _conv(hw, lc, lc, 1) + _conv(hw, lc, ch[-1]) + mid(hw, ch[-1])
    rev = list(reversed(ch))
    cin = rev[0]
    for i, cout in enumerate(rev):
        for _ in range(L + 1):
            dec += _resnet(hw, cin, cout, 0)
            cin = cout
        if i != len(rev) - 1:
            hw *= 4
            dec += _conv(hw, cout, cout)
    dec += _conv(hw, ch[0], cfg["out_channels"])
    return dict(encode=enc, decode=dec)


def clip_flops(cfg, tokens=77):
    h, f = cfg["hidden"], cfg["intermediate"]
    per = 2 * tokens * h * 4 * h + 4 * tokens * tokens * h + 2 * tokens * h * f * 2
    return cfg["layers"] * per + 2 * cfg["projection_dim"] * h


def image_flops(cfgs, evals, cfg_batch, image_hw=(1024, 1024)):
    lat = (image_hw[0] // 8, image_hw[1] // 8)
    u = unet_like_flops(cfgs["unet"], lat)["total"]
    c = unet_like_flops(cfgs["controlnet"], lat)["total"]
    v = vae_flops(cfgs["vae"], image_hw)
    t = clip_flops(cfgs["clip_l"]) + clip_flops(cfgs["clip_g"])
    return dict(unet=u, controlnet=c, vae_encode=v["encode"], vae_decode=v["decode"], clip=t,
                total=evals * cfg_batch * (u + c) + v["encode"] + v["decode"] + cfg_batch * t)


def report():
    from . import presets as P
    for c in (P.UNET_SDXL, P.UNET_SSD1B_A1, P.UNET_SSD1B_A, P.UNET_SSD1B_B, P.CONTROLNET_FULL, P.CONTROLNET_SMALL):
        f = unet_like_flops(c)
        print(f"{c['name']:45s} " + "  ".join(f"{k}={v / 1e12:.3f}" for k, v in f.items()))
    v = vae_flops(P.VAE_SDXL)
    print(f"vae encode={v['encode'] / 1e12:.3f} decode={v['decode'] / 1e12:.3f}")
    print(f"clip-l={clip_flops(P.CLIP_L) / 1e12:.4f} bigG={clip_flops(P.CLIP_BIGG) / 1e12:.4f}")
