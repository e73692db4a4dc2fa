"""CLIP text encoders on HIP kernels (upstream transformers modeling_clip.py).

Pre-LN transformer, causal 77x77 attention (head dim 64 in both CLIP-L and OpenCLIP-bigG), returns what
encode_prompt() consumes: hidden_states[-2] (penultimate layer, before the final LayerNorm) and -- for the
encoder with a projection -- the projected final-LN state at the EOS position."""
import torch

from . import hip
from .nn import Linear, Norm, F16, _dev


def eos_positions(ids, eos_token_id):
    """Column of the pooled token per row (modeling_clip.py:561-582 of the local transformers): legacy configs with
    eos_token_id == 2 (as shipped with SDXL's text encoders) take argmax(ids) -- <|endoftext|> is the largest id of the CLIP
    vocabulary --, newer ones the first occurrence of eos_token_id."""
    if eos_token_id == 2:
        return ids.argmax(dim=-1)
    return (ids == eos_token_id).int().argmax(dim=-1)


class ClipText:
    def __init__(self, ctx, cfg, sd):
        self.ctx, self.cfg = ctx, cfg
        self.tok = _dev(ctx, sd["text_model.embeddings.token_embedding.weight"])
        self.pos = _dev(ctx, sd["text_model.embeddings.position_embedding.weight"])
        self.layers = []
        for i in range(cfg["layers"]):
            p = f"text_model.encoder.layers.{i}."
            w = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
            b = torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
            self.layers.append(dict(
                ln1=Norm(ctx, sd, p + "layer_norm1"), qkv=Linear(ctx, None, None, w=w, b=b),
                out=Linear(ctx, sd, p + "self_attn.out_proj"), ln2=Norm(ctx, sd, p + "layer_norm2"),
                fc1=Linear(ctx, sd, p + "mlp.fc1"), fc2=Linear(ctx, sd, p + "mlp.fc2")))
        self.final_ln = Norm(ctx, sd, "text_model.final_layer_norm")
        self.proj = Linear(ctx, None, None, w=sd["text_projection.weight"]) if cfg["projection_dim"] else None
        self.act = hip.ACT_QUICK_GELU if cfg["act"] == "quick_gelu" else hip.ACT_GELU
        self.zero_row = torch.zeros((1, cfg["hidden"]), device=ctx.device, dtype=ctx.dtype)     # "position table" of the EOS-row gather

    def __call__(self, ids, eos_rows=None):
        """ids: int tensor [B, T] (host or device).  Returns (penultimate [B*T, C] f16, pooled [B, P] f16 or None)."""
        ctx, cfg = self.ctx, self.cfg
        b, t = ids.shape
        c, heads = cfg["hidden"], cfg["heads"]
        ids_dev = ids.to(ctx.device, torch.int32).contiguous()
        if ctx._keep is not None:
            ctx._keep.append(ids_dev)             # (a converted copy: new token ids must then be written into ids_dev's source)
        x = ctx.clip_embed(ids_dev, self.tok, self.pos)
        n_run = cfg["layers"] if self.proj is not None else cfg["layers"] - 1   # last layer only feeds the pooled output
        penult = None
        for i in range(n_run):
            if i == cfg["layers"] - 1:
                penult = x
            L = self.layers[i]
            y = ctx.layernorm(x, L["ln1"].g, L["ln1"].b, cfg["eps"])
            qkv = L["qkv"](ctx, y)
            a = ctx.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], heads, c // heads, t, t, b, causal=True)
            x = L["out"](ctx, a, residual=x)
            y = ctx.layernorm(x, L["ln2"].g, L["ln2"].b, cfg["eps"])
            x = L["fc2"](ctx, L["fc1"](ctx, y, act=self.act), residual=x)
        if penult is None:
            penult = x
        pooled = None
        if self.proj is not None:
            last = ctx.layernorm(x, self.final_ln.g, self.final_ln.b, cfg["eps"])
            if eos_rows is None:                                                  # first EOS (host index logic)
                eos = eos_positions(ids.to("cpu"), cfg["eos_token_id"])
                eos_rows = (torch.arange(b) * t + eos).to(ctx.device, torch.int32)
            # EOS rows gathered by the embedding-gather kernel (table = the final-LN states, one "position" of zeros): no torch
            # kernel inside the graph, so the whole encoder is a launch program (fie_clip_text_forward)
            idx = eos_rows if eos_rows.dtype == torch.int32 else eos_rows.to(torch.int32)
            if ctx._keep is not None:
                ctx._keep.append(idx)             # a launch program reads the indices by raw pointer: keep a converted copy alive
            pooled = self.proj(ctx, ctx.clip_embed(idx.view(-1, 1), last, self.zero_row))
        return penult, pooled
