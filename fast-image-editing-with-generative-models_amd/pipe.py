"""The object `FastEditor` stores in ``self.pipe``: same call signature as the diffusers
StableDiffusionXLControlNetImg2ImgPipeline call at /root/reference/src/pipeline.py:261-272, but every stage
(CLIP text, VAE encode, ControlNet + UNet evaluations, CFG + LCM step, VAE decode, pixel conversion) runs in the
hand-written HIP kernels of csrc/ (call order: SURVEY.md 3.2 steps 1-9)."""
import itertools
import os
import threading
import types

import numpy as np
import torch
from PIL import Image

from . import cabi, hip
from .clip import ClipText, eos_positions
from .lcm import LCMSchedule
from .nn import ControlNet, UNet
from .presets import LCM_SCHED
from .tokenizer import StandInTokenizer
from .vae import VAE

F16 = torch.float16
_PIPE_SEQ = itertools.count()


class HipImg2ImgPipeline:
    def __init__(self, ctx, cfgs, sds, tokenizers=None, sched_cfg=None, noise_dtype=None, weight_dtype="f16"):
        """cfgs / sds: dicts with keys unet, controlnet, vae, clip_l, clip_g (configs / diffusers-named state dicts).
        weight_dtype "f8e4m3" (BASELINE config 5): the Linear / conv weights of the UNet and the ControlNet are stored as fp8 e4m3
        with per-output-channel scales (csrc/gemm_w8.hip); the 1280-wide 3x3 convs, VAE, text encoders, embedding MLPs and the few convs
        the LDS-DMA kernels cannot address (Cin % 64 != 0) stay fp16.  ACTIVATIONS: the transformer-block projections and the resnet convs read e4m3
        written by LayerNorm / attention / the GEGLU epilogue / GroupNorm (value / s, saturating at +-448, NaN kept) and run the block-scaled fp8
        MFMA (csrc/gemm_x8.hip) with s folded into the weight scale; every other fp8-weight GEMM / conv rounds its fp16 activations to e4m3 per
        fragment in registers.  So this configuration is W8A8 (not "fp8 weights only").  The per-tensor scales s are 1 until calibrate_fp8() has
        measured them on one edit (powers of two; load_fp8_scales() takes a stored set): on real checkpoints, activations beyond +-448 clip
        without that step.  Parity: SSIM vs the fp16 pipeline and an evaluation against the oracle on the dequantised weights
        (tests/test_fp8_gpu.py, tests/test_sdxl_gpu.py)."""
        if weight_dtype not in ("f16", "f8e4m3"):
            raise ValueError(f"weight_dtype {weight_dtype!r}: 'f16' or 'f8e4m3'")
        if weight_dtype != "f16" and ctx.f32:
            raise ValueError("fp8 weights belong to the fp16 path")
        self.ctx, self.cfgs, self.weight_dtype = ctx, cfgs, weight_dtype
        ctx.w8 = weight_dtype == "f8e4m3"
        try:
            self.unet = UNet(ctx, cfgs["unet"], sds["unet"])
            self.controlnet = ControlNet(ctx, cfgs["controlnet"], sds["controlnet"])
        finally:
            ctx.w8 = False
        self.vae = VAE(ctx, cfgs["vae"], sds["vae"])
        self.clip_l = ClipText(ctx, cfgs["clip_l"], sds["clip_l"])
        self.clip_g = ClipText(ctx, cfgs["clip_g"], sds["clip_g"])
        # The VAE and the two text encoders run through the C++ graph walks of the library (include/fie.h: fie_vae_encode_f16, fie_vae_decode_f16,
        # fie_clip_text_forward_f16; csrc/graphs.cpp) -- bit-identical to the Python walks of vae.py / clip.py (tests/test_cabi_graphs_gpu.py), which stay as
        # the weight packers and the A/B (FIE_CPP_WALKS=0).  Weights are registered under a prefix of this pipeline's own: contexts are shared.
        self.weight_prefix = f"pipe{next(_PIPE_SEQ)}."
        self.cpp_walks = (not ctx.f32) and ctx.device.type == "cuda" and os.environ.get("FIE_CPP_WALKS", "1") != "0"
        if self.cpp_walks:
            cabi.register_vae(self.vae, self.weight_prefix)
            cabi.register_clip(self.clip_l, self.weight_prefix + "text_encoder.")
            cabi.register_clip(self.clip_g, self.weight_prefix + "text_encoder_2.")
        self.tok_l, self.tok_g = tokenizers or (StandInTokenizer(cfgs["clip_l"]["pad_token_id"]),
                                                StandInTokenizer(cfgs["clip_g"]["pad_token_id"]))
        self.scheduler = LCMSchedule(**(sched_cfg or LCM_SCHED))
        self.noise_dtype = noise_dtype or ctx.dtype        # upstream randn_tensor draws in the pipeline dtype
        self.progress = {}
        self.last_stats = {}
        self.timing = None         # set to [] to collect per-stage HIP-event timings in run_device()
        self.use_graph = os.environ.get("FIE_NO_GRAPH", "0") != "1"
        # GEMM / conv tile selection by measurement: the eager warm-up in front of every capture meets each shape once
        # (include/fie.h: fie_gemm_autotune); FIE_AUTOTUNE=0 keeps the built-in rule
        self.autotune = ctx.device.type == "cuda" and os.environ.get("FIE_AUTOTUNE", "1") != "0"
        ctx.autotune(2 if self.autotune else 0)     # shapes are tuned during the eager warm-up of a capture only (_capture)
        # hipGraph cache, bounded: key = (size, CFG batch, step plan, guidance, control scale, slot, images); at most MAX_GRAPHS
        # keys are captured (insertion-ordered dict, least recently used first); a key beyond the cap runs EAGERLY instead of
        # evicting -- destroying a graph and capturing another one was measured to give a slow graph (121 vs 81 ms), and a
        # parameter sweep must not grow HBM without bound.  All graphs of a slot share one capture memory pool (a slot
        # replays one graph at a time and its output is consumed before the next replay), so k keys cost one pool, not k.
        self._graphs = {}
        self._pools = {}
        self.max_graphs = int(os.environ.get("FIE_MAX_GRAPHS", "12"))
        self.eager_overflow = 0     # calls served eagerly because the cache was full
        self._job_consts = {}       # (size, CFG batch, step plan) -> device tensors every job of that shape shares (time ids, timesteps)
        self.repeated_jobs = 0      # device jobs run a second time because a post_check said their inputs had changed (see __call__)
        self._side = None
        self._slot_streams = {}
        self._n_forked = 0
        self._fork_override = None
        self._host_out = {}
        self.eager_lock = threading.RLock()
        self.fork_streams = os.environ.get("FIE_NO_FORK", "0") != "1"
        # Which hardware queue a stream lands on depends on the order in which streams are first used, and two edits in flight
        # only overlap when their streams (and their graphs' branch streams) sit on different queues: creating the slot streams
        # HERE, before any graph, was measured to lose the whole in-flight gain (12.5 instead of 14.4 images/s).  They are
        # created lazily instead, and calibrate_streams() below re-draws streams when a measured pair does not overlap.
        if ctx.device.type == "cuda" and os.environ.get("FIE_EAGER_STREAMS") == "1":    # reproducer of a bad queue mapping (tests of calibrate_streams)
            warm = [self.slot_stream(0), self.slot_stream(1)] + ([self._side_stream()] if self.fork_streams else [])
            for st in warm:
                with torch.cuda.stream(st):
                    torch.zeros(1, device=ctx.device)
            torch.cuda.synchronize(ctx.device)

    def __del__(self):
        try:                       # the registry holds raw pointers into this pipeline's weight tensors: forget them with the pipeline
            if getattr(self, "cpp_walks", False) and self.ctx.h:
                hip.lib().fie_weights_clear_prefix(self.ctx.h, self.weight_prefix.encode())
        except Exception:
            pass

    # -- diffusers API surface the reference touches
    def set_progress_bar_config(self, **kw):          # run_batch.py:157-158
        self.progress.update(kw)

    def _randn(self, shape, generator):
        """diffusers utils/torch_utils.py::randn_tensor: a CPU generator draws on the host and the result is moved;
        a device generator draws on the device.  Drawn in `noise_dtype` (the pipeline dtype upstream)."""
        gdev = generator.device.type if generator is not None else self.ctx.device.type
        if gdev == "cpu":
            x = torch.randn(shape, generator=generator, device="cpu", dtype=self.noise_dtype).to(self.ctx.device)
        else:
            x = torch.randn(shape, generator=generator, device=self.ctx.device, dtype=self.noise_dtype)
        return x.float().contiguous()

    def encode_prompt(self, texts):
        """texts: list of strings -> ([len*77, Dl+Dg] f16, pooled [len, P] f16)."""
        pl, _ = self.clip_l(self.tok_l(texts))
        pg, pooled = self.clip_g(self.tok_g(texts))
        return torch.cat([pl, pg], dim=1), pooled

    def prepare(self, prompt, negative_prompt="", image=None, control_image=None, strength=0.8,
                num_inference_steps=4, guidance_scale=1.5, controlnet_conditioning_scale=0.5, generator=None):
        """`image` / `control_image`: PIL images, or u8 [H, W, 3] tensors already on the device (FastEditor.edit keeps the
        resized source and its device-side Canny map in HBM instead of bouncing them through PIL)."""
        return self._prepare(prompt, negative_prompt, image, control_image, strength, num_inference_steps, guidance_scale,
                             controlnet_conditioning_scale, generator)

    def _prepare(self, prompt, negative_prompt, image, control_image, strength, num_inference_steps, guidance_scale,
                 controlnet_conditioning_scale, generator):
        """Host side of one call: argument checks, tokenisation, RNG draws (in upstream order: posterior sample, init
        noise, one per non-final step) and the H2D copies.  Returns the device-resident job for run_device()."""
        ctx = self.ctx
        if image is None or control_image is None:
            raise ValueError("`image` and `control_image` are both required")
        if strength < 0 or strength > 1:
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        size_of = lambda im: (im.shape[1], im.shape[0]) if torch.is_tensor(im) else im.size
        if size_of(image) != size_of(control_image):
            raise ValueError("image and control_image must have the same size")
        w, h = size_of(image)
        if h % 8 or w % 8:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {h} and {w}.")
        steps = self.scheduler.plan(num_inference_steps, strength)
        if not steps:
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of "
                             f"pipeline steps is 0 which is < 1 and not appropriate for this pipeline.")
        do_cfg = guidance_scale > 1.0
        dev = ctx.device
        texts = [negative_prompt or "", prompt] if do_cfg else [prompt]
        lh, lw = h // 8, w // 8
        u8 = lambda im: im.to(dev).contiguous() if torch.is_tensor(im) else torch.from_numpy(np.array(im.convert("RGB"))).to(dev)
        n_noise = 2 + sum(1 for st in steps if not st["last"])
        nb = 2 if do_cfg else 1
        ids_g = self.tok_g(texts)
        eos = eos_positions(ids_g, self.cfgs["clip_g"]["eos_token_id"])                 # pooled-token column per row
        # the three prompt-dependent int32 tensors travel in ONE host-to-device copy (each pageable copy is a host-synchronous ~40 us); what depends on
        # the size and the step plan only (time ids, timesteps) is uploaded once per (size, plan) and shared by every later job
        ids_l = self.tok_l(texts).to(torch.int32)
        ids_g32 = ids_g.to(torch.int32)
        eos_rows = (torch.arange(nb) * ids_g.shape[1] + eos).to(torch.int32)
        packed = torch.cat([ids_l.reshape(-1), ids_g32.reshape(-1), eos_rows.reshape(-1)]).to(dev)
        n_l, n_g = ids_l.numel(), ids_g32.numel()
        ckey = (h, w, nb, tuple(float(st["t"]) for st in steps), str(dev))
        const = self._job_consts.get(ckey)
        if const is None:
            if len(self._job_consts) > 64:
                self._job_consts.clear()
            const = self._job_consts[ckey] = (torch.tensor([[h, w, 0, 0, h, w]], dtype=torch.float32).repeat(nb, 1).to(dev),
                                              [torch.full((nb, 1), float(st["t"]), dtype=torch.float32).to(dev) for st in steps])
        return dict(
            ids_l=packed[:n_l].view(ids_l.shape), ids_g=packed[n_l:n_l + n_g].view(ids_g32.shape), eos_rows=packed[n_l + n_g:],
            img_u8=u8(image), ctl_u8=u8(control_image), hw=(h, w), steps=steps, nb=nb,
            guidance=float(guidance_scale), cn_scale=float(controlnet_conditioning_scale),
            time_ids=const[0], t_dev=const[1],
            noises=[self._randn((1, 4, lh, lw), generator) for _ in range(n_noise)])

    def prepare_batch(self, prompts, negative_prompts, images, control_images, strength=0.8, num_inference_steps=4,
                      guidance_scale=1.5, controlnet_conditioning_scale=0.5, generators=None):
        """[additive] n independent edits as ONE device job (BASELINE config "batch=8"): the UNet / ControlNet / CLIP run
        at batch n * nb, the VAE per image.  Rows are image-major ([img0 uncond, img0 cond, img1 uncond, ...]); each
        image keeps its own generator, so image i of a batch draws exactly the noise a single call with that generator
        draws (upstream: a list of generators, one per prompt)."""
        n = len(prompts)
        if not (n == len(images) == len(control_images)) or n == 0:
            raise ValueError("prompts, images and control_images must be non-empty lists of one length")
        negative_prompts = negative_prompts or [""] * n
        generators = generators or [None] * n
        jobs = [self._prepare(prompts[i], negative_prompts[i], images[i], control_images[i], strength, num_inference_steps,
                              guidance_scale, controlnet_conditioning_scale, generators[i]) for i in range(n)]
        if any(j["hw"] != jobs[0]["hw"] for j in jobs):
            raise ValueError("all images of a batch must have one size")
        if n == 1:
            return jobs[0]                                # the single-image job (and its graph)
        nb, t77 = jobs[0]["nb"], jobs[0]["ids_g"].shape[1]
        job = dict(jobs[0])
        job["n"] = n
        for k in ("ids_l", "ids_g", "time_ids"):
            job[k] = torch.cat([j[k] for j in jobs], dim=0)
        job["eos_rows"] = torch.cat([j["eos_rows"] + i * nb * t77 for i, j in enumerate(jobs)])
        job["img_u8"] = torch.stack([j["img_u8"] for j in jobs])
        job["ctl_u8"] = torch.stack([j["ctl_u8"] for j in jobs])
        job["t_dev"] = [t.repeat(n, 1) for t in jobs[0]["t_dev"]]
        job["noises"] = [z for j in jobs for z in j["noises"]]          # image-major: image i owns [i*k, (i+1)*k)
        return job

    def _side_stream(self):
        """Second stream for the independent branches of one edit (CLIP beside the VAE encode, UNet encoder beside the
        ControlNet trunk).  `fork_streams = False` keeps the whole edit on one stream.  Overlap between the branches
        (and between edits in flight) needs the streams to land on different hardware queues: see GPU_MAX_HW_QUEUES in
        fie_amd.py."""
        if not (self.fork_streams if self._fork_override is None else self._fork_override):
            return torch.cuda.current_stream(self.ctx.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.ctx.device)
        return self._side

    def _mark(self, name):
        self.ctx.oplog_mark(name)
        if self.timing is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.timing.append((name, ev))

    @torch.no_grad()
    def run_device(self, job):
        """Device side: everything between the H2D and D2H copies.  Returns the u8 HWC image on the device."""
        ctx, dev = self.ctx, self.ctx.device
        h, w = job["hw"]
        nb, steps = job["nb"], job["steps"]
        lh, lw = h // 8, w // 8
        hw = lh * lw
        self._mark("start")
        main = torch.cuda.current_stream(dev)
        side = self._side_stream()
        # 1-2. text encoders (negative prompt "" is really encoded: SURVEY 0 item 4).  Independent of the VAE encode:
        # issued on a second HIP stream so the latency-bound 77-token GEMMs hide under the VAE's large convs.
        n = job.get("n", 1)                               # images in this job (prepare_batch); rows are image-major
        imgs = job["img_u8"] if n > 1 else job["img_u8"][None]
        ctls = job["ctl_u8"] if n > 1 else job["ctl_u8"][None]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if self.cpp_walks:
                pl, _ = cabi.clip_forward(self.clip_l, self.weight_prefix + "text_encoder.", job["ids_l"])
                pg, pooled = cabi.clip_forward(self.clip_g, self.weight_prefix + "text_encoder_2.", job["ids_g"], job["eos_rows"])
            else:
                pl, _ = self.clip_l(job["ids_l"])
                pg, pooled = self.clip_g(job["ids_g"], eos_rows=job["eos_rows"])
            text = torch.cat([pl, pg], dim=1)
            # 6. per-image invariants: nothing here depends on the latents, so the whole block rides on the side stream beside the VAE
            # encode (text-time addition embeddings, the ControlNet's edge-map embedding, the first step's timestep projections)
            self.unet.begin_image(pooled, job["time_ids"])
            self.controlnet.begin_image(pooled, job["time_ids"])
            # the edge-map embedding is the same for the CFG rows of an image: computed once per image, then repeated (upstream runs
            # it on the duplicated batch; the values are identical)
            conds = [ctx.pixels_in(ctls[i], False) for i in range(n)]
            cond_emb = self.controlnet.cond_embedding(conds[0] if n == 1 else torch.cat(conds, dim=0))
            if nb > 1:
                cond_emb = cond_emb.repeat_interleave(nb, dim=0)
            tb0 = (self.unet.time_rowbias(job["t_dev"][0]), self.controlnet.time_rowbias(job["t_dev"][0]))
        text_len = text.shape[0] // (n * nb)
        # 3. pixels, 5. prepare_latents: VAE posterior sample (draw #1), init noise (draw #2), add_noise -- per image (the
        # 1024^2 VAE tensors of a batch would cross the 2 GiB operand limit of the buffer-load kernels, and gain nothing)
        per = len(job["noises"]) // n                     # noise tensors per image
        latents = torch.empty((n, hw, 4), device=dev, dtype=torch.float32)
        model_in = torch.empty((n * nb, lh, lw, 8), device=dev, dtype=ctx.dtype)
        sf = self.cfgs["vae"]["scaling_factor"]
        for i in range(n):
            x_img = ctx.pixels_in(imgs[i], True)
            moments = cabi.vae_encode(self.vae, x_img) if self.cpp_walks else self.vae.encode_moments(x_img)[0]
            ctx.latent_prep(moments, job["noises"][i * per], job["noises"][i * per + 1], hw, sf, steps[0]["sqrt_ab"],
                            steps[0]["sqrt_1mab"], latents[i], model_in[i * nb:(i + 1) * nb])
        next_noise = 2
        main.wait_stream(side)
        self._mark("clip+vae_encode")
        decode_in = torch.empty((n, lh, lw, 8), device=dev, dtype=ctx.dtype)
        self._mark("cond_embed")
        # 7. denoising loop
        for k, (st, t_dev) in enumerate(zip(steps, job["t_dev"])):
            tb_u, tb_c = tb0 if k == 0 else (self.unet.time_rowbias(t_dev), self.controlnet.time_rowbias(t_dev))
            self._mark("embed")
            # UNet encoder and ControlNet trunk are independent until the zero-conv adds: two HIP streams, so their
            # small-grid kernels (32x32 latent level: <= 1 block per CU each) share the 256 CUs
            side.wait_stream(main)
            with torch.cuda.stream(side):
                c_skips, c_mid = self.controlnet.encode_cond(model_in, cond_emb, tb_c, text, text_len)
            skips, mid = self.unet.encode(self.unet.conv_in(ctx, model_in), tb_u, text, text_len)
            main.wait_stream(side)
            self._mark("unet_enc+controlnet")
            skips, mid = self.controlnet.add_residuals(c_skips, c_mid, job["cn_scale"], skips, mid)
            eps = self.unet.decode(mid, skips, tb_u, text, text_len)
            self._mark("unet_dec")
            for i in range(n):
                z = None if st["last"] else job["noises"][i * per + next_noise]
                ctx.lcm_step(eps[i * nb:(i + 1) * nb], nb, job["guidance"], latents[i], z, hw, st["sqrt_ab"], st["sqrt_1mab"],
                             st["c_skip"], st["c_out"], st["sqrt_ab_prev"], st["sqrt_1mab_prev"], model_in[i * nb:(i + 1) * nb],
                             1.0 / sf, decode_in[i:i + 1])
            next_noise += 1
            self._mark("lcm_step")
        # 8-9. decode + postprocess
        dec = (lambda z: cabi.vae_decode(self.vae, z)) if self.cpp_walks else self.vae.decode
        outs = [ctx.pixels_out(dec(decode_in[i:i + 1])) for i in range(n)]
        out_u8 = outs[0] if n == 1 else torch.stack(outs)
        self._mark("vae_decode")
        self.last_stats = dict(unet_evals=len(steps), cfg_batch=nb, latent_hw=(lh, lw), images=n)
        self._latents = latents
        job["_result"] = dict(stats=dict(self.last_stats), latents=latents)     # what THIS job produced (graph entries keep theirs)
        return out_u8

    def _run_eager(self, job):
        """run_device() with the tuner meeting new shapes (as the eager warm-up of a capture does): an eager call and a graph replay of
        the same job then use the same kernels.  That matters since round 3: a split-K choice changes the fp32 summation order, so
        unlike the tile choice it can move the last bit of an f16 output (eager == replay is asserted bit for bit by the tests)."""
        with self.eager_lock:
            self.ctx.autotune(self._tune_mode())
            try:
                return self.run_device(job)
            finally:
                self.ctx.autotune(2 if self.autotune else 0)

    # ------------------------------------------------------------------------------------------------ fp8 activation scales (config 5)
    def _fp8_layers(self):
        """(name, layer) of every layer that reads e4m3 activations: the transformer blocks (six tensors each) and the fp8-activation resnets (two)."""
        out = []
        for mname, net in (("unet", self.unet), ("controlnet", self.controlnet)):
            for i, t in enumerate(net.transformers()):
                out += [(f"{mname}.transformer{i}.block{k}", b) for k, b in enumerate(t.blocks) if b.a8]
            out += [(mname + "." + p.rstrip("."), r) for p, r in net.temb_names if r.c1.a8 or (r.c2.a8 and r.wp_plus is None)]
        return out

    def calibrate_fp8(self, prompt, image, control_image, margin=2.0, **edit_kw):
        """One-pass activation-scale calibration of the fp8 configuration (weight_dtype="f8e4m3"; include/fie.h: fie_amax_f16).  Runs ONE eager edit
        with the fp8-activation layers on f16 activations, folding max |x| of every tensor they would have quantised into a device float; then
        sets, per tensor, the power-of-two scale s = 2^ceil(log2(amax * margin / 448)) -- the producer writes value / s as e4m3, the consumer folds
        s into its per-channel weight scale.  A tensor beyond +-448 is no longer clipped, a small one uses the upper binades of e4m3 instead of its
        subnormals.  Captured graphs hold the old scales and are dropped.  Returns {layer name: [scales]} (load_fp8_scales takes it back)."""
        import math
        if self.weight_dtype != "f8e4m3":
            raise ValueError("calibrate_fp8: the pipeline was not built with weight_dtype='f8e4m3'")
        layers = self._fp8_layers()
        for _, l in layers:
            l.amax = torch.zeros(len(l.s8), device=self.ctx.device, dtype=torch.float32)
        graph, self.use_graph, self.ctx.calib = self.use_graph, False, True
        try:
            self(prompt=prompt, image=image, control_image=control_image, **edit_kw)
        finally:
            self.use_graph, self.ctx.calib = graph, False
        torch.cuda.synchronize(self.ctx.device)
        scales = {}
        for name, l in layers:
            am = l.amax.cpu().tolist()
            l.amax = None
            l.s8 = [2.0 ** math.ceil(math.log2(a * margin / 448.0)) if a > 0 else 1.0 for a in am]
            scales[name] = list(l.s8)
        self._drop_graphs()
        return scales

    def _drop_graphs(self):
        """Captured graphs hold the scales (and kernels) of their capture: forget them AND their memory pools (a pool whose graphs are all gone cannot
        take a new capture: torch's allocator asserts on its use count)."""
        self._graphs.clear()
        self._pools.clear()
        self._n_forked = 0

    def load_fp8_scales(self, scales):
        """Scales from an earlier calibrate_fp8 of the same model (a JSON-able dict); unknown / missing layer names raise."""
        layers = dict(self._fp8_layers())
        if set(scales) != set(layers):
            raise ValueError(f"load_fp8_scales: layer names differ ({len(set(scales) ^ set(layers))} unmatched)")
        for name, sc in scales.items():
            if len(sc) != len(layers[name].s8) or any(not (v > 0) for v in sc):
                raise ValueError(f"load_fp8_scales: bad scales for {name}")
            layers[name].s8 = [float(v) for v in sc]
        self._drop_graphs()

    def _tune_mode(self):
        """fie_gemm_autotune mode of a pass that may meet new shapes: 1 (time them), or 2 (remembered choices only) when the context's choices
        are frozen to a loaded table (FIE_TUNE_TABLE + FIE_TUNE_FROZEN=1: the test session), 0 with FIE_AUTOTUNE=0."""
        return 0 if not self.autotune else 2 if self.ctx.tune_frozen else 1

    MAX_FORKED_GRAPHS = 6

    _TENSOR_KEYS = ("ids_l", "ids_g", "eos_rows", "img_u8", "ctl_u8", "time_ids")

    def run_device_graphed(self, job, slot=0):
        """run_device() replayed from a hipGraph: the ~2 500 launches of one edit are captured once per
        (size, CFG batch, step plan, scales) and replayed with the job's inputs copied into the graph's static buffers.
        The returned u8 image is the graph's static output buffer (consume it before the next replay).
        `slot` selects an independent graph instance (own static buffers / scratch) so that several edits can be in
        flight on different streams of one GPU."""
        base = (job["hw"], job["nb"], tuple(st["t"] for st in job["steps"]), job["guidance"], job["cn_scale"], slot, job.get("n", 1))
        # A forked graph owns extra runtime streams; past ~8 such graphs in one process new ones start sharing hardware queues
        # with their own launch stream and replay 50 % slower (measured: 82 -> 125 ms, tools/edit_ab.py).  Beyond the budget a
        # new key is captured on one stream instead (87 ms): slower than a healthy forked graph, never pathological.
        fork = self.fork_streams and ((base, True) in self._graphs or self._n_forked < self.MAX_FORKED_GRAPHS)
        key = (base, fork)
        entry = self._graphs.get(key)
        if entry is None:
            with self.eager_lock:                       # eager launches + capture go through the one C-ABI context
                entry = self._graphs.get(key)
                if entry is None:
                    if len(self._graphs) >= self.max_graphs:
                        self.eager_overflow += 1
                        # cache full: serve this parameter set eagerly (see __init__) -- with the SLOT's workspaces (GroupNorm scratch,
                        # split-K slabs, the timestep-embedding barrier counters are keyed by (stream, ws_tag)): another slot's graph may
                        # be replaying on its own stream meanwhile
                        self.ctx.ws_tag = slot
                        try:
                            return self._run_eager(job)
                        finally:
                            self.ctx.ws_tag = 0
                    self._fork_override = fork           # read by _side_stream() during this capture only
                    try:
                        entry = self._capture(key, job, slot)
                    finally:
                        self._fork_override = None
                    self._n_forked += int(fork)
        else:
            self._graphs[key] = self._graphs.pop(key)    # most recently used last
        graph, static, out = entry
        if static is not job:
            for k in self._TENSOR_KEYS:
                static[k].copy_(job[k], non_blocking=True)
            for d, s_ in zip(static["noises"], job["noises"]):
                d.copy_(s_, non_blocking=True)
        graph.replay()
        job["_result"] = static["_result"]               # the replayed entry's own latents buffer / stats
        self.last_stats = dict(static["_result"]["stats"])
        return out

    def _capture(self, key, job, slot):
        static = dict(job)
        for k in self._TENSOR_KEYS:
            static[k] = job[k].clone()
        static["noises"] = [n.clone() for n in job["noises"]]
        static["t_dev"] = [t.clone() for t in job["t_dev"]]
        timing, self.timing = self.timing, None
        self.ctx.ws_tag = slot
        self.ctx.autotune(self._tune_mode())
        self.run_device(static)                     # eager warm-up: lazy workspaces / function attributes, tile autotune
        torch.cuda.synchronize()
        self.ctx.autotune(2 if self.autotune else 0)   # frees the tuner's scratch; the capture below uses what it remembered
        graph = torch.cuda.CUDAGraph()
        pool = self._pools.get(slot)
        if pool is None:
            pool = self._pools[slot] = torch.cuda.graph_pool_handle()
        # The capture stream is ours (created here, where torch would create its default one: same first-use order of streams) so that
        # its split-K workspace can be bound BEFORE the capture: allocated inside, its torch.zeros became a 96 MB memset node that every
        # replay of the slot's first graph re-ran
        cap = self.ctx.capture_stream()
        # thread_local: other worker threads keep replaying / allocating on their own streams during this capture
        with torch.cuda.graph(graph, pool=pool, stream=cap, capture_error_mode="thread_local"):
            out = self.run_device(static)
        self.ctx.ws_tag = 0
        self.timing = timing
        self._graphs[key] = (graph, static, out)
        return self._graphs[key]

    def stage_ms(self):
        """Per-stage device milliseconds of the last run_device() (needs `self.timing = []` before the call)."""
        torch.cuda.synchronize()
        out = {}
        for (_, e0), (name, e1) in zip(self.timing[:-1], self.timing[1:]):
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        return out

    def new_slot_stream(self, slot, equal_priority=False):
        """A stream for graph slot `slot`.  Odd slots take a high-priority stream: ROCm keeps separate hardware queues per
        priority, so slots 0 and 1 can never be dealt onto one queue (which would serialise the two edits), and with worker
        threads (PIL in -> PIL out) the priority split measures best: 13.3 vs 12.9 images/s.  `equal_priority` is for a caller
        that replays device-resident jobs back to back (bench.py's timed loop: 14.1 vs 13.8 images/s) and needs >= 8 hardware
        queues (fie_amd.py sets 16) for the streams to stay on separate queues."""
        few_queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8
        return torch.cuda.Stream(device=self.ctx.device, priority=0 if equal_priority and not few_queues else -(slot % 2))

    def calibrate_streams(self, jobs, streams, tries=4, log=None):
        """Pick launch streams on which `len(jobs)` edits really overlap.  jobs[i] is replayed on slot i (its graph must exist or
        is captured here); candidates are `streams` and up to `tries - 1` freshly drawn sets; each is timed with two rounds of
        concurrent replays and the fastest set is returned.  Replay is stream-agnostic, so no graph is re-captured."""
        def timed(strs):
            torch.cuda.synchronize(self.ctx.device)
            t0 = torch.cuda.Event(enable_timing=True)
            t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(2):
                for i, (job, st) in enumerate(zip(jobs, strs)):
                    with torch.cuda.stream(st):
                        self.run_device_graphed(job, slot=i)
            torch.cuda.synchronize(self.ctx.device)
            t1.record()
            torch.cuda.synchronize(self.ctx.device)
            return t0.elapsed_time(t1)

        for i, (job, st) in enumerate(zip(jobs, streams)):          # captures (if needed) + first replays: untimed
            with torch.cuda.stream(st):
                self.run_device_graphed(job, slot=i)
        best, best_ms = list(streams), timed(streams)
        first_ms = best_ms
        for _ in range(tries - 1):
            cand = [self.new_slot_stream(i, equal_priority=all(s.priority == 0 for s in streams)) for i in range(len(jobs))]
            ms = timed(cand)
            if ms < best_ms:
                best, best_ms = cand, ms
        if log:
            log(f"stream calibration: first set {first_ms:.1f} ms, chosen {best_ms:.1f} ms for 2 rounds of {len(jobs)} edits in flight")
        return best

    def slot_stream(self, slot):
        """Stream of graph slot `slot`.  Every slot, slot 0 included, owns a stream: replaying an edit's graph on the legacy
        null stream costs +30 ms per edit once other streams exist in the process (measured, 92 vs 122 ms end to end)."""
        if slot not in self._slot_streams:
            self._slot_streams[slot] = self.new_slot_stream(slot)
        return self._slot_streams[slot]

    def __call__(self, prompt, negative_prompt="", image=None, control_image=None, strength=0.8,
                 num_inference_steps=4, guidance_scale=1.5, controlnet_conditioning_scale=0.5, generator=None,
                 output_type="pil", slot=0, post_check=None, **unused):
        """`slot` (additive): independent hipGraph instance + stream, so that several calls may be in flight from different
        host threads on one GPU (graph mode only).  `post_check` (additive): a callable run after the result has reached the host
        (the stream is idle then); when it returns True the device-resident inputs have changed meanwhile and the device job is run
        again.  FastEditor.edit() passes the second half of its asynchronous device Canny: the edge map is computed with a fixed
        number of hysteresis rounds in front of the edit, and whether they had reached the fixed point is only looked at here --
        no host wait in front of the edit, a repeated job in the rare case that they had not."""
        if slot and not self.use_graph:
            raise ValueError("slots > 0 need hipGraph replay (the eager path shares per-image state)")
        caller, st = torch.cuda.current_stream(self.ctx.device), self.slot_stream(slot)
        st.wait_stream(caller)                           # device-resident inputs may still be in flight on the caller's stream
        with torch.cuda.stream(st):
            out = self._call(prompt, negative_prompt, image, control_image, strength, num_inference_steps, guidance_scale,
                             controlnet_conditioning_scale, generator, output_type, slot, post_check)
        caller.wait_stream(st)
        return out

    def _to_host(self, out_u8, slot):
        """D2H of the u8 result through a pinned staging buffer of the slot (a pageable `.cpu()` costs ~3x as much), then one
        host memcpy so the caller owns its array.  Synchronises the current stream only."""
        key = (slot, tuple(out_u8.shape))
        host = self._host_out.get(key)
        if host is None:
            host = self._host_out[key] = torch.empty(out_u8.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(out_u8, non_blocking=True)
        self.ctx.fetch_device_errors()                   # 16 bytes behind the image: a kernel that had to give up does not stay silent
        torch.cuda.current_stream(self.ctx.device).synchronize()
        self.ctx.check_device_errors(fetch=False)
        return host.numpy().copy()

    def _call(self, prompt, negative_prompt, image, control_image, strength, num_inference_steps, guidance_scale,
              controlnet_conditioning_scale, generator, output_type, slot, post_check=None):
        if isinstance(prompt, (list, tuple)):            # [additive] a batch: lists of prompts / images / generators
            job = self.prepare_batch(list(prompt), negative_prompt if isinstance(negative_prompt, (list, tuple)) else None,
                                     list(image), list(control_image), strength, num_inference_steps, guidance_scale,
                                     controlnet_conditioning_scale, generator if isinstance(generator, (list, tuple)) else None)
            out_u8 = self.run_device_graphed(job, slot) if self.use_graph else self._run_eager(job)
            arr = self._to_host(out_u8, slot)
            if post_check is not None and post_check():
                out_u8 = self.run_device_graphed(job, slot) if self.use_graph else self._run_eager(job)
                arr = self._to_host(out_u8, slot)
            arr = arr[None] if arr.ndim == 3 else arr
            if output_type == "np":
                return types.SimpleNamespace(images=list(arr))
            return types.SimpleNamespace(images=[Image.fromarray(a) for a in arr])
        job = self.prepare(prompt, negative_prompt, image, control_image, strength, num_inference_steps,
                           guidance_scale, controlnet_conditioning_scale, generator)
        out_u8 = self.run_device_graphed(job, slot) if self.use_graph else self._run_eager(job)
        if output_type == "latent":
            res = job["_result"]
            lh, lw = res["stats"]["latent_hw"]
            return types.SimpleNamespace(images=[res["latents"].view(lh, lw, 4).clone()])
        arr = self._to_host(out_u8, slot)              # device -> host sync, as `.images[0]` implies upstream
        if post_check is not None and post_check():
            self.repeated_jobs += 1
            out_u8 = self.run_device_graphed(job, slot) if self.use_graph else self._run_eager(job)
            arr = self._to_host(out_u8, slot)
        if output_type == "np":
            return types.SimpleNamespace(images=[arr])
        return types.SimpleNamespace(images=[Image.fromarray(arr)])
