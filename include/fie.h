/*
 * fie.h -- C ABI of the MI355X (gfx950) hot-path library `libfie_hip.so`.
 *
 * The reference (/root/reference) is pure Python and has NO FFI of its own: every multiply-add of its hot call
 * `self.pipe(...)` (src/pipeline.py:261-272) runs inside diffusers/transformers/torch/OpenCV.  This header is
 * therefore the boundary a maintainer would bind from Python (ctypes stub in INTEGRATION.md) to replace those
 * upstream internals.  Each entry names the reference call site / upstream module it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns 0 on success or a
 *     negative FIE_E* code; `fie_last_error()` returns a thread-local message for the last failure.
 *   - All device pointers are owned by the caller (PyTorch-ROCm allocations: tensor.data_ptr()).  The library never
 *     allocates device memory; kernels that need scratch take a caller-provided workspace.
 *   - Every launch is asynchronous on the ctx's stream and never synchronises (hipGraph-capturable).
 *   - fp16 storage ("f16" = IEEE binary16), fp32 accumulation.  Activations are NHWC; a tensor with C channels
 *     has C % 8 == 0 (3/4-channel images and latents are stored zero-padded to 8 channels unless stated).
 *   - One ctx per (thread, device); a ctx is not thread-safe; distinct ctxs are independent.
 */
#ifndef FIE_H_
#define FIE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FIE_OK 0
#define FIE_EINVAL (-1)   /* bad argument / unsupported shape */
#define FIE_EHIP (-2)     /* HIP runtime error */
#define FIE_ENODEV (-3)   /* no gfx950 device */

/* Device-side failures cannot return a code from an asynchronous launch: a kernel that has to give up (today: the bounded spin of the
 * in-launch barrier of fie_time_embed_f16) stores one of these into the word bound with fie_ctx_error_flag, and the host reads that word
 * at its next synchronisation (fie_amd: after the D2H copy of every edit; non-zero raises).  The word is sticky until the host clears it. */
#define FIE_DEVERR_TIME_EMBED_BARRIER 1u

/* epilogue activations (applied to acc + bias [+ rowbias]) */
#define FIE_ACT_NONE 0
#define FIE_ACT_SILU 1
#define FIE_ACT_GELU 2        /* exact erf GELU */
#define FIE_ACT_QUICK_GELU 3  /* x * sigmoid(1.702 x), CLIP-L */
#define FIE_ACT_GEGLU 4       /* weight rows interleaved (value, gate): out[j] = v_j * gelu(g_j); N counts both */

typedef struct fie_ctx fie_ctx;

int fie_version(void);
const char* fie_last_error(void);
/* stream: hipStream_t (0 = default stream).  Replaces: nothing in the reference (torch owns streams there). */
int fie_ctx_create(int device, void* stream, fie_ctx** out);
int fie_ctx_set_stream(fie_ctx* ctx, void* stream);
int fie_ctx_error_flag(fie_ctx* ctx, void* device_word);    /* caller-owned, zero-initialised uint32 in device memory (NULL: none); see FIE_DEVERR_* */
int fie_ctx_destroy(fie_ctx* ctx);

/* ---- Graph-level entries: the upstream calls they replace are the model forwards inside the diffusers pipeline call
 * at /root/reference/src/pipeline.py:261-272 -- UNet2DConditionModel.forward, ControlNetModel.forward,
 * AutoencoderKL.encode / .decode, CLIPTextModel(.WithProjection).forward.
 *
 * Mechanism: LAUNCH PROGRAMS.  The host walks a graph once, with the device buffers it will keep using (weights, static
 * input / output / scratch tensors), between fie_program_begin and fie_program_end: every kernel launch the op entries below
 * issue meanwhile EXECUTES as usual and is ALSO appended to the program (kernel, grid, block, LDS bytes, argument values).
 * fie_program_run re-issues the whole list on the ctx stream from C++: no host-side shape logic, no Python, asynchronous and
 * hipGraph-capturable like any single op.  New inputs = new CONTENTS of the same input buffers.  The caller keeps every buffer
 * the program references alive and destroys programs it created.
 * ORDERING CONTRACT (round 4).  A program's launches carry FROZEN pointers: its static buffers and the split-K workspace
 * (fie_splitk_workspace) that was bound to the ctx while it was recorded -- arrival counters and partial-tile slabs included.  Therefore
 *   (1) a program must never be in flight twice.  fie_program_run enforces this for eager runs: the recording pass and every run record
 *       an event, and a run on a DIFFERENT stream than the previous pass first makes that stream wait for it (same stream: stream order).
 *       Under stream capture nothing is ordered by the library: a captured graph that contains the program inherits the rule (do not
 *       replay it while another copy of the program, captured or eager, is running);
 *   (2) nothing else may use that split-K workspace concurrently: record a program with a workspace of its OWN bound
 *       (fie_amd/hip.py: Context.record() allocates one per program and restores the previous binding afterwards);
 *   (3) the buffers' producers are the caller's to order: make the run's stream wait for whatever wrote the inputs.
 * A caller that breaks (1) / (2) gets undefined sums from the overlapping launches, but no lasting damage: the arrival counters are
 * self-healing (every S-th arrival takes S off the counter; csrc/gemm_common.h: splitk_reduce).
 * What the five NAMED entries (fie_unet_forward ...) are NOT: a C++ implementation of the model graphs.  They are REPLAY HANDLES for a launch list
 * a host walk recorded, with every pointer frozen in.  The C++ implementations are the *_f16 forwards below (round 3).  The product path
 * (hipGraph replay of the Python walk, which takes more fusions) goes through neither; tests/test_programs_gpu.py covers both.
 *   fie_graph_register binds a program to one of the names "unet_forward", "controlnet_forward", "vae_encode", "vae_decode",
 *   "clip_text_forward"; the five named entries run the program bound to their name (FIE_EINVAL if none). */
/* ---- Graph-level forwards that ARE forwards (round 3, csrc/graphs.cpp): the five model calls inside the pipeline call at
 * /root/reference/src/pipeline.py:261-272 sequenced in C++ over the op entries below, on weights registered once by name -- tensor arguments in
 * and out, no Python graph code, no recorded launch list; what a non-Python host binds.
 *   fie_weights_register(name, ptr, n, ld): name = "<prefix><diffusers parameter name>" (prefix = the `prefix` argument of the forward, "" for the
 *     VAE).  "<conv>.weight" = fie_pack_conv3x3_f16 output (n = Cout, ld = ldw), "<linear>.weight" = fie_pack_rows_f16 output (n = N), biases /
 *     norm gains as plain f16 vectors (ld = 0).  Fused matrices (rows concatenated, then packed; biases likewise): "...attn1.to_qkv",
 *     "...attn2.to_kv", "...self_attn.qkv_proj" (CLIP), "<vae>.mid_block.attentions.0.to_qkv", "time_emb_proj_all" (every resnet's time_emb_proj
 *     in walk order: down, mid, up); "...ff.net.0.proj" in the GEGLU layout (fie_pack_rows_f16 with geglu: value / gate rows interleaved, bias
 *     likewise); "post_quant_conv.{weight,bias}" zero-padded to the 8-channel latent layout; CLIP: "zero_row" = `hidden` zeros.
 *   Workspaces: ONE caller-provided device buffer per call, >= the matching *_workspace_bytes(cfg) (the walk's allocation sequence replayed
 *     without launching: exact).  Asynchronous on the ctx stream, hipGraph-capturable, no allocation, no synchronisation.
 *     The walks' GEMMs / convs split K where the tuner chose so, on the split-K workspace currently bound to the ctx (fie_splitk_workspace):
 *     bind the workspace of the stream you are about to issue on BEFORE calling a forward (fie_amd/cabi.py does, per (stream, slot)).
 *   Same kernels and order as the Python walks (fie_amd/{clip,vae,nn}.py) apart from fusions those take and these do not (GroupNorm sums from the
 *     producing epilogue, 2x2-parity up-samplers, conv2 + shortcut as one GEMM, the one-launch timestep embedding, zero convs adding into the UNet's
 *     skips): results agree to rounding (tests/test_programs_gpu.py: <= 3e-3 of the output's max against the Python walk at full size).
 *   Layouts: activations NHWC f16; images and latents 8-channel zero-padded ([B, H, W, 8]); eps / VAE pixels 4-channel ([.., 4], 3 or 4 real). */
typedef struct fie_vae_config {
    int latent_h, latent_w;
    int num_blocks;                /* len(block_out_channels), <= 8 */
    int block_out_channels[8];     /* encoder order, e.g. 128, 256, 512, 512 */
    int layers_per_block;          /* 2: the decoder runs layers_per_block + 1 resnets per up block */
    int norm_num_groups;           /* 32 */
    float norm_eps;                /* 1e-6 */
    int out_channels;              /* 3 */
    const char* prefix;            /* the weights were registered as "<prefix>encoder....", "<prefix>post_quant_conv", ...; NULL: no prefix */
} fie_vae_config;
typedef struct fie_clip_config {
    int batch, tokens;             /* ids: int32 [batch * tokens] on the device */
    int hidden, heads, layers, intermediate;
    int projection_dim;            /* 0: no text_projection (CLIP-L as SDXL uses it): the last layer is not run */
    int quick_gelu;                /* 1: quick_gelu (CLIP-L), 0: gelu (OpenCLIP bigG) */
    float eps;                     /* 1e-5 */
} fie_clip_config;
typedef struct fie_unet_config {   /* UNet2DConditionModel (SDXL family) and the ControlNetModel built on its encoder */
    int batch;                     /* rows of x (CFG rows included) */
    int latent_h, latent_w;        /* multiples of 2^(num_blocks - 1) */
    int text_len;                  /* text rows per batch row (77) */
    int num_blocks;                /* <= 4 */
    int block_out_channels[4];     /* 320, 640, 1280 */
    int layers_per_block;          /* 2 (<= 3) */
    int down_attn[4][4];           /* transformer depth behind resnet j of down block i (0: none) */
    int up_attn[4][4];             /* likewise for the layers_per_block + 1 resnets of up block i (UNet only) */
    int mid_attn;                  /* depth of the mid-block transformer (0: none) */
    int mid_resnets;               /* 2: resnet, transformer, resnet;  1: one resnet */
    int head_dim;                  /* 64 */
    int norm_num_groups;           /* 32 */
    float norm_eps;                /* 1e-5 */
    int cross_attention_dim;       /* 2048 */
    int addition_time_embed_dim;   /* 256 */
    int pooled_dim;                /* 1280: add_embedding.linear_1 reads pooled_dim + 6 * addition_time_embed_dim columns */
    int num_cond_channels;         /* ControlNet only: len(conditioning_embedding_out_channels), <= 8; cond is [B, h * 2^(n-1), w * 2^(n-1), 8] */
    int cond_channels[8];          /* 16, 32, 96, 256 */
} fie_unet_config;
int fie_weights_register(fie_ctx* ctx, const char* name, const void* ptr, int64_t n, int64_t ld);
int fie_weights_clear(fie_ctx* ctx);
int fie_weights_clear_prefix(fie_ctx* ctx, const char* prefix);     /* forgets every name that starts with prefix (a model that goes away) */
/* AutoencoderKL.decode(latents / scaling_factor): z [1, h, w, 8] -> out [1, 8 h, 8 w, 4];  names "post_quant_conv", "decoder...." */
int64_t fie_vae_decode_workspace_bytes(const fie_vae_config* cfg, int latent_h, int latent_w);
int fie_vae_decode_f16(fie_ctx* ctx, const fie_vae_config* cfg, const void* z, void* out, void* workspace, int64_t workspace_bytes);
/* AutoencoderKL.encode(x).latent_dist parameters: x [1, 8 h, 8 w, 8] in [-1, 1] -> moments [h * w, 8] (mean | logvar);  names "encoder....", "quant_conv" */
int64_t fie_vae_encode_workspace_bytes(const fie_vae_config* cfg);
int fie_vae_encode_f16(fie_ctx* ctx, const fie_vae_config* cfg, const void* x, void* moments, void* workspace, int64_t workspace_bytes);
/* CLIPTextModel(.WithProjection)(ids, output_hidden_states=True): penultimate = hidden_states[-2] [batch * tokens, hidden]; pooled (projection_dim > 0
 * and pooled != NULL) = text_projection(final_layer_norm(last)[eos_rows]) [batch, projection_dim]; eos_rows: int32 [batch] ROW indices b * tokens + eos */
int64_t fie_clip_text_workspace_bytes(const fie_clip_config* cfg);
int fie_clip_text_forward_f16(fie_ctx* ctx, const fie_clip_config* cfg, const char* prefix, const int32_t* ids, const int32_t* eos_rows, void* penultimate,
                              void* pooled, void* workspace, int64_t workspace_bytes);
/* UNet2DConditionModel.forward(sample, t, encoder_hidden_states, added_cond_kwargs = {text_embeds, time_ids}, down_block_additional_residuals,
 * mid_block_additional_residual): x [B, h, w, 8], t f32 [B], text [B * text_len, cross_attention_dim], pooled [B, pooled_dim], time_ids f32 [B, 6]
 * (all on the device) -> eps [B, h, w, 4].  down_residuals: fie_unet_num_residuals(cfg) device pointers in skip order (conv_in output first), or NULL. */
int fie_unet_num_residuals(const fie_unet_config* cfg);
int64_t fie_unet_workspace_bytes(const fie_unet_config* cfg);
int fie_unet_forward_f16(fie_ctx* ctx, const fie_unet_config* cfg, const char* prefix, const void* x, const float* t, const void* text, const void* pooled,
                         const float* time_ids, const void* const* down_residuals, const void* mid_residual, void* eps_out, void* workspace,
                         int64_t workspace_bytes);
/* ControlNetModel.forward(sample, t, encoder_hidden_states, controlnet_cond, conditioning_scale, added_cond_kwargs): cond [B, H, W, 8] in [0, 1] ->
 * down_out[i] (shapes of the UNet's skips), mid_out: the zero-conv outputs times conditioning_scale, as upstream returns them */
int64_t fie_controlnet_workspace_bytes(const fie_unet_config* cfg);
int fie_controlnet_forward_f16(fie_ctx* ctx, const fie_unet_config* cfg, const char* prefix, const void* x, const float* t, const void* text, const void* pooled,
                               const float* time_ids, const void* cond, float conditioning_scale, void* const* down_out, void* mid_out, void* workspace,
                               int64_t workspace_bytes);
/* Step cache (optional): what a UNet / ControlNet forward computes from inputs that do NOT change over the denoising steps of one image -- the cross-
 * attention K / V of the text in every transformer block (upstream attention_processor.py: to_k / to_v of encoder_hidden_states) and the ControlNet's
 * conditioning embedding of the edge map -- kept in a caller-owned device buffer bound to the context under the model's prefix.  The first
 * fie_unet_forward_f16 / fie_controlnet_forward_f16 with that prefix after a bind or a reset FILLS the buffer (and ignores nothing: results are the
 * same bits), every later one READS it and skips those launches (SSD-1B: ~60 small GEMMs per UNet call, seven convs per ControlNet call).  The host
 * resets at every new image (new text / edge map); text, cond and the batch must stay what they were at the fill.  The decision is taken on the host when
 * the call is issued, so a hipGraph captured over a whole edit (reset, N x forwards) replays correctly.  Without a bound cache every call computes them. */
int64_t fie_unet_step_cache_bytes(const fie_unet_config* cfg, int controlnet);
int fie_step_cache_bind(fie_ctx* ctx, const char* prefix, void* ptr, int64_t bytes);     /* ptr NULL: unbind */
int fie_step_cache_reset(fie_ctx* ctx, const char* prefix);
/* the two element-wise helpers the walks need (also plain ops): out = a + b over n f16 values (n % 8 == 0); a strided row copy (cols % 8 == 0) */
int fie_add_f16(fie_ctx* ctx, const void* a, const void* b, void* out, int64_t n);
int fie_copy_rows_f16(fie_ctx* ctx, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols);

typedef struct fie_program fie_program;
int fie_program_begin(fie_ctx* ctx, fie_program** out);
int fie_program_end(fie_ctx* ctx);
int fie_program_launches(const fie_program* p);             /* number of recorded launches (-1: NULL) */
int fie_program_run(fie_ctx* ctx, fie_program* p);          /* see ORDERING CONTRACT above */
int fie_program_destroy(fie_ctx* ctx, fie_program* p);      /* also drops its name bindings on ctx (ctx may be NULL) */
int fie_graph_register(fie_ctx* ctx, const char* name, fie_program* p);
int fie_unet_forward(fie_ctx* ctx);
int fie_controlnet_forward(fie_ctx* ctx);
int fie_vae_encode(fie_ctx* ctx);
int fie_vae_decode(fie_ctx* ctx);
int fie_clip_text_forward(fie_ctx* ctx);

/* ---- K2 GEMM (Linear / 1x1 conv).  Replaces torch.nn.Linear / 1x1 Conv2d dispatched by upstream
 * diffusers models/attention.py, transformer_2d.py, resnet.py (conv_shortcut), controlnet.py (zero convs),
 * embeddings.py (TimestepEmbedding), transformers modeling_clip.py -- all reached from src/pipeline.py:261.
 *   C[m, n] = epi( sum_k A[m, k] * W[n, k] ),  A = [A1 | A2] column-concatenated (A2 may be NULL; K1 = K then)
 *   epi(v) = act(v + bias[n] + rowbias[m / rows_per_batch, n]) * scale + residual[m, n]
 * W is the PACKED weight: [Npad][Kpad] f16, Kpad = roundup(K, 64), Npad = roundup(N, 128), zero filled (see
 * fie_pack_rows).  lda*, ldc, ldr in elements, multiples of 8 (ldc, ldr: multiples of 4).  K1 % 8 == 0, K % 8 == 0.
 * For FIE_ACT_GEGLU the output has N/2 columns. */
int fie_gemm_f16(fie_ctx* ctx, const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2,
                 const void* Wpacked, int64_t ldw, void* C, int64_t ldc, int M, int N, int K,
                 const void* bias, const void* rowbias, int64_t ld_rowbias, int rows_per_batch,
                 const void* residual, int64_t ldr, float scale, int act);

/* ---- LayerNorm folded into its consumer GEMM.  Replaces the pair nn.LayerNorm -> nn.Linear of upstream diffusers models/attention.py
 * BasicTransformerBlock (norm1 -> attn1.to_q/k/v, norm2 -> attn2.to_q, norm3 -> ff.net.0.proj), reached from src/pipeline.py:261.
 *   C[m, n] = act( sum_k LN(X)[m, k] * W[n, k] + bias[n] ),  LN(x) = (x - mean) * rsqrt(var + eps) * gamma + beta over the K columns of row m
 * computed WITHOUT ever forming LN(X):  = rstd[m] * (sum_k X[m, k] * Wf[n, k] - mean[m] * ln_tab[n][0]) + ln_tab[n][1]  with
 *   Wfolded = pack(W * gamma) (f16, the packed layout of fie_gemm_f16),  ln_tab[n] = (sum_k Wfolded[n, k], sum_k W[n, k] * beta[k] + bias[n]) in fp32,
 * in the PACKED column order (GEGLU: value / gate interleaved as fie_pack_rows_f16 with interleave2).  The row statistics are summed (fp32) from the
 * activation fragments on their way to the MFMAs; var = E[x^2] - mean^2, clamped at 0.  X: [M, K] f16 un-normalised (ldx elements), K % 64 == 0;
 * act = FIE_ACT_NONE or FIE_ACT_GEGLU (N % 320 == 0: the 256x320 tile; output has N/2 columns).  Differences from the two-launch sequence: the
 * normalised activation is not rounded to f16, W * gamma is (once, at load). */
int fie_gemm_ln_f16(fie_ctx* ctx, const void* X, int64_t ldx, const void* Wfolded, int64_t ldw, const float* ln_tab, float eps, void* C, int64_t ldc,
                    int M, int N, int K, int act);

/* ---- K1 3x3 convolution, NHWC, implicit GEMM on fp16 MFMA.  Replaces torch Conv2d(3x3) dispatched by upstream
 * resnet.py / downsampling.py / upsampling.py / vae.py / controlnet.py conditioning embedding.
 *   X: [B, H, W, Cin] f16 (Cin % 8 == 0).  If upsample2x != 0 the conv sees nearest-2x upsampled X (never stored).
 *   pad_mode 0: symmetric padding 1.  pad_mode 1: pad (right, bottom) only -- the VAE encoder's F.pad(0,1,0,1) + pad 0.
 *   Y: [B, OH, OW, ldc] with OH = (Hin + pads - 3)/stride + 1.  Wpacked: [Npad][Kpad], k = (ky*3 + kx)*Cin + ci.
 *   epilogue as fie_gemm_f16 with rows_per_batch = OH*OW (rowbias = per-image time-embedding projection). */
int fie_conv3x3_nhwc_f16(fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, int upsample2x, int stride,
                         int pad_mode, const void* Wpacked, int64_t ldw, void* Y, int64_t ldc, int Cout,
                         const void* bias, const void* rowbias, int64_t ld_rowbias,
                         const void* residual, int64_t ldr, float scale, int act);

/* ---- K3/K3v/K4 + CLIP attention: softmax(scale * Q K^T [+ causal mask]) V, online softmax, one kernel family.
 * Replaces F.scaled_dot_product_attention reached via upstream attention_processor.py (UNet/ControlNet attn1/attn2,
 * VAE mid-block attention) and modeling_clip.py self-attention.
 *   element (b, t, h, d) of Q lives at Q[(b*Tq + t)*ldq + h*D + d]; K/V likewise with Tk, ldk, ldv; O with ldo.
 *   D in {64, 512}.  kv_batch_stride_zero != 0 broadcasts one K/V over the batch. */
int fie_attention_f16(fie_ctx* ctx, const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V,
                      int64_t ldv, void* O, int64_t ldo, int B, int H, int Tq, int Tk, int D, float scale,
                      int causal);

/* ---- K5 GroupNorm (+SiLU), NHWC, fp32 statistics.  Replaces torch GroupNorm + SiLU in resnet.py /
 * transformer_2d.py / vae.py.  Input is the channel concatenation [X1 (C1) | X2 (C2)] (X2 may be NULL): this
 * is how the UNet up blocks' torch.cat([x, skip]) is consumed without materialising the concat.
 *   workspace: fie_groupnorm_workspace_bytes(B, rows, G) bytes of device scratch. */
int64_t fie_groupnorm_workspace_bytes(int B, int64_t rows_per_image, int groups);
int fie_groupnorm_nhwc_f16(fie_ctx* ctx, const void* X1, int C1, const void* X2, int C2, void* Y, int B,
                           int64_t rows_per_image, int groups, const void* gamma, const void* beta, float eps,
                           int silu, void* workspace);

/* ---- K6 LayerNorm over the last dim (affine).  Replaces torch LayerNorm in attention.py / modeling_clip.py. */
int fie_layernorm_f16(fie_ctx* ctx, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t rows, int C,
                      const void* gamma, const void* beta, float eps);

/* ---- K7 sinusoidal embeddings.  Replaces embeddings.py::get_timestep_embedding (flip_sin_to_cos, shift 0).
 *   out[b, i*dim .. (i+1)*dim) = [cos(v f), sin(v f)] for each of the `nvals` scalars v of row b; out row
 *   stride ld_out, column offset col0 (so the 6 add_time_ids land behind the pooled text embedding). */
int fie_sinusoid_f16(fie_ctx* ctx, const float* vals, int B, int nvals, int dim, void* out, int64_t ld_out,
                     int col0);

/* ---- K7 fused: the whole per-step timestep path in ONE launch (sinusoid -> Linear -> SiLU -> Linear -> + text-time embedding ->
 * the SiLU in front of every resnet's time projection).  Replaces Timesteps + TimestepEmbedding of embeddings.py and the
 * `emb = emb + aug_emb` / `nonlinearity(temb)` steps of unet_2d_condition.py / resnet.py for the M = batch rows of one step.
 *   t: f32 [B] on the device; W1 [E][C0], W2 [E][E] plain row-major f16 (NOT packed), b1, b2 [E]; add: [B][ld_add] f16 or NULL;
 *   out[b, n] = silu(W2 silu(W1 [cos(t f) | sin(t f)] + b1) + b2 + add[b, n]).  B <= 4, C0 % 32 == 0, E % 64 == 0.
 *   workspace: fie_time_embed_workspace_bytes(E) bytes, ZERO-filled once by the caller (hidden vector + the two counters of the
 *   in-launch barrier between the layers; the kernel leaves the counters at zero; the third counter word is a sticky "a workgroup's
 *   bounded spin expired: the output of that launch is not valid" flag, mirrored into the fie_ctx_error_flag word).  One workspace per
 *   (stream, model): launches that may run concurrently must not share one. */
int64_t fie_time_embed_workspace_bytes(int E);
int fie_time_embed_f16(fie_ctx* ctx, const float* t, int B, int C0, int E, const void* W1, const void* b1, const void* W2,
                       const void* b2, const void* add, int64_t ld_add, void* out, int64_t ld_out, void* workspace);

/* ---- K12 token + position embedding gather (CLIPTextEmbeddings). ids: int32 [B*T] on device. */
int fie_clip_embed_f16(fie_ctx* ctx, const int32_t* ids, int B, int T, int C, const void* tok_table,
                       const void* pos_table, void* out);

/* ---- K10 pixels.  Replaces image_processor.py::VaeImageProcessor.preprocess / postprocess.
 *   in : u8 HWC (3 ch) -> f16 NHWC padded to 8 ch; normalize != 0: x/255*2-1, else x/255; `copies` batch copies.
 *   out: f16 NHWC (ld_in channels, first 3 used) -> u8 HWC: round(clamp(x/2+0.5, 0, 1) * 255). */
int fie_pixels_in_u8_f16(fie_ctx* ctx, const uint8_t* src, int H, int W, int normalize, void* dst, int copies);
int fie_pixels_out_f16_u8(fie_ctx* ctx, const void* src, int64_t ld_in, int H, int W, uint8_t* dst);

/* ---- K9 latent prep.  Replaces DiagonalGaussianDistribution.sample, * scaling_factor, LCMScheduler.add_noise.
 *   moments: f16 [H*W, 8] (mean 0..3, logvar 4..7); eps_post, noise: f32 [4, H*W] (NCHW draw order, as torch.randn
 *   of shape [1,4,H,W]); latents_out: f32 [H*W, 4]; model_in: f16 [copies, H*W, 8] (zero padded). */
int fie_latent_prep(fie_ctx* ctx, const void* moments, const float* eps_post, const float* noise, int64_t HW,
                    float scaling_factor, float sqrt_ab, float sqrt_1mab, float* latents_out, void* model_in,
                    int copies);

/* ---- K8 CFG combine + LCMScheduler.step.  Replaces pipeline CFG line + scheduling_lcm.py::step.
 *   eps: f16 [nb, H*W, ld_eps] (nb = 2 when guidance is applied: [uncond, cond]); latents: f32 [H*W, 4] (in/out);
 *   noise: f32 [4, H*W] or NULL on the last step; model_in as in fie_latent_prep; denoised_out may be NULL. */
int fie_lcm_step(fie_ctx* ctx, const void* eps, int64_t ld_eps, int nb, float guidance, float* latents,
                 const float* noise, int64_t HW, float sqrt_ab_t, float sqrt_1mab_t, float c_skip, float c_out,
                 float sqrt_ab_prev, float sqrt_1mab_prev, void* model_in, int copies, float inv_scaling,
                 void* decode_in);

/* ---- weight repack (one-time, on device): src [N][K] f16 row-major (ld_src) -> dst [Npad][Kpad] zero padded.
 *   conv: src is OIHW [Cout][Cin][3][3]; dst k-order (ky, kx, ci) with Cin padded to cin_pad.
 *   interleave2 != 0: rows (j, j + N/2) become adjacent (GEGLU value/gate pairing). */
int fie_pack_rows_f16(fie_ctx* ctx, const void* src, int64_t ld_src, int N, int K, void* dst, int64_t ldw,
                      int Npad, int interleave2);
int fie_pack_conv3x3_f16(fie_ctx* ctx, const void* src_oihw, int Cout, int Cin, int cin_pad, void* dst,
                         int64_t ldw, int Npad);

/* ---- fp8 (OCP e4m3) WEIGHTS on the fp8 MFMA: BASELINE.json config 5, the low-precision stretch beside the precision choice at
 * /root/reference/src/pipeline.py:67-71 (`FastEditor(..., weight_dtype="f8e4m3")`, additive).  Weights are e4m3 with one fp32
 * scale per output channel (scale[n] = max_k |W[n,k]| / 448); activations stay fp16 in HBM and are converted to e4m3 per fragment
 * in registers (saturating RNE, scale 1) for v_mfma_f32_16x16x32_fp8_fp8; fp32 accumulation;  epi(v) takes v * scale[n].
 *   fie_pack_rows_f8 / fie_pack_conv3x3_f8: as their _f16 twins, dst = [Npad][ldw] BYTES (ldw % 64 == 0), scales = [Npad] fp32 (out).
 *   fie_gemm_w8_f16 / fie_conv3x3_w8_nhwc_f16: as fie_gemm_f16 / fie_conv3x3_nhwc_f16 with (W8packed, ldw in bytes, w_scale).
 *   Shapes must be eligible for the LDS-DMA kernels (operands < 2 GiB, Cin % 64 == 0, K1 == K or K1 % 64 == 0), else FIE_EINVAL. */
int fie_pack_rows_f8(fie_ctx* ctx, const void* src, int64_t ld_src, int N, int K, void* dst, int64_t ldw, int Npad, float* scales,
                     int interleave2);
int fie_pack_conv3x3_f8(fie_ctx* ctx, const void* src_oihw, int Cout, int Cin, int cin_pad, void* dst, int64_t ldw, int Npad,
                        float* scales);
int fie_gemm_w8_f16(fie_ctx* ctx, const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2, const void* W8packed,
                    int64_t ldw, const float* w_scale, void* C, int64_t ldc, int M, int N, int K, const void* bias,
                    const void* rowbias, int64_t ld_rowbias, int rows_per_batch, const void* residual, int64_t ldr, float scale,
                    int act);
int fie_conv3x3_w8_nhwc_f16(fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, int upsample2x, int stride, int pad_mode,
                            const void* W8packed, int64_t ldw, const float* w_scale, void* Y, int64_t ldc, int Cout,
                            const void* bias, const void* rowbias, int64_t ld_rowbias, const void* residual, int64_t ldr,
                            float scale, int act);

/* ---- fp8 (OCP e4m3) ACTIVATIONS as well (config 5, round 3): the transformer-block projections of the UNet / ControlNet (upstream
 * attention.py BasicTransformerBlock: to_q/k/v, to_out, GEGLU proj, FF out) read e4m3 activations written by their PRODUCER -- LayerNorm
 * (fie_layernorm_f16_o8), attention (fie_attention_f16_o8), the FF1 GEGLU epilogue (fie_gemm_x8_f16 with out_f8) -- and multiply them with
 * e4m3 weights on the block-scaled MFMA v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales: twice the fp16 MFMA rate and half the
 * operand bytes.  Quantisation: value * inv_scale, round to nearest even, finite values saturated to +-448; a NaN is written as the e4m3 NaN
 * and +-Inf as NaN too (a blow-up upstream stays visible: the consuming MFMA spreads it); the consumer multiplies the fp32 accumulator by
 * a_scale (= 1 / inv_scale of its input, one per tensor) * w_scale[n].  inv_scale is the caller's: 1 clips activations beyond +-448 without
 * a signal, so a host calibrates it -- fie_amax_f16 folds max |x| of a tensor into a device float (atomic max; the caller zeroes it), run over
 * one representative edit with f16 activations, and inv_scale = 2^-k with k = ceil(log2(amax * margin / 448)) per quantised tensor (a power of
 * two: scaling is exact in both directions; fie_amd/pipe.py::calibrate_fp8 does this with margin 2).  The residual stream, q/k/v and every
 * GEMM output that is not a GEMM input stay fp16.
 *   fie_gemm_x8_f16: A8 [M, K] e4m3 bytes (row stride lda BYTES, K % 16 == 0, lda % 16 == 0); W8packed / w_scale from fie_pack_rows_f8 with
 *     ldw % 128 == 0; epilogue as fie_gemm_f16; out_f8 != 0: C is e4m3 bytes (value * out_inv_scale; ldc in BYTES; GEGLU: N / 2 bytes per row;
 *     no residual), else f16 (ldc in elements).
 *   fie_quantize_f8: plain f16 -> e4m3 conversion of a [rows, C] tensor (tests; producers without a fused form). */
int fie_gemm_x8_f16(fie_ctx* ctx, const void* A8, int64_t lda, const void* W8packed, int64_t ldw, const float* w_scale, float a_scale, void* C, int64_t ldc,
                    int M, int N, int K, const void* bias, const void* rowbias, int64_t ld_rowbias, int rows_per_batch, const void* residual, int64_t ldr,
                    float scale, int act, int out_f8, float out_inv_scale);
int fie_layernorm_f16_o8(fie_ctx* ctx, const void* X, int64_t ldx, void* Y8, int64_t ldy8, int64_t rows, int C, const void* gamma, const void* beta,
                         float eps, float inv_scale);
int fie_attention_f16_o8(fie_ctx* ctx, const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O8, int64_t ldo8,
                         int B, int H, int Tq, int Tk, int D, float scale, int causal, float inv_scale);
int fie_quantize_f8(fie_ctx* ctx, const void* X, int64_t ldx, void* Y8, int64_t ldy, int64_t rows, int C, float inv_scale);
int fie_amax_f16(fie_ctx* ctx, const void* X, int64_t ldx, int64_t rows, int C, float* amax);
/* The resnet convs of an fp8 model (upstream resnet.py: GroupNorm -> SiLU -> conv): GroupNorm writes e4m3 (the _o8 twins of fie_groupnorm_nhwc_f16 /
 * fie_groupnorm_stats_nhwc_f16; Y8 = [B, rows, C] bytes) and fie_conv3x3_x8_nhwc_f16 runs the conv view of the same block-scaled MFMA kernel: X8 NHWC
 * e4m3 with Cin % 128 == 0 (a K-step is 128 channels of one tap), W8packed from fie_pack_conv3x3_f8 with cin_pad == Cin; output f16.  Conv + 1x1 side
 * inputs and the 2x2 parity up-samplers have no fp8-activation form (their other inputs are the fp16 residual stream). */
int fie_groupnorm_nhwc_f16_o8(fie_ctx* ctx, const void* X1, int C1, const void* X2, int C2, void* Y8, int B, int64_t rows_per_image, int groups,
                              const void* gamma, const void* beta, float eps, int silu, void* workspace, float inv_scale);
int fie_groupnorm_stats_nhwc_f16_o8(fie_ctx* ctx, const void* X, int C, void* Y8, int B, int64_t rows_per_image, int groups, const void* gamma,
                                    const void* beta, float eps, int silu, const void* partial, void* workspace, int partial_groups, float inv_scale);
int fie_conv3x3_x8_nhwc_f16(fie_ctx* ctx, const void* X8, int B, int H, int W, int Cin, int upsample2x, int stride, int pad_mode, const void* W8packed,
                            int64_t ldw, const float* w_scale, float a_scale, void* Y, int64_t ldc, int Cout, const void* bias, const void* rowbias,
                            int64_t ld_rowbias, const void* residual, int64_t ldr, float scale, int act);

/* ---- fp32 path (`FastEditor(use_full_precision=True)`, run_batch.py --full_precision / --quality_mode; reference:
 * src/pipeline.py:67-71,94-99).  Same graphs, fp32 storage, exact fp32 arithmetic on v_mfma_f32_16x16x4_f32.
 *   fie_gemm_f32: as fie_gemm_f16 with plain (unpacked) weights W [N][ldw] -- or [K][ldw] when w_is_kn -- and a two-level
 *     batch (z1 < nb1, z2 < nb2; element strides sA*, sW*, sC*) that serves the attention products over (image, head).
 *   fie_conv3x3_nhwc_f32: weights [Cout][(ky, kx, ci)] (ldw >= 9*Cin).
 *   fie_softmax_rows_f32: in-place softmax(scale * S) per row, keys > (row % tq) masked when causal.
 *   the *_f32 norm / element-wise entries mirror their f16 twins with fp32 tensors. */
int fie_gemm_f32(fie_ctx* ctx, const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, const float* W,
                 int64_t ldw, int w_is_kn, float* C, int64_t ldc, int M, int N, int K, const float* bias,
                 const float* rowbias, int64_t ld_rowbias, int rows_per_batch, const float* residual, int64_t ldr,
                 float scale, int act, int nb1, int nb2, int64_t sA1, int64_t sA2, int64_t sW1, int64_t sW2, int64_t sC1,
                 int64_t sC2);
int fie_conv3x3_nhwc_f32(fie_ctx* ctx, const float* X, int B, int H, int W, int Cin, int upsample2x, int stride, int pad_mode,
                         const float* Wkc, int64_t ldw, float* Y, int64_t ldc, int Cout, const float* bias,
                         const float* rowbias, int64_t ld_rowbias, const float* residual, int64_t ldr, float scale, int act);
int fie_softmax_rows_f32(fie_ctx* ctx, float* S, int64_t rows, int cols, int64_t ld, float scale, int causal, int tq);
int fie_groupnorm_nhwc_f32(fie_ctx* ctx, const void* X1, int C1, const void* X2, int C2, void* Y, int B,
                           int64_t rows_per_image, int groups, const void* gamma, const void* beta, float eps,
                           int silu, void* workspace);
int fie_layernorm_f32(fie_ctx* ctx, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t rows, int C,
                      const void* gamma, const void* beta, float eps);
int fie_sinusoid_f32(fie_ctx* ctx, const float* vals, int B, int nvals, int dim, void* out, int64_t ld_out, int col0);
int fie_clip_embed_f32(fie_ctx* ctx, const int32_t* ids, int B, int T, int C, const void* tok_table,
                       const void* pos_table, void* out);
int fie_pixels_in_u8_f32(fie_ctx* ctx, const uint8_t* src, int H, int W, int normalize, void* dst, int copies);
int fie_pixels_out_f32_u8(fie_ctx* ctx, const void* src, int64_t ld_in, int H, int W, uint8_t* dst);
int fie_latent_prep_f32(fie_ctx* ctx, const void* moments, const float* eps_post, const float* noise, int64_t HW,
                        float scaling_factor, float sqrt_ab, float sqrt_1mab, float* latents_out, void* model_in,
                        int copies);
int fie_lcm_step_f32(fie_ctx* ctx, const void* eps, int64_t ld_eps, int nb, float guidance, float* latents,
                     const float* noise, int64_t HW, float sqrt_ab_t, float sqrt_1mab_t, float c_skip, float c_out,
                     float sqrt_ab_prev, float sqrt_1mab_prev, void* model_in, int copies, float inv_scaling,
                     void* decode_in);

/* ---- tuning / test hooks (not part of the drop-in surface).  Tile hooks are PER CTX: nothing process-global sits on the
 * launch path.  Tile codes: 1/2/3 register-staged 128x128 / 128x64 / 64x64 (any shape); 41/42/43 LDS-DMA ring 128x128 /
 * 128x64 / 64x64; 51 ring 128x128 x 8 waves; 61/62 ring 256x256 / 256x128 x 8 waves; 81 phased 256x256 (gemm8.hip).
 * + 1000 / + 2000: force n-tiles / m-tiles fastest tile order (plain codes estimate the order that re-streams fewer bytes).  A code the shape is not eligible for returns FIE_EINVAL from the op. */
int fie_debug_force_tile(fie_ctx* ctx, int tile);                  /* 0 = heuristic */
int fie_debug_tile_override(fie_ctx* ctx, const char* spec);       /* "mode,M,N,K=code;..." (mode 0 GEMM, 1 conv); NULL clears; returns the count */
/* Per-shape tile autotune.  on = 1: the first EAGER launch of every GEMM / 3x3-conv problem (M, N, K, geometry, weight type) times
 * the eligible tile kernels on a scratch output (the caller's C is written once, by the winner), each launch with the weights
 * cold (a 384 MB flush) and the activations warm, as a layer inside the network sees them, and the context remembers the
 * fastest; launches under stream capture or program recording never tune, they use what is remembered, else the built-in rule.
 * Every tile accumulates K in the same order, so the choice does not change results.  on = 2 keeps using what is remembered but
 * tunes nothing new and frees the scratch (0.4 GB + one output); on = 0 returns to the built-in rule.  fie_gemm_autotune_report writes one "gemm|conv M= N= K= K1= geom= w8= -> code" line per remembered problem
 * into buf (NUL-terminated, truncated to cap) and returns the number of problems. */
/* A resnet's second conv together with its 1x1 shortcut (upstream models/resnet.py ResnetBlock2D.forward: conv2(h) + conv_shortcut(input))
 * as ONE GEMM: Y = conv3x3(X) + [X2 | X3] W1x1^T (+ bias, row bias, activation), Wpacked rows = [9 * Cin taps | C2 | C3] (the conv's packed
 * matrix with the shortcut's columns appended; bias = the two biases added).  X [B, H, W, Cin]; X2 [B*H*W, C2] and the optional X3
 * [B*H*W, C3] (row strides ld2 / ld3) are the shortcut's input, or its two halves when that is a concat [x | skip]; Cin, C2, C3 % 64 == 0.
 * The shortcut GEMM, its [M, Cout] output and the residual read disappear; the sum is rounded once. */
int fie_conv3x3_plus_nhwc_f16(fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, const void* Wpacked, int64_t ldw, void* Y, int64_t ldc, int Cout,
                              const void* bias, const void* rowbias, int64_t ld_rowbias, float scale, int act, const void* X2, int64_t ld2, int C2,
                              const void* X3, int64_t ld3, int C3);

/* conv3x3 on the nearest-2x upsampled input (upstream models/upsampling.py Upsample2D: F.interpolate(scale_factor=2, mode="nearest") then
 * conv) computed as four 2x2 convs on the input itself, one per output parity, with the weights of the taps that coincide on one input
 * pixel summed beforehand: the same sums (zero padding included) from 4 instead of 9 multiply-adds per output.  W4 = the four packed
 * matrices [py * 2 + px][Npad][ldw] with K index (a * 2 + b) * Cin + ci (fie_amd/hip.py: pack_conv_up2x builds them from the OIHW
 * weights).  X [B, H, W, Cin] (Cin % 64 == 0) -> Y [B, 2H, 2W, ldc]; epilogue options as fie_conv3x3_nhwc_f16 without a residual; an
 * armed fie_gn_stats_target (rows_per_image = 4 H W) is honoured.  Differs from the 9-tap form by the f16 rounding of the summed weights. */
int fie_conv_up2x_nhwc_f16(fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, const void* W4, int64_t ldw, int Npad, void* Y, int64_t ldc,
                           int Cout, const void* bias, const void* rowbias, int64_t ld_rowbias, float scale, int act);

/* GroupNorm statistics from the producer's epilogue (upstream: the first of the three passes torch.nn.GroupNorm makes over the tensor,
 * models/resnet.py norm1 / norm2, models/autoencoders/vae.py).  fie_gn_stats_target arms the NEXT fie_gemm_f16 / fie_conv3x3_nhwc_f16
 * launch on this context (one shot): besides its output [M, N] it writes, per image (rows_per_image rows, a multiple of 32), per 32-row
 * granule and per group, the sum and the sum of squares of the f16-ROUNDED outputs: partial[B][rows_per_image / 32][groups][2] floats
 * (fie_gn_stats_bytes).  N / groups must be 4, 8 or 16 channels (the SDXL VAE's 128 / 256 / 512-channel maps with 32 groups); no GEGLU.
 * Every slot has one writer: deterministic.  fie_groupnorm_stats_nhwc_f16 is fie_groupnorm_nhwc_f16 for such a tensor: it reduces the
 * partials (fp64) and applies, reading X once instead of twice; workspace as for fie_groupnorm_nhwc_f16.
 * Group widths that are not 4 / 8 / 16 but a multiple of 4 (the UNet's 20 / 40 channels per group at 640 / 1280 channels): arm the producer
 * with groups = N / 4 (it then writes one slot per 4-channel QUAD, which is all a lane of the MFMA layout can sum without crossing a group
 * border) and pass partial_groups = N / 4 to fie_groupnorm_stats_nhwc_f16, which sums the N / 4 / groups quads of each real group
 * (partial_groups 0 or == groups: slots are groups, as above). */
int fie_gn_stats_target(fie_ctx* ctx, void* partial, int64_t rows_per_image, int groups);
/* GroupNorm (+ SiLU) FUSED INTO THE CONSUMING 3x3 CONV (round 4; upstream models/resnet.py: norm1 -> nonlinearity -> conv1, norm2 -> nonlinearity -> conv2).
 * fie_groupnorm_coef_f16 finishes the statistics a producer's epilogue left (as fie_groupnorm_stats_nhwc_f16 does) but writes, instead of a normalised
 * tensor, coef[B][C][2] floats: (sc, sh) = (rstd * gamma, beta - mean * rstd * gamma).  fie_conv3x3_gn_nhwc_f16 is the same-size stride-1 3x3 conv of
 * silu(X * sc + sh) (silu optional): the halo-resident kernel applies the expression to every 64-channel chunk of a 16x16 patch's halo while it sits in LDS,
 * behind the MFMAs of the chunk before; padding pixels stay zero (the conv pads the NORMALISED tensor).  Same fp32 expression as the apply kernel: the
 * result has the bits of fie_groupnorm_stats_nhwc_f16 followed by fie_conv3x3_nhwc_f16 on tile code 72, without the read + write of the normalised tensor.
 * Built for what fie_conv3x3_gn_ok returns 1 for (one image, H, W % 16 == 0, Cin in whole 64-channel chunks from 128 to 1024, Cout % 128 == 0) WITH a
 * fie_gn_stats_target armed for the OUTPUT at 4 / 8 / 16 channels per group (out_groups) -- every resnet conv of the VAE; elsewhere the caller keeps the two
 * launches.  workspace as for fie_groupnorm_nhwc_f16. */
int fie_groupnorm_coef_f16(fie_ctx* ctx, int C, int B, int64_t rows_per_image, int groups, const void* gamma, const void* beta, float eps, const void* partial,
                           void* workspace, int partial_groups, float* coef);
int fie_conv3x3_gn_ok(fie_ctx* ctx, int B, int H, int W, int Cin, int Cout, int out_groups);
int fie_conv3x3_gn_nhwc_f16(fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, const float* coef, int silu, const void* Wpacked, int64_t ldw, void* Y,
                            int64_t ldc, int Cout, const void* bias, const void* residual, int64_t ldr);
int64_t fie_gn_stats_bytes(int B, int64_t rows_per_image, int groups);
int fie_groupnorm_stats_nhwc_f16(fie_ctx* ctx, const void* X, int C, void* Y, int B, int64_t rows_per_image, int groups, const void* gamma,
                                 const void* beta, float eps, int silu, const void* partial, void* workspace, int partial_groups);

/* Touches one dword of every 128-byte line of [ptr, ptr + bytes) with `blocks` small workgroups on `stream` (NULL: the context's): pulls a
 * weight matrix from HBM into the Infinity Cache ahead of the kernel that will stream it.  Reads only; no result. */
int fie_prefetch(fie_ctx* ctx, const void* ptr, int64_t bytes, void* stream, int blocks);
/* Split-K for the GEMM / conv ring kernels (M = 2048-class problems whose big tiles leave most CUs idle): binds a caller-owned device
 * workspace to the context -- 16 KiB of arrival counters that MUST be zero when bound (every launch leaves them zero again) followed by
 * the fp32 partial-tile slabs.  Launches that run concurrently (different streams) need different workspaces: bind the stream's own
 * before issuing on it -- that holds for the op entries, for the *_forward_f16 walks and for recorded programs / captured graphs, which
 * keep the workspace that was bound when they were recorded (see ORDERING CONTRACT at fie_program_run).  NULL unbinds (no launch splits
 * K then).  The counters are self-healing (a launch takes its arrivals off again with one atomic subtract), so a violation of the rule
 * corrupts only the overlapping launches' sums, never later ones.  A launch splits only where the autotuner measured it faster, or an
 * override asks (tile codes: split factor * 10000 + code); it falls back to the unsplit kernel when the workspace is too small.
 * Determinism: slices are summed in slice order by the block that arrives last, so results do not depend on timing. */
int fie_splitk_workspace(fie_ctx* ctx, void* workspace, int64_t bytes);
int fie_debug_splitk(fie_ctx* ctx, int mode);                       /* 0 = never split K, 1 = default */
int fie_gemm_autotune(fie_ctx* ctx, int on);
int fie_gemm_autotune_report(fie_ctx* ctx, char* buf, int cap);
/* Remembered choices from a report's text (the inverse of fie_gemm_autotune_report; unknown lines are skipped): a host saves the report of a
 * tuned process and loads it into the next one -- no timing launches at start-up and the same kernels on every box (a split-K choice moves
 * the last f16 bit).  FIE_TUNE_TABLE=<file> makes fie_amd load it into every context.  Returns the number of problems loaded. */
int fie_gemm_autotune_load(fie_ctx* ctx, const char* text);
int fie_debug_tune_exclude(fie_ctx* ctx, const char* codes);        /* A/B hook (tools/tuner_ab.py): comma-separated tile codes the tuner must not offer (10000 = every split-K variant; NULL / "" = none); forgets every remembered choice; returns the number of codes parsed */
int fie_debug_gemm_probe(fie_ctx* ctx, int mode);                  /* TIMING-ONLY probes of the LDS-DMA kernels (outputs are wrong): 0 off, 1 = DMA loads dropped by the descriptor, 2 = every tile loads tile (0,0), 3 = ring kernels: no DMA issued in the K loop, 4 = no epilogue */
int fie_debug_epilogue_prefetch(fie_ctx* ctx, int on);            /* A/B switch (default on): the ring GEMM / conv kernels load the bias row and the residual tile BEFORE the K loop; results are identical either way */
int fie_debug_gemm_stamps(fie_ctx* ctx, void* buf);                  /* device buffer for the stamped ring kernels (tile codes 97 / 98): per tile and wave 8 uint32 cycle sums -- [0] drain + barrier, [1]/[4] DMA issue, [2]/[5] fragment reads, [3]/[6] MFMA issue (code 98: [0] = whole K-steps); NULL detaches */
const char* fie_debug_last_gemm_kernel(fie_ctx* ctx);              /* kernel / tile of the last fie_gemm_f16 / fie_conv3x3_nhwc_f16 launch */
int fie_debug_oplog(fie_ctx* ctx, int on);                         /* launch log for the per-shape profile (tools/shape_profile.py): while on, every launch appends "kernel symbol|blocks|threads|LDS bytes|op description (shape, tile code, algorithmic flop / bytes)" */
int fie_debug_oplog_mark(fie_ctx* ctx, const char* text);           /* appends "#text" (a stage boundary) when the log is on */
int64_t fie_debug_oplog_read(fie_ctx* ctx, char* buf, int64_t cap);  /* newline-joined log into buf when it fits; returns its length (cap 0: size query) */
int fie_debug_attn_variant(fie_ctx* ctx, int variant);   /* per context; 0 = default kernel, 1 = first-generation kernel, 2 / 3 = 128 / 64 queries per block forced, 4 = 64 queries per block as two waves x 32 (A/B benchmarking) */
int fie_debug_gn_onepass(fie_ctx* ctx, int enable);      /* per context; 1 = default (single-pass GroupNorm on small maps), 0 = always partial/finalize/apply */

/* ---- K11 Canny on the host (integer exact).  Replaces cv2.cvtColor(RGB2GRAY) + cv2.Canny at
 * src/pipeline.py:200,205 and the 3-channel stack at :208.  rgb, edges_rgb: host u8 [H, W, 3]. */
int fie_canny_rgb_u8(const uint8_t* rgb, int H, int W, int low, int high, uint8_t* edges_rgb);

/* ---- K11 on the device: same integer-exact result as fie_canny_rgb_u8, on a u8 HWC image already in HBM.
 *   workspace: fie_canny_workspace_bytes(H, W) bytes.  The hysteresis fixed point reads a flag back, so this entry
 *   SYNCHRONISES the ctx stream (it runs in the host-side preparation of an edit, never inside the captured graph).
 *   iterations (optional, host int): hysteresis passes launched. */
int64_t fie_canny_workspace_bytes(int H, int W);
int fie_canny_rgb_device_u8(fie_ctx* ctx, const uint8_t* rgb, int H, int W, int low, int high, void* workspace,
                            uint8_t* edges_rgb, int* iterations);
/* The same in two halves, so that the host need not wait for the hysteresis flag before it goes on: begin launches NMS, `rounds` rounds of four
 * hysteresis passes, the edge map of that state and an asynchronous copy of the LAST round's four flag words into host_flags (4 ints of PINNED host
 * memory) and returns without waiting; finish synchronises the stream, runs further rounds only if that round's last pass still changed something
 * (a weak chain that crosses more than 4 * rounds - 1 tiles of 32x32), rewrites the edge map then, and reports the passes IT launched (0: the map begin
 * wrote was final).  fie_amd's FastEditor.edit() begins with 4 rounds, issues the whole edit behind it and calls finish at the edit's own final
 * synchronisation: no host wait at all in the common case, a repeated device job in the rare one.  Same result as the one-call form. */
int fie_canny_rgb_device_begin_u8(fie_ctx* ctx, const uint8_t* rgb, int H, int W, int low, int high, int rounds, void* workspace, uint8_t* edges_rgb,
                                  int* host_flags);
int fie_canny_rgb_device_finish_u8(fie_ctx* ctx, int H, int W, void* workspace, uint8_t* edges_rgb, int* host_flags, int* iterations);

/* ---- K13 LANCZOS resize on the device, bit-exact with Pillow's 8-bit resample.  Replaces
 * `image.resize((1024, 1024), Image.LANCZOS)` at src/pipeline.py:251 (the PIL image is uploaded at its own size instead).
 *   src u8 [H, W, 3] -> dst u8 [OH, OW, 3] on the device; kx/ky: int32 [OW|OH, ks] 22-bit fixed-point taps, bx/by: int32
 *   [OW|OH, 2] = (first input index, tap count) -- Pillow's precompute_coeffs / normalize_coeffs_8bpc, restated in
 *   fie_amd/resize.py; a table may be NULL when that axis keeps its size.  tmp: u8 [H, OW, 3] scratch (both axes change).
 *   Asynchronous on the ctx stream. */
int fie_resize_rgb_u8(fie_ctx* ctx, const uint8_t* src, int H, int W, uint8_t* dst, int OH, int OW, const int* kx,
                      const int* bx, int ksx, const int* ky, const int* by, int ksy, uint8_t* tmp);

#ifdef __cplusplus
}
#endif
#endif /* FIE_H_ */
