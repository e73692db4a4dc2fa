// K7-K10, K12-embed: small element-wise kernels of the denoising loop (include/fie.h).  All are latency/HBM-bound;
// scheduler scalars are computed on the host in fp32/fp64 and passed by value (c_skip ~ 1e-8 underflows
// in fp16, so the LCM update is done in fp32 on an fp32 master copy of the latents).
#include "fie_internal.h"

namespace {

template <typename T>
__global__ void sinusoid_kernel(const float* vals, int B, int nvals, int dim, T* out, int64_t ld_out, int col0) {
    const int half = dim / 2;
    const int total = B * nvals * half;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int f = i % half;
    const int v = (i / half) % nvals;
    const int b = i / (half * nvals);
    // embeddings.py::get_timestep_embedding, flip_sin_to_cos=True, downscale_freq_shift=0: [cos | sin]
    const float freq = expf(-9.210340371976184f * (float)f / (float)half);
    const float arg = vals[b * nvals + v] * freq;
    T* row = out + (int64_t)b * ld_out + col0 + v * dim;
    row[f] = (T)cosf(arg);
    row[half + f] = (T)sinf(arg);
}

// K7, fused (north_star: "timestep-embedding fused into wavefront-level HIP kernels"): per denoising step
//     out[b, :] = SiLU( W2 SiLU(W1 sinus(t_b) + b1) + b2 + add[b, :] )
// = Timesteps(C0, flip_sin_to_cos, shift 0) -> TimestepEmbedding (Linear, SiLU, Linear) -> + text-time embedding -> the SiLU every
// resnet applies in front of its time projection (upstream embeddings.py, resnet.py) -- ONE launch instead of sinusoid + two GEMM
// launches whose M is the batch (2 rows).  E / 64 workgroups; workgroup i owns neurons [64 i, 64 i + 64) of BOTH layers, a thread
// owns a quarter of one neuron's K range (all its 16-byte weight loads are independent and in flight together; the four quarters
// meet by two shuffles).  Between the layers the workgroups exchange the hidden vector through `ws` behind a counter barrier
// (cdna_hip_programming.md Guideline 16: every storing wave drains, one agent-scope release + relaxed arrive per workgroup, relaxed
// poll + ONE agent-scope acquire; E / 64 <= 256 workgroups are co-resident by grid size; the spin is bounded, and a workgroup that
// gives up says so: sync[2] of the workspace and the context's device error word, fie_ctx_error_flag, become non-zero).  The last workgroup
// to leave resets the two counters, so the zero-initialised workspace is clean for the next launch on the same stream.
constexpr int kTeMaxB = 4;

__global__ __launch_bounds__(256) void time_embed_kernel(const float* t, int B, int C0, int E, const half_t* W1, const half_t* b1,
                                                         const half_t* W2, const half_t* b2, const half_t* add, int64_t ld_add,
                                                         half_t* out, int64_t ld_out, half_t* hbuf, unsigned* sync, unsigned* err) {
    extern __shared__ __attribute__((aligned(16))) half_t te_smem[];
    half_t* x = te_smem;                       // [B][C0]   sinusoid, storage type (as the unfused path stores it)
    half_t* h = te_smem + kTeMaxB * 512;       // [B][E]    hidden vector after the barrier
    const int tid = threadIdx.x;
    const int nl = tid >> 2, kq = tid & 3;     // neuron inside the workgroup, K quarter
    const int n = blockIdx.x * 64 + nl;
    const int half_c = C0 / 2;
    for (int i = tid; i < B * half_c; i += 256) {
        const int b = i / half_c, f = i - b * half_c;
        const float arg = t[b] * expf(-9.210340371976184f * (float)f / (float)half_c);
        x[b * C0 + f] = (half_t)cosf(arg);
        x[b * C0 + half_c + f] = (half_t)sinf(arg);
    }
    __syncthreads();
    auto quarter_dot = [&](const half_t* wrow, const half_t* vec, int K, float (&acc)[kTeMaxB]) {
        const int k0 = kq * (K / 4), k1 = k0 + K / 4;                  // K % 32 == 0: whole 16-byte vectors per quarter
#pragma unroll
        for (int b = 0; b < kTeMaxB; ++b) acc[b] = 0.f;
#pragma unroll 8
        for (int k = k0; k < k1; k += 8) {
            float w[8];
            fie_load8(wrow + k, w);
#pragma unroll
            for (int b = 0; b < kTeMaxB; ++b)
                if (b < B) {
                    float v[8];
                    fie_load8(vec + b * K + k, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[b] += w[j] * v[j];
                }
        }
#pragma unroll
        for (int b = 0; b < kTeMaxB; ++b) {
            acc[b] += __shfl_xor(acc[b], 1);
            acc[b] += __shfl_xor(acc[b], 2);
        }
    };
    float acc[kTeMaxB];
    // ---- layer 1: this workgroup's 64 neurons -> hbuf
    if (n < E) {
        quarter_dot(W1 + (int64_t)n * C0, x, C0, acc);
        if (kq == 0) {
            const float bias = (float)b1[n];
#pragma unroll
            for (int b = 0; b < kTeMaxB; ++b)
                if (b < B) hbuf[b * E + n] = (half_t)fie_silu(acc[b] + bias);
        }
    }
    // ---- barrier across the E / 64 workgroups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // every storing wave drains its stores
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(2);
        if (spins >= (1u << 22)) {             // gave up: the hidden vector may be incomplete.  Sticky words the host reads at its next sync
            __hip_atomic_store(&sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (err) __hip_atomic_store(err, FIE_DEVERR_TIME_EMBED_BARRIER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int i = tid * 8; i < B * E; i += 256 * 8) {                    // hidden vector -> LDS (plain loads behind the acquire)
        float v[8];
        fie_load8(hbuf + i, v);
        fie_store8(h + i, v);
    }
    __syncthreads();
    // ---- layer 2: the same 64 neuron indices of the output
    if (n < E) {
        quarter_dot(W2 + (int64_t)n * E, h, E, acc);
        if (kq == 0) {
            const float bias = (float)b2[n];
#pragma unroll
            for (int b = 0; b < kTeMaxB; ++b)
                if (b < B) out[(int64_t)b * ld_out + n] = (half_t)fie_silu(acc[b] + bias + (add ? (float)add[(int64_t)b * ld_add + n] : 0.f));
        }
    }
    // ---- leave: the last workgroup out resets the counters (all have passed the poll by then: each left only after seeing the full count)
    if (tid == 0) {
        const unsigned left = __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == gridDim.x - 1) {
            __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <typename E>
__global__ void clip_embed_kernel(const int32_t* ids, int rows, int T, int C, const E* tok, const E* pos, E* out) {
    const int nch = C >> 3;
    const int64_t total = (int64_t)rows * nch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / nch), ch = (int)(i - (int64_t)r * nch);
        const int t = r % T;
        float a[8], b[8], o[8];
        fie_load8(tok + (int64_t)ids[r] * C + ch * 8, a);
        fie_load8(pos + (int64_t)t * C + ch * 8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = a[j] + b[j];
        fie_store8(out + (int64_t)r * C + ch * 8, o);
    }
}

template <typename T>
__global__ void pixels_in_kernel(const uint8_t* src, int64_t npix, int normalize, T* dst, int copies) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float x = (float)src[i * 3 + c] / 255.0f;      // VaeImageProcessor: /255 then 2x-1
            if (normalize) x = 2.0f * x - 1.0f;
            o[c] = x;
        }
        for (int k = 0; k < copies; ++k) fie_store8(dst + ((int64_t)k * npix + i) * 8, o);
    }
}

template <typename T>
__global__ void pixels_out_kernel(const T* src, int64_t ld, int64_t npix, uint8_t* dst) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float x = (float)src[i * ld + c] * 0.5f + 0.5f;
            x = fminf(fmaxf(x, 0.f), 1.f);
            dst[i * 3 + c] = (uint8_t)rintf(x * 255.0f);   // numpy .round(): half to even
        }
    }
}

template <typename T>
__global__ void latent_prep_kernel(const T* moments, const float* eps_post, const float* noise, int64_t HW, float sf,
                                   float sqrt_ab, float sqrt_1mab, float* lat, T* model_in, int copies) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x) {
        float m[8], o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        fie_load8(moments + i * 8, m);
        float4 l;
        float* lp = &l.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float logvar = fminf(fmaxf(m[4 + c], -30.f), 20.f);
            const float z0 = (m[c] + expf(0.5f * logvar) * eps_post[c * HW + i]) * sf;
            const float x = sqrt_ab * z0 + sqrt_1mab * noise[c * HW + i];
            lp[c] = x;
            o[c] = x;
        }
        *reinterpret_cast<float4*>(lat + i * 4) = l;
        for (int k = 0; k < copies; ++k) fie_store8(model_in + ((int64_t)k * HW + i) * 8, o);
    }
}

template <typename T>
struct LcmArgs {
    const T* eps; int64_t ld_eps; int nb; float guidance;
    float* lat; const float* noise; int64_t HW;
    float sab_t, s1mab_t, c_skip, c_out, sab_p, s1mab_p;
    T* model_in; int copies; float inv_sf; T* decode_in;
};

template <typename T>
__global__ void lcm_step_kernel(LcmArgs<T> p) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < p.HW; i += (int64_t)gridDim.x * blockDim.x) {
        float e0[4], e1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            e0[c] = (float)p.eps[i * p.ld_eps + c];
            e1[c] = p.nb == 2 ? (float)p.eps[(p.HW + i) * p.ld_eps + c] : e0[c];
        }
        float4 l = *reinterpret_cast<const float4*>(p.lat + i * 4);
        float* lp = &l.x;
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0}, od[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float e = e0[c];
            if (p.nb == 2) e = e + p.guidance * (e1[c] - e);              // eps_u + g (eps_c - eps_u)
            const float x = lp[c];
            const float x0 = (x - p.s1mab_t * e) / p.sab_t;
            float den = p.c_out * x0 + p.c_skip * x;
            if (p.noise) den = p.sab_p * den + p.s1mab_p * p.noise[c * p.HW + i];
            lp[c] = den;
            o[c] = den;
            od[c] = den * p.inv_sf;
        }
        *reinterpret_cast<float4*>(p.lat + i * 4) = l;
        if (p.model_in)
            for (int k = 0; k < p.copies; ++k) fie_store8(p.model_in + ((int64_t)k * p.HW + i) * 8, o);
        if (p.decode_in) fie_store8(p.decode_in + i * 8, od);
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (unsigned)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

template <typename T>
int sinusoid_t(fie_ctx* ctx, const float* vals, int B, int nvals, int dim, void* out, int64_t ld_out, int col0) {
    FIE_REQUIRE(ctx && vals && out, "fie_sinusoid: NULL argument");
    FIE_REQUIRE(B > 0 && nvals > 0 && dim > 0 && dim % 2 == 0 && ld_out >= col0 + nvals * dim, "fie_sinusoid: bad shape");
    const int total = B * nvals * (dim / 2);
    fie_launch(ctx, sinusoid_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, vals, B, nvals, dim, (T*)out, ld_out, col0);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

template <typename T>
int clip_embed_t(fie_ctx* ctx, const int32_t* ids, int B, int Tn, int C, const void* tok, const void* pos, void* out) {
    FIE_REQUIRE(ctx && ids && tok && pos && out, "fie_clip_embed: NULL argument");
    FIE_REQUIRE(B > 0 && Tn > 0 && C > 0 && C % 8 == 0, "fie_clip_embed: bad shape");
    fie_launch(ctx, clip_embed_kernel<T>, dim3(grid_for((int64_t)B * Tn * C / 8)), dim3(256), 0, ids, B * Tn, Tn, C, (const T*)tok, (const T*)pos, (T*)out);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

template <typename T>
int pixels_in_t(fie_ctx* ctx, const uint8_t* src, int H, int W, int normalize, void* dst, int copies) {
    FIE_REQUIRE(ctx && src && dst && H > 0 && W > 0 && copies > 0, "fie_pixels_in: bad argument");
    const int64_t n = (int64_t)H * W;
    fie_launch(ctx, pixels_in_kernel<T>, dim3(grid_for(n)), dim3(256), 0, src, n, normalize, (T*)dst, copies);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

template <typename T>
int pixels_out_t(fie_ctx* ctx, const void* src, int64_t ld_in, int H, int W, uint8_t* dst) {
    FIE_REQUIRE(ctx && src && dst && H > 0 && W > 0 && ld_in >= 3, "fie_pixels_out: bad argument");
    const int64_t n = (int64_t)H * W;
    fie_launch(ctx, pixels_out_kernel<T>, dim3(grid_for(n)), dim3(256), 0, (const T*)src, ld_in, n, dst);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

template <typename T>
int latent_prep_t(fie_ctx* ctx, const void* moments, const float* eps_post, const float* noise, int64_t HW, float sf, float sqrt_ab,
                  float sqrt_1mab, float* latents_out, void* model_in, int copies) {
    FIE_REQUIRE(ctx && moments && eps_post && noise && latents_out && model_in && HW > 0 && copies > 0, "fie_latent_prep: bad argument");
    fie_launch(ctx, latent_prep_kernel<T>, dim3(grid_for(HW)), dim3(256), 0, (const T*)moments, eps_post, noise, HW, sf, sqrt_ab, sqrt_1mab, latents_out, (T*)model_in, copies);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

template <typename T>
int lcm_step_t(fie_ctx* ctx, const void* eps, int64_t ld_eps, int nb, float guidance, float* latents, const float* noise, int64_t HW,
               float sqrt_ab_t, float sqrt_1mab_t, float c_skip, float c_out, float sqrt_ab_prev, float sqrt_1mab_prev, void* model_in,
               int copies, float inv_scaling, void* decode_in) {
    FIE_REQUIRE(ctx && eps && latents && HW > 0, "fie_lcm_step: bad argument");
    FIE_REQUIRE(nb == 1 || nb == 2, "fie_lcm_step: nb=%d (1 or 2)", nb);
    FIE_REQUIRE(ld_eps % 4 == 0 && ld_eps >= 4, "fie_lcm_step: ld_eps must be a multiple of 4");
    FIE_REQUIRE(sqrt_ab_t > 0.f, "fie_lcm_step: sqrt(alpha_bar_t) must be positive");
    LcmArgs<T> p = {(const T*)eps, ld_eps, nb, guidance, latents, noise, HW, sqrt_ab_t, sqrt_1mab_t, c_skip, c_out,
                    sqrt_ab_prev, sqrt_1mab_prev, (T*)model_in, copies, inv_scaling, (T*)decode_in};
    fie_launch(ctx, lcm_step_kernel<T>, dim3(grid_for(HW)), dim3(256), 0, p);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

}  // namespace

// One dword of every 128-B line of [p, p + bytes): pulls the range from HBM into the Infinity Cache (and this XCD's L2) ahead of the
// kernel that will stream it.  The loads are never consumed; the asm keeps them.
__global__ __launch_bounds__(256) void prefetch_kernel(const unsigned* p, size_t lines) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < lines; i += (size_t)gridDim.x * 256) acc += p[i * 32];
    asm volatile("" ::"v"(acc));
}

// out = a + b over n f16 values (n % 8 == 0): 16 B per lane and operand (the ControlNet residual adds of the C++ UNet walk, csrc/graphs.cpp;
// the Python walk fuses them into the zero-conv epilogues instead)
__global__ __launch_bounds__(256) void add_kernel(const half_t* a, const half_t* b, half_t* out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float x[8], y[8];
        fie_load8(a + i * 8, x);
        fie_load8(b + i * 8, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        fie_store8(out + i * 8, x);
    }
}

// max |x| over a [rows, C] f16 tensor, folded into *out with an atomic max on the bit pattern (non-negative floats order like unsigned integers):
// the one-pass activation-scale calibration of the fp8 configuration (fie_amd/pipe.py: calibrate_fp8).  NaN / Inf are skipped.
__global__ __launch_bounds__(256) void amax_kernel(const half_t* x, int64_t ldx, int64_t rows, int cols8, unsigned* out) {
    const int64_t n = rows * cols8;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols8;
        const int c = (int)(i - r * cols8) * 8;
        float v[8];
        fie_load8(x + r * ldx + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float av = fabsf(v[j]);
            if (av <= 65504.f) m = fmaxf(m, av);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// dst[r, 0..cols) = src[r, 0..cols) for `rows` rows (cols % 8 == 0; row strides in elements)
__global__ __launch_bounds__(256) void copy_rows_kernel(const half_t* src, int64_t lds_, half_t* dst, int64_t ldd, int rows, int cols8) {
    const int64_t n = (int64_t)rows * cols8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cols8), c = (int)(i - (int64_t)r * cols8);
        *reinterpret_cast<uint4*>(dst + r * ldd + c * 8) = *reinterpret_cast<const uint4*>(src + r * lds_ + c * 8);
    }
}

extern "C" {

int fie_add_f16(fie_ctx* ctx, const void* a, const void* b, void* out, int64_t n) {
    FIE_REQUIRE(ctx && a && b && out && n > 0 && n % 8 == 0, "fie_add_f16: bad argument");
    const int64_t n8 = n / 8, blocks = (n8 + 255) / 256;
    fie_launch(ctx, add_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (const half_t*)a, (const half_t*)b, (half_t*)out, n8);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

int fie_amax_f16(fie_ctx* ctx, const void* X, int64_t ldx, int64_t rows, int C, float* amax) {
    FIE_REQUIRE(ctx && X && amax && rows > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0, "fie_amax_f16: bad argument");
    const int64_t n = rows * (C / 8), blocks = (n + 255) / 256;
    fie_launch(ctx, amax_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (const half_t*)X, ldx, rows, C / 8, reinterpret_cast<unsigned*>(amax));
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

int fie_copy_rows_f16(fie_ctx* ctx, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int cols) {
    FIE_REQUIRE(ctx && src && dst && rows > 0 && cols > 0 && cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, "fie_copy_rows_f16: bad argument");
    const int64_t n = (int64_t)rows * (cols / 8), blocks = (n + 255) / 256;
    fie_launch(ctx, copy_rows_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (const half_t*)src, ld_src, (half_t*)dst, ld_dst, rows, cols / 8);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

int fie_prefetch(fie_ctx* ctx, const void* ptr, int64_t bytes, void* stream, int blocks) {
    FIE_REQUIRE(ctx && ptr && bytes > 0 && blocks > 0 && blocks <= 4096, "fie_prefetch: bad argument");
    const size_t lines = (size_t)bytes / 128;
    if (lines == 0) return FIE_OK;
    hipLaunchKernelGGL(prefetch_kernel, dim3((unsigned)blocks), dim3(256), 0, stream ? (hipStream_t)stream : ctx->stream, (const unsigned*)ptr, lines);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

int fie_sinusoid_f16(fie_ctx* ctx, const float* vals, int B, int nvals, int dim, void* out, int64_t ld_out, int col0) {
    return sinusoid_t<half_t>(ctx, vals, B, nvals, dim, out, ld_out, col0);
}
int fie_sinusoid_f32(fie_ctx* ctx, const float* vals, int B, int nvals, int dim, void* out, int64_t ld_out, int col0) {
    return sinusoid_t<float>(ctx, vals, B, nvals, dim, out, ld_out, col0);
}
int64_t fie_time_embed_workspace_bytes(int E) { return (int64_t)kTeMaxB * E * (int64_t)sizeof(half_t) + 16; }

int fie_time_embed_f16(fie_ctx* ctx, const float* t, int B, int C0, int E, const void* W1, const void* b1, const void* W2, const void* b2,
                       const void* add, int64_t ld_add, void* out, int64_t ld_out, void* workspace) {
    FIE_REQUIRE(ctx && t && W1 && b1 && W2 && b2 && out && workspace, "fie_time_embed_f16: NULL argument");
    FIE_REQUIRE(B > 0 && B <= kTeMaxB && C0 > 0 && C0 % 32 == 0 && C0 <= 512 && E > 0 && E % 64 == 0 && E <= 256 * 64 && ld_out >= E &&
                    (!add || ld_add >= E),
                "fie_time_embed_f16: bad shape (B <= %d, C0 %% 32 == 0 and <= 512, E %% 64 == 0)", kTeMaxB);
    const int lds = kTeMaxB * (512 + E) * (int)sizeof(half_t);
    FIE_REQUIRE(lds <= 64 * 1024, "fie_time_embed_f16: E too large");
    half_t* hbuf = (half_t*)workspace;
    unsigned* sync = (unsigned*)((char*)workspace + (int64_t)kTeMaxB * E * sizeof(half_t));      // must be ZERO before the first launch
    fie_launch(ctx, time_embed_kernel, dim3(E / 64), dim3(256), (unsigned)lds, t, B, C0, E, (const half_t*)W1, (const half_t*)b1,
               (const half_t*)W2, (const half_t*)b2, (const half_t*)add, ld_add, (half_t*)out, ld_out, hbuf, sync, ctx->err_flag);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}
int fie_clip_embed_f16(fie_ctx* ctx, const int32_t* ids, int B, int T, int C, const void* tok_table, const void* pos_table, void* out) {
    return clip_embed_t<half_t>(ctx, ids, B, T, C, tok_table, pos_table, out);
}
int fie_clip_embed_f32(fie_ctx* ctx, const int32_t* ids, int B, int T, int C, const void* tok_table, const void* pos_table, void* out) {
    return clip_embed_t<float>(ctx, ids, B, T, C, tok_table, pos_table, out);
}
int fie_pixels_in_u8_f16(fie_ctx* ctx, const uint8_t* src, int H, int W, int normalize, void* dst, int copies) {
    return pixels_in_t<half_t>(ctx, src, H, W, normalize, dst, copies);
}
int fie_pixels_in_u8_f32(fie_ctx* ctx, const uint8_t* src, int H, int W, int normalize, void* dst, int copies) {
    return pixels_in_t<float>(ctx, src, H, W, normalize, dst, copies);
}
int fie_pixels_out_f16_u8(fie_ctx* ctx, const void* src, int64_t ld_in, int H, int W, uint8_t* dst) {
    return pixels_out_t<half_t>(ctx, src, ld_in, H, W, dst);
}
int fie_pixels_out_f32_u8(fie_ctx* ctx, const void* src, int64_t ld_in, int H, int W, uint8_t* dst) {
    return pixels_out_t<float>(ctx, src, ld_in, H, W, dst);
}
int fie_latent_prep(fie_ctx* ctx, const void* moments, const float* eps_post, const float* noise, int64_t HW,
                    float scaling_factor, float sqrt_ab, float sqrt_1mab, float* latents_out, void* model_in, int copies) {
    return latent_prep_t<half_t>(ctx, moments, eps_post, noise, HW, scaling_factor, sqrt_ab, sqrt_1mab, latents_out, model_in, copies);
}
int fie_latent_prep_f32(fie_ctx* ctx, const void* moments, const float* eps_post, const float* noise, int64_t HW,
                        float scaling_factor, float sqrt_ab, float sqrt_1mab, float* latents_out, void* model_in, int copies) {
    return latent_prep_t<float>(ctx, moments, eps_post, noise, HW, scaling_factor, sqrt_ab, sqrt_1mab, latents_out, model_in, copies);
}
int fie_lcm_step(fie_ctx* ctx, const void* eps, int64_t ld_eps, int nb, float guidance, float* latents, const float* noise, int64_t HW,
                 float sqrt_ab_t, float sqrt_1mab_t, float c_skip, float c_out, float sqrt_ab_prev, float sqrt_1mab_prev,
                 void* model_in, int copies, float inv_scaling, void* decode_in) {
    return lcm_step_t<half_t>(ctx, eps, ld_eps, nb, guidance, latents, noise, HW, sqrt_ab_t, sqrt_1mab_t, c_skip, c_out, sqrt_ab_prev,
                              sqrt_1mab_prev, model_in, copies, inv_scaling, decode_in);
}
int fie_lcm_step_f32(fie_ctx* ctx, const void* eps, int64_t ld_eps, int nb, float guidance, float* latents, const float* noise,
                     int64_t HW, float sqrt_ab_t, float sqrt_1mab_t, float c_skip, float c_out, float sqrt_ab_prev,
                     float sqrt_1mab_prev, void* model_in, int copies, float inv_scaling, void* decode_in) {
    return lcm_step_t<float>(ctx, eps, ld_eps, nb, guidance, latents, noise, HW, sqrt_ab_t, sqrt_1mab_t, c_skip, c_out, sqrt_ab_prev,
                             sqrt_1mab_prev, model_in, copies, inv_scaling, decode_in);
}

}  // extern "C"
