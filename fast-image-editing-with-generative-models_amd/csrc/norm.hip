// K5 GroupNorm(+SiLU) and K6 LayerNorm for gfx950 -- HBM-bound, 16-byte vector accesses, fp32 statistics.
// (include/fie.h: fie_groupnorm_nhwc_f16, fie_layernorm_f16)
//
// GroupNorm over NHWC [B, rows, C] with the channel dim optionally split over two source tensors (the UNet up
// blocks' concat is never materialised before its GroupNorm).  Three launches:
//   1. partial sums   grid (chunks, B, csplit): each thread owns ONE 8-channel column chunk and walks rows,
//                     block-reduces to per-group (sum, sumsq) partials -> workspace [B][chunks][G][2] (deterministic)
//   2. finalize       one thread per (b, g): reduce chunk partials -> (mean, rstd)
//   3. apply          same thread->column mapping; y = x * a_c + b_c with a, b folded from gamma/beta/mean/rstd, SiLU
#include "fie_internal.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_CHUNKS = 1024;

template <typename T>
struct GnArgsT {
    const T* X1; int C1;
    const T* X2; int C2;
    T* Y;
    int C, G, cg;
    int64_t rows;            // per image
    int rows_per_chunk, nchunks;
    int ncol;                // 8-channel column chunks handled per block (<= 256)
    int rpp;                 // rows per pass = 256 / ncol
    const T* gamma; const T* beta;
    float eps; int silu;
    float* partial;          // [B][nchunks][G][2]
    float* stats;            // [B][G][2] mean, rstd
    float o8_inv;            // > 0: Y is e4m3 bytes (value * o8_inv, saturated to +-448): the consumer is an fp8-activation conv (gemm_x8.hip)
    int pg;                  // producer-written partials only (gn_finalize_wide_kernel): slots per granule row; 0 = G.  pg > G: group g is the sum
                             // of the pg / G consecutive slots (4-channel quads) from g * pg / G on -- the 20 / 40-channel groups of the UNet
};

using GnArgs = GnArgsT<half_t>;

__device__ __forceinline__ int gn_pack4_f8(const float* o, float inv) {
    auto q = [&](float x) { return fie_sat448(x * inv); };
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q(o[0]), q(o[1]), 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(q(o[2]), q(o[3]), pk, true);
}

template <typename T>
__device__ __forceinline__ void gn_load(const GnArgsT<T>& p, int64_t row_global, int c0, float (&x)[8]) {
    if (c0 < p.C1) fie_load8(p.X1 + row_global * p.C1 + c0, x);
    else fie_load8(p.X2 + row_global * p.C2 + (c0 - p.C1), x);
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_partial_kernel(GnArgsT<T> p) {
    __shared__ float red[GN_THREADS * 16];
    const int tid = threadIdx.x;
    const int col = tid % p.ncol, rsub = tid / p.ncol;
    const int b = blockIdx.y;
    const int c0 = (blockIdx.z * p.ncol + col) * 8;
    const bool active = rsub < p.rpp && c0 < p.C;
    float s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
    if (active) {
        const int64_t r_begin = (int64_t)blockIdx.x * p.rows_per_chunk;
        const int64_t r_end = min(r_begin + p.rows_per_chunk, p.rows);
        for (int64_t r = r_begin + rsub; r < r_end; r += p.rpp) {
            float v[8];
            gn_load(p, (int64_t)b * p.rows + r, c0, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] += v[j];
                ss[j] += v[j] * v[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[tid * 16 + j] = s[j];
        red[tid * 16 + 8 + j] = ss[j];
    }
    __syncthreads();
    // groups covered by this block's channel range: [cbase, cbase + ncol*8)
    const int cbase = blockIdx.z * p.ncol * 8;
    const int cend = min(cbase + p.ncol * 8, p.C);
    const int g_first = cbase / p.cg, g_last = (cend - 1) / p.cg;
    for (int g = g_first + tid; g <= g_last; g += GN_THREADS) {
        float a = 0.f, q = 0.f;
        const int ch_lo = max(g * p.cg, cbase), ch_hi = min((g + 1) * p.cg, cend);
        for (int ch = ch_lo; ch < ch_hi; ++ch) {
            const int lc = (ch - cbase) >> 3, j = (ch - cbase) & 7;
            for (int rs = 0; rs < p.rpp; ++rs) {
                a += red[(rs * p.ncol + lc) * 16 + j];
                q += red[(rs * p.ncol + lc) * 16 + 8 + j];
            }
        }
        float* dst = p.partial + (((int64_t)b * p.nchunks + blockIdx.x) * p.G + g) * 2;
        // a group never straddles two column splits (checked on the host), so plain stores suffice
        dst[0] = a;
        dst[1] = q;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_finalize_kernel(GnArgsT<T> p, int B) {
    // one wave per (b, g): lanes stride over the chunk partials, xor-shuffle reduce in fp64
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B * p.G) return;
    const int b = i / p.G, g = i - b * p.G;
    double a = 0.0, q = 0.0;
    for (int c = lane; c < p.nchunks; c += 64) {
        const float2 v = *reinterpret_cast<const float2*>(p.partial + (((int64_t)b * p.nchunks + c) * p.G + g) * 2);
        a += (double)v.x;
        q += (double)v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        q += __shfl_xor(q, o);
    }
    if (lane == 0) {
        const double n = (double)p.rows * p.cg;
        const double mean = a / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        p.stats[i * 2] = (float)mean;
        p.stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
}

// The same reduction for the producer-written partials (one per 32-row granule: up to 32 768 per image at 1024^2): a whole
// 256-thread block per (b, g), four independent accumulator pairs per thread so that the loads stay in flight.
template <typename T>
__global__ __launch_bounds__(256) void gn_finalize_wide_kernel(GnArgsT<T> p, int B) {
    __shared__ double red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x;
    const int b = i / p.G, g = i - b * p.G;
    double a0 = 0.0, q0 = 0.0, a1 = 0.0, q1 = 0.0, a2 = 0.0, q2 = 0.0, a3 = 0.0, q3 = 0.0;
    if (p.pg > p.G) {
        // quad partials: thread = (granule, quad of this group); fixed assignment and order -> deterministic
        const int upg = p.pg / p.G;
        const float2* src = reinterpret_cast<const float2*>(p.partial) + (int64_t)b * p.nchunks * p.pg + g * upg;
        const int total = p.nchunks * upg;
        for (int i = tid; i < total; i += 256) {
            const int c = i / upg, u = i - c * upg;
            const float2 v = src[(int64_t)c * p.pg + u];
            a0 += (double)v.x; q0 += (double)v.y;
        }
    } else {
    const float2* src = reinterpret_cast<const float2*>(p.partial) + (int64_t)b * p.nchunks * p.G + g;
    int c = tid;
    for (; c + 768 < p.nchunks; c += 1024) {
        const float2 v0 = src[(int64_t)c * p.G], v1 = src[(int64_t)(c + 256) * p.G], v2 = src[(int64_t)(c + 512) * p.G], v3 = src[(int64_t)(c + 768) * p.G];
        a0 += (double)v0.x; q0 += (double)v0.y; a1 += (double)v1.x; q1 += (double)v1.y;
        a2 += (double)v2.x; q2 += (double)v2.y; a3 += (double)v3.x; q3 += (double)v3.y;
    }
    for (; c < p.nchunks; c += 256) {
        const float2 v = src[(int64_t)c * p.G];
        a0 += (double)v.x; q0 += (double)v.y;
    }
    }
    double a = (a0 + a1) + (a2 + a3), q = (q0 + q1) + (q2 + q3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        q += __shfl_xor(q, o);
    }
    if (lane == 0) { red[wave * 2] = a; red[wave * 2 + 1] = q; }
    __syncthreads();
    if (tid == 0) {
        a = (red[0] + red[2]) + (red[4] + red[6]);
        q = (red[1] + red[3]) + (red[5] + red[7]);
        const double n = (double)p.rows * p.cg;
        const double mean = a / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        p.stats[i * 2] = (float)mean;
        p.stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
}

// Large maps (512^2 / 1024^2 VAE tensors: 8 192 - 32 768 granules per image): the walk above reads one float2 per 256-byte row of the
// producer's [granule][group] table from only B * G blocks (43 us at 1024^2).  First gather the granules with coalesced reads: block (i, b)
// sums `per` consecutive granules for ALL groups (thread = (row lane, group): a wave instruction reads two whole table rows), fp64 in
// registers, and writes one (sum, sum of squares) row in the layout of the ordinary chunk partials; gn_finalize_kernel then reduces the few
// hundred rows.  Fixed assignment and order: deterministic.
template <typename T>
__global__ __launch_bounds__(256) void gn_gather_granules_kernel(GnArgsT<T> p, const float* granules, int ngran, int per) {
    __shared__ double red[256][2];
    const int tid = threadIdx.x, g = tid % p.G, sub = tid / p.G, nsub = 256 / p.G;
    const int b = blockIdx.y, c0 = blockIdx.x * per, c1 = min(c0 + per, ngran);
    const float2* src = reinterpret_cast<const float2*>(granules) + (int64_t)b * ngran * p.G + g;
    double a0 = 0.0, q0 = 0.0, a1 = 0.0, q1 = 0.0;
    int c = c0 + sub;
    for (; c + nsub < c1; c += 2 * nsub) {
        const float2 v0 = src[(int64_t)c * p.G], v1 = src[(int64_t)(c + nsub) * p.G];
        a0 += (double)v0.x; q0 += (double)v0.y; a1 += (double)v1.x; q1 += (double)v1.y;
    }
    if (c < c1) { const float2 v = src[(int64_t)c * p.G]; a0 += (double)v.x; q0 += (double)v.y; }
    red[tid][0] = a0 + a1; red[tid][1] = q0 + q1;
    __syncthreads();
    if (sub == 0) {
        double a = 0.0, q = 0.0;
        for (int s = 0; s < nsub; ++s) { a += red[s * p.G + g][0]; q += red[s * p.G + g][1]; }
        *reinterpret_cast<float2*>(p.partial + (((int64_t)b * gridDim.x + blockIdx.x) * p.G + g) * 2) = make_float2((float)a, (float)q);
    }
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(GnArgsT<T> p) {
    const int tid = threadIdx.x;
    const int col = tid % p.ncol, rsub = tid / p.ncol;
    const int b = blockIdx.y;
    const int c0 = (blockIdx.z * p.ncol + col) * 8;
    if (!(rsub < p.rpp && c0 < p.C)) return;
    float sc[8], sh[8];
    float gm[8], bt[8];
    fie_load8(p.gamma + c0, gm);
    fie_load8(p.beta + c0, bt);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (c0 + j) / p.cg;
        const float mean = p.stats[(b * p.G + g) * 2], rstd = p.stats[(b * p.G + g) * 2 + 1];
        sc[j] = rstd * gm[j];
        sh[j] = bt[j] - mean * sc[j];
    }
    const int64_t r_begin = (int64_t)blockIdx.x * p.rows_per_chunk;
    const int64_t r_end = min(r_begin + p.rows_per_chunk, p.rows);
    auto emit = [&](int64_t rg, const float (&v)[8]) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float y = v[j] * sc[j] + sh[j];
            if (p.silu) y = sizeof(T) == 4 ? y / (1.0f + expf(-y)) : fie_silu(y);
            o[j] = y;
        }
        if (p.o8_inv > 0.f)
            *reinterpret_cast<int2*>(reinterpret_cast<unsigned char*>(p.Y) + rg * p.C + c0) = make_int2(gn_pack4_f8(o, p.o8_inv), gn_pack4_f8(o + 4, p.o8_inv));
        else
            fie_store8(p.Y + rg * p.C + c0, o);
    };
    // four rows in flight per thread: with one, a fully occupied chip holds ~8 MB of loads = ~4 TB/s at HBM latency (measured 4.4-5.3 on the VAE maps)
    constexpr int U = 4;
    int64_t r = r_begin + rsub;
    for (; r + (int64_t)(U - 1) * p.rpp < r_end; r += (int64_t)U * p.rpp) {
        float v[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) gn_load(p, (int64_t)b * p.rows + r + (int64_t)u * p.rpp, c0, v[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) emit((int64_t)b * p.rows + r + (int64_t)u * p.rpp, v[u]);
    }
    for (; r < r_end; r += p.rpp) {
        float v[8];
        gn_load(p, (int64_t)b * p.rows + r, c0, v);
        emit((int64_t)b * p.rows + r, v);
    }
}

// ---- single-pass GroupNorm for the small feature maps (32x32 / 64x64 latents): ONE block of 1024 threads owns one (image, group),
// keeps the group's rows x cg values in registers (<= NV vectors of V channels per thread), reduces mean and then the CENTRED
// second moment across the block, and writes the normalised (+SiLU) values -- one launch and one read of the tensor instead of
// partial / finalize / apply (three launches, two reads; ~25 us -> ~8 us on a 2 x 1024 x 1280 map, which is pure launch latency).
// A thread keeps one fixed channel piece (the block uses (1024 / npv) * npv threads), so gamma / beta live in registers.
constexpr int GN1_THREADS = 1024;

template <int ND>
struct alignas(ND * 4) GnRaw { uint32_t d[ND]; };       // one vector of V channels as raw dwords (ND = V * sizeof(T) / 4)

template <int V, typename T>
__device__ __forceinline__ void gn_unpack(const uint32_t* d, float (&f)[V]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < V; ++j) f[j] = __builtin_bit_cast(float, d[j]);
    } else {
#pragma unroll
        for (int j = 0; j < V / 2; ++j) {
            const f16x2 h = __builtin_bit_cast(f16x2, d[j]);
            f[2 * j] = (float)h[0];
            f[2 * j + 1] = (float)h[1];
        }
    }
}

template <int V, int NV, typename T>
__global__ __launch_bounds__(GN1_THREADS) void gn_onepass_kernel(GnArgsT<T> p) {
    constexpr int ND = V * (int)sizeof(T) / 4;
    __shared__ float red[2][GN1_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, g = blockIdx.x;
    const int npv = p.cg / V;                           // vectors per row of this group
    const int rstep = GN1_THREADS / npv;                // rows covered per pass
    const bool live = tid < rstep * npv;
    const int piece = tid % npv, rsub = tid / npv;
    const int c0 = g * p.cg + piece * V;                // first channel of this thread's vector
    const T* src; int ldx;
    if (c0 < p.C1) { src = p.X1 + c0; ldx = p.C1; }
    else { src = p.X2 + (c0 - p.C1); ldx = p.C2; }
    src += ((int64_t)b * p.rows + rsub) * ldx;
    const int64_t sstep = (int64_t)rstep * ldx;
    GnRaw<ND> x[NV];                                    // kept in the storage type (80 f16 values = 40 VGPRs) ...
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (live && i * rstep + rsub < p.rows) {
            x[i] = *reinterpret_cast<const GnRaw<ND>*>(src + i * sstep);
            float f[V];
            gn_unpack<V, T>(x[i].d, f);
#pragma unroll
            for (int j = 0; j < V; ++j) sum += f[j];
        }
    }
    auto block_sum = [&](float v, int slot) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[slot][wave] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < GN1_THREADS / 64; ++w) t += red[slot][w];
        return t;
    };
    const float n = (float)p.rows * (float)p.cg;
    const float mean = block_sum(sum, 0) / n;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (live && i * rstep + rsub < p.rows) {
            // ... and re-converted in each phase: the empty asm stops the compiler from keeping the fp32 copies alive
#pragma unroll
            for (int k = 0; k < ND; ++k) asm volatile("" : "+v"(x[i].d[k]));
            float f[V];
            gn_unpack<V, T>(x[i].d, f);
#pragma unroll
            for (int j = 0; j < V; ++j) { const float d = f[j] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(block_sum(sq, 1) / n + p.eps);
    if (!live) return;
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        sc[j] = rstd * (float)p.gamma[c0 + j];
        sh[j] = (float)p.beta[c0 + j] - mean * sc[j];
    }
    T* dst = p.Y + ((int64_t)b * p.rows + rsub) * p.C + c0;
    const int64_t dstep = (int64_t)rstep * p.C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (i * rstep + rsub < p.rows) {
#pragma unroll
            for (int k = 0; k < ND; ++k) asm volatile("" : "+v"(x[i].d[k]));
            float f[V];
            gn_unpack<V, T>(x[i].d, f);
            GnRaw<ND> o;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float y = f[j] * sc[j] + sh[j];
                if (p.silu) y = sizeof(T) == 4 ? y / (1.0f + expf(-y)) : fie_silu(y);
                f[j] = y;
            }
            if (p.o8_inv > 0.f) {                       // e4m3 bytes: V values = V bytes at the same (row, channel) position of a byte tensor
                unsigned char* d8 = reinterpret_cast<unsigned char*>(p.Y) + ((int64_t)b * p.rows + rsub + (int64_t)i * rstep) * p.C + c0;
                if constexpr (V == 8) *reinterpret_cast<int2*>(d8) = make_int2(gn_pack4_f8(f, p.o8_inv), gn_pack4_f8(f + 4, p.o8_inv));
                else *reinterpret_cast<int*>(d8) = gn_pack4_f8(f, p.o8_inv);
                continue;
            }
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int j = 0; j < V; ++j) o.d[j] = __builtin_bit_cast(uint32_t, f[j]);
            } else {
#pragma unroll
                for (int j = 0; j < V / 2; ++j) {
                    f16x2 h;
                    h[0] = (half_t)f[2 * j];
                    h[1] = (half_t)f[2 * j + 1];
                    o.d[j] = __builtin_bit_cast(uint32_t, h);
                }
            }
            *reinterpret_cast<GnRaw<ND>*>(dst + i * dstep) = o;
        }
    }
}

// (V, NV) of the single-pass kernel for this shape, or false: the group's rows x cg values must fit NV vectors per thread
template <typename T>
bool gn_onepass(fie_ctx* ctx, const GnArgsT<T>& p, int B) {
    if (p.C2 && p.C1 % p.cg != 0) return false;                          // a group never straddles the two sources
    const int V = p.cg % 8 == 0 ? 8 : (p.cg % 4 == 0 ? 4 : 0);
    if (!V || p.C1 % V || p.C2 % V || p.cg / V > GN1_THREADS) return false;
    const int npv = p.cg / V, rstep = GN1_THREADS / npv;
    const int64_t nv = (p.rows + rstep - 1) / rstep;
    const dim3 grid((unsigned)p.G, (unsigned)B);
    FIE_DESC(ctx, "groupnorm onepass B=%d rows=%lld C=%d G=%d bytes=%.0f", B, (long long)p.rows, p.C, p.G, 2.0 * B * p.rows * p.C * sizeof(T));
    if (V == 8 && nv <= 5) fie_launch(ctx, (gn_onepass_kernel<8, 5, T>), grid, dim3(GN1_THREADS), 0, p);
    else if (V == 8 && nv <= 10) fie_launch(ctx, (gn_onepass_kernel<8, 10, T>), grid, dim3(GN1_THREADS), 0, p);
    // V = 4 at rows = 4096 (64x64 latents, 21 vectors) measured SLOWER than the three-kernel path (28.5 vs 22.0 us: 40-byte
    // slices at a 1280-byte stride), so the bound stays at 20
    else if (V == 4 && nv <= 20) fie_launch(ctx, (gn_onepass_kernel<4, 20, T>), grid, dim3(GN1_THREADS), 0, p);
    else return false;
    return true;
}

// ---- LayerNorm: one wave per row, row kept in registers (C <= 4096)
constexpr int LN_MAXV = 8;   // 8 chunks x 8 values per lane

// O8: the output is written as e4m3 bytes, value * inv8 saturated to +-448 (fie_layernorm_f16_o8: the consumer is an fp8-activation GEMM)
// NV = 16-byte chunks per lane (2: C <= 1024, 3: C <= 1536, 8: the rest).  With NV <= 3 gamma / beta are loaded WITH the row, ahead of the two
// reductions, instead of after them: the kernel is one latency chain (5-6 us for 5 MB), and the second round trip was a fifth of it.
template <typename T, bool O8 = false, int NV = LN_MAXV>
__global__ __launch_bounds__(256) void ln_kernel(const T* X, int64_t ldx, T* Y, int64_t ldy, int64_t rows,
                                                 int C, const T* gamma, const T* beta, float eps, float inv8 = 1.f) {
    constexpr bool HOIST = NV <= 3;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = C >> 3;
    float v[NV][8], g[HOIST ? NV : 1][8], bb[HOIST ? NV : 1][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            fie_load8(X + row * ldx + ch * 8, v[i]);
            if constexpr (HOIST) {
                fie_load8(gamma + ch * 8, g[i]);
                fie_load8(beta + ch * 8, bb[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[i][j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[i][j] - mean;
                var += d * d;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = rsqrtf(var / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            float gl[8], bl[8], o[8];
            if constexpr (!HOIST) {
                fie_load8(gamma + ch * 8, gl);
                fie_load8(beta + ch * 8, bl);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * (HOIST ? g[HOIST ? i : 0][j] : gl[j]) + (HOIST ? bb[HOIST ? i : 0][j] : bl[j]);
            if constexpr (O8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fie_sat448(o[j] * inv8);
                int lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], 0, false);
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], lo, true);
                int hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[4], o[5], 0, false);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[6], o[7], hi, true);
                *reinterpret_cast<int2*>(reinterpret_cast<unsigned char*>(Y) + row * ldy + ch * 8) = make_int2(lo, hi);
            } else {
                fie_store8(Y + row * ldy + ch * 8, o);
            }
        }
    }
}

template <typename T, bool O8, typename... Args>
void ln_launch(fie_ctx* ctx, int64_t rows, int C, Args... args) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (C <= 1024) fie_launch(ctx, (ln_kernel<T, O8, 2>), grid, block, 0, args...);
    else if (C <= 1536) fie_launch(ctx, (ln_kernel<T, O8, 3>), grid, block, 0, args...);
    else fie_launch(ctx, (ln_kernel<T, O8, LN_MAXV>), grid, block, 0, args...);
}

template <typename T>
int gn_plan(GnArgsT<T>& p, int C, int G, int64_t rows, int B, int* csplit) {
    const int nc8 = C / 8;
    int split = (nc8 + GN_THREADS - 1) / GN_THREADS;
    while (split <= nc8 && (nc8 % split != 0 || ((nc8 / split) * 8) % (C / G) != 0)) ++split;
    if (split > nc8) return -1;
    p.ncol = nc8 / split;
    p.rpp = GN_THREADS / p.ncol;
    *csplit = split;
    // aim at ~2048 blocks per launch: small tensors (the 32x32 / 64x64 latent levels) are latency-bound otherwise --
    // a thread should walk only a handful of rows
    int64_t want = 2048 / (B > 0 ? B : 1);
    if (want < 1) want = 1;
    if (want > GN_MAX_CHUNKS) want = GN_MAX_CHUNKS;
    int64_t rpc = fie_roundup((rows + want - 1) / want, p.rpp);
    int64_t nch = (rows + rpc - 1) / rpc;
    p.rows_per_chunk = (int)rpc;
    p.nchunks = (int)nch;
    return 0;
}

template <typename T>
int groupnorm_t(const char* who, fie_ctx* ctx, const void* X1, int C1, const void* X2, int C2, void* Y, int B, int64_t rows_per_image,
                int groups, const void* gamma, const void* beta, float eps, int silu, void* workspace, float o8_inv = 0.f) {
    FIE_REQUIRE(ctx && X1 && Y && gamma && beta && workspace, "%s: NULL argument", who);
    FIE_REQUIRE(C1 > 0 && C1 % 8 == 0 && C2 >= 0 && C2 % 8 == 0 && (C2 == 0 || X2), "%s: C1=%d C2=%d invalid", who, C1, C2);
    const int C = C1 + C2;
    FIE_REQUIRE(B > 0 && rows_per_image > 0 && groups > 0 && C % groups == 0, "%s: bad shape C=%d G=%d", who, C, groups);
    GnArgsT<T> p = {};
    p.X1 = (const T*)X1; p.C1 = C1; p.X2 = (const T*)X2; p.C2 = C2; p.Y = (T*)Y;
    p.C = C; p.G = groups; p.cg = C / groups; p.rows = rows_per_image;
    p.gamma = (const T*)gamma; p.beta = (const T*)beta; p.eps = eps; p.silu = silu;
    p.o8_inv = o8_inv;
    int csplit = 1;
    FIE_REQUIRE(gn_plan(p, C, groups, rows_per_image, B, &csplit) == 0, "%s: cannot split C=%d (groups=%d) into aligned column blocks", who, C, groups);
    if (ctx->gn_onepass && gn_onepass(ctx, p, B)) {
        FIE_LAUNCH_CHECK();
        return FIE_OK;
    }
    p.partial = (float*)workspace;
    p.stats = p.partial + (int64_t)B * GN_MAX_CHUNKS * groups * 2;
    const dim3 grid((unsigned)p.nchunks, (unsigned)B, (unsigned)csplit);
    FIE_DESC(ctx, "groupnorm partial B=%d rows=%lld C=%d G=%d bytes=%.0f", B, (long long)rows_per_image, C, groups, 1.0 * B * rows_per_image * C * sizeof(T));
    fie_launch(ctx, gn_partial_kernel<T>, grid, dim3(GN_THREADS), 0, p);
    FIE_DESC(ctx, "groupnorm finalize B=%d rows=%lld C=%d G=%d bytes=0", B, (long long)rows_per_image, C, groups);
    fie_launch(ctx, gn_finalize_kernel<T>, dim3((B * groups + 3) / 4), dim3(256), 0, p, B);
    FIE_DESC(ctx, "groupnorm apply B=%d rows=%lld C=%d G=%d bytes=%.0f", B, (long long)rows_per_image, C, groups, 2.0 * B * rows_per_image * C * sizeof(T));
    fie_launch(ctx, gn_apply_kernel<T>, grid, dim3(GN_THREADS), 0, p);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

template <typename T>
int layernorm_t(const char* who, fie_ctx* ctx, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t rows, int C, const void* gamma,
                const void* beta, float eps) {
    FIE_REQUIRE(ctx && X && Y && gamma && beta, "%s: NULL argument", who);
    FIE_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAXV, "%s: C=%d unsupported", who, C);
    FIE_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, "%s: bad strides", who);
    FIE_DESC(ctx, "layernorm rows=%lld C=%d bytes=%.0f", (long long)rows, C, 2.0 * rows * C * sizeof(T));
    ln_launch<T, false>(ctx, rows, C, (const T*)X, ldx, (T*)Y, ldy, rows, C, (const T*)gamma, (const T*)beta, eps, 1.f);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

}  // namespace

// GroupNorm whose first pass was done by the producer: `partial` holds (sum, sum of squares) per image, 32-row granule and group, written
// by the epilogue of the GEMM / conv that produced X (fie_gn_stats_target); finalize over the granules, then apply: one read of the tensor
// instead of two.
template <typename T>
int groupnorm_stats_t(const char* who, fie_ctx* ctx, const void* X, int C, void* Y, int B, int64_t rows_per_image, int groups, const void* gamma,
                      const void* beta, float eps, int silu, const void* partial, void* workspace, int partial_groups, float o8_inv = 0.f) {
    FIE_REQUIRE(ctx && X && Y && gamma && beta && partial && workspace, "%s: NULL argument", who);
    FIE_REQUIRE(C > 0 && C % 8 == 0 && B > 0 && groups > 0 && C % groups == 0 && rows_per_image > 0 && rows_per_image % 32 == 0,
                "%s: bad shape C=%d G=%d rows=%lld", who, C, groups, (long long)rows_per_image);
    GnArgsT<T> p = {};
    p.X1 = (const T*)X; p.C1 = C; p.X2 = nullptr; p.C2 = 0; p.Y = (T*)Y;
    p.C = C; p.G = groups; p.cg = C / groups; p.rows = rows_per_image;
    p.gamma = (const T*)gamma; p.beta = (const T*)beta; p.eps = eps; p.silu = silu;
    p.o8_inv = o8_inv;
    int csplit = 1;
    FIE_REQUIRE(gn_plan(p, C, groups, rows_per_image, B, &csplit) == 0, "%s: cannot split C=%d (groups=%d) into aligned column blocks", who, C, groups);
    const dim3 grid((unsigned)p.nchunks, (unsigned)B, (unsigned)csplit);
    p.stats = (float*)workspace + (int64_t)B * GN_MAX_CHUNKS * groups * 2;
    GnArgsT<T> f = p;                                       // finalize walks the producer's granules, apply its own row chunks
    f.partial = const_cast<float*>((const float*)partial);
    f.nchunks = (int)(rows_per_image / 32);
    FIE_REQUIRE(partial_groups == 0 || partial_groups == groups || (partial_groups > groups && partial_groups % groups == 0 && partial_groups * 4 == C),
                "%s: partial_groups=%d must be 0, the group count, or C / 4 quads divisible into the %d groups", who, partial_groups, groups);
    f.pg = partial_groups;
    const int ngran = (int)(rows_per_image / 32);
    if (ngran >= 4096 && groups <= 256 && 256 % groups == 0 && partial_groups <= groups) {        // big maps: coalesced gather into <= 256 rows, then the ordinary finalize
        const int nb = 256, per = (ngran + nb - 1) / nb;
        GnArgsT<T> g2 = p;
        g2.partial = (float*)workspace;                     // the chunk-partial area of the workspace is free on this path: [B][nb][G][2]
        g2.nchunks = nb;
        FIE_DESC(ctx, "groupnorm gather-granules B=%d rows=%lld C=%d G=%d bytes=%.0f", B, (long long)rows_per_image, C, groups, 8.0 * B * ngran * groups);
        fie_launch(ctx, gn_gather_granules_kernel<T>, dim3(nb, B), dim3(256), 0, g2, (const float*)partial, ngran, per);
        FIE_DESC(ctx, "groupnorm finalize B=%d rows=%lld C=%d G=%d bytes=0", B, (long long)rows_per_image, C, groups);
        fie_launch(ctx, gn_finalize_kernel<T>, dim3((B * groups + 3) / 4), dim3(256), 0, g2, B);
    } else {
        FIE_DESC(ctx, "groupnorm finalize-from-epilogue B=%d rows=%lld C=%d G=%d bytes=0", B, (long long)rows_per_image, C, groups);
        fie_launch(ctx, gn_finalize_wide_kernel<T>, dim3(B * groups), dim3(256), 0, f, B);
    }
    FIE_DESC(ctx, "groupnorm apply B=%d rows=%lld C=%d G=%d bytes=%.0f", B, (long long)rows_per_image, C, groups, 2.0 * B * rows_per_image * C * sizeof(T));
    fie_launch(ctx, gn_apply_kernel<T>, grid, dim3(GN_THREADS), 0, p);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

// GroupNorm as per-(image, channel) coefficients instead of a normalised tensor: coef[b][c] = (sc, sh) with sc = rstd * gamma, sh = beta - mean * sc, the
// very fp32 values gn_apply_kernel forms per thread -- for a consumer that applies y = x * sc + sh itself (conv_halo.hip, GNA)
namespace {
template <typename T>
__global__ __launch_bounds__(256) void gn_coef_kernel(GnArgsT<T> p, float* coef) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    const int g = c / p.cg;
    const float mean = p.stats[(b * p.G + g) * 2], rstd = p.stats[(b * p.G + g) * 2 + 1];
    const float sc = rstd * (float)p.gamma[c];
    const float sh = (float)p.beta[c] - mean * sc;
    reinterpret_cast<float2*>(coef)[(int64_t)b * p.C + c] = make_float2(sc, sh);
}
}  // namespace

extern "C" {

int fie_groupnorm_coef_f16(fie_ctx* ctx, int C, int B, int64_t rows_per_image, int groups, const void* gamma, const void* beta, float eps, const void* partial,
                           void* workspace, int partial_groups, float* coef) {
    const char* who = "fie_groupnorm_coef_f16";
    FIE_REQUIRE(ctx && gamma && beta && partial && workspace && coef, "%s: NULL argument", who);
    FIE_REQUIRE(C > 0 && C % 8 == 0 && B > 0 && groups > 0 && C % groups == 0 && rows_per_image > 0 && rows_per_image % 32 == 0,
                "%s: bad shape C=%d G=%d rows=%lld", who, C, groups, (long long)rows_per_image);
    FIE_REQUIRE(partial_groups == 0 || partial_groups == groups || (partial_groups > groups && partial_groups % groups == 0 && partial_groups * 4 == C),
                "%s: partial_groups=%d must be 0, the group count, or C / 4 quads divisible into the %d groups", who, partial_groups, groups);
    GnArgs p = {};
    p.C1 = C; p.C = C; p.G = groups; p.cg = C / groups; p.rows = rows_per_image;
    p.gamma = (const half_t*)gamma; p.beta = (const half_t*)beta; p.eps = eps;
    p.stats = (float*)workspace + (int64_t)B * GN_MAX_CHUNKS * groups * 2;
    const int ngran = (int)(rows_per_image / 32);
    if (ngran >= 4096 && groups <= 256 && 256 % groups == 0 && partial_groups <= groups) {        // as groupnorm_stats_t: gather, then the ordinary finalize
        const int nb = 256, per = (ngran + nb - 1) / nb;
        GnArgs g2 = p;
        g2.partial = (float*)workspace;
        g2.nchunks = nb;
        FIE_DESC(ctx, "groupnorm gather-granules B=%d rows=%lld C=%d G=%d bytes=%.0f", B, (long long)rows_per_image, C, groups, 8.0 * B * ngran * groups);
        fie_launch(ctx, gn_gather_granules_kernel<half_t>, dim3(nb, B), dim3(256), 0, g2, (const float*)partial, ngran, per);
        FIE_DESC(ctx, "groupnorm finalize B=%d rows=%lld C=%d G=%d bytes=0", B, (long long)rows_per_image, C, groups);
        fie_launch(ctx, gn_finalize_kernel<half_t>, dim3((B * groups + 3) / 4), dim3(256), 0, g2, B);
    } else {
        GnArgs f = p;
        f.partial = const_cast<float*>((const float*)partial);
        f.nchunks = ngran;
        f.pg = partial_groups;
        FIE_DESC(ctx, "groupnorm finalize-from-epilogue B=%d rows=%lld C=%d G=%d bytes=0", B, (long long)rows_per_image, C, groups);
        fie_launch(ctx, gn_finalize_wide_kernel<half_t>, dim3(B * groups), dim3(256), 0, f, B);
    }
    FIE_DESC(ctx, "groupnorm coefficients B=%d rows=%lld C=%d G=%d bytes=0", B, (long long)rows_per_image, C, groups);
    fie_launch(ctx, gn_coef_kernel<half_t>, dim3((C + 255) / 256, B), dim3(256), 0, p, coef);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

int64_t fie_gn_stats_bytes(int B, int64_t rows_per_image, int groups) { return (int64_t)B * (rows_per_image / 32) * groups * 2 * (int64_t)sizeof(float); }

int fie_groupnorm_stats_nhwc_f16(fie_ctx* ctx, const void* X, int C, void* Y, int B, int64_t rows_per_image, int groups, const void* gamma,
                                 const void* beta, float eps, int silu, const void* partial, void* workspace, int partial_groups) {
    return groupnorm_stats_t<half_t>("fie_groupnorm_stats_nhwc_f16", ctx, X, C, Y, B, rows_per_image, groups, gamma, beta, eps, silu, partial, workspace,
                                     partial_groups);
}

// GroupNorm (+SiLU) with an e4m3 OUTPUT (BASELINE config 5: the resnet convs read fp8 activations, csrc/gemm_x8.hip): Y8 [B, rows, C] bytes,
// value * inv_scale saturated to the e4m3 range; otherwise as their f16 twins.
int fie_groupnorm_nhwc_f16_o8(fie_ctx* ctx, const void* X1, int C1, const void* X2, int C2, void* Y8, int B, int64_t rows_per_image, int groups,
                              const void* gamma, const void* beta, float eps, int silu, void* workspace, float inv_scale) {
    FIE_REQUIRE(inv_scale > 0.f, "fie_groupnorm_nhwc_f16_o8: inv_scale must be positive");
    return groupnorm_t<half_t>("fie_groupnorm_nhwc_f16_o8", ctx, X1, C1, X2, C2, Y8, B, rows_per_image, groups, gamma, beta, eps, silu, workspace, inv_scale);
}

int fie_groupnorm_stats_nhwc_f16_o8(fie_ctx* ctx, const void* X, int C, void* Y8, int B, int64_t rows_per_image, int groups, const void* gamma,
                                    const void* beta, float eps, int silu, const void* partial, void* workspace, int partial_groups, float inv_scale) {
    FIE_REQUIRE(inv_scale > 0.f, "fie_groupnorm_stats_nhwc_f16_o8: inv_scale must be positive");
    return groupnorm_stats_t<half_t>("fie_groupnorm_stats_nhwc_f16_o8", ctx, X, C, Y8, B, rows_per_image, groups, gamma, beta, eps, silu, partial, workspace,
                                     partial_groups, inv_scale);
}

int64_t fie_groupnorm_workspace_bytes(int B, int64_t rows_per_image, int groups) {
    (void)rows_per_image;
    return ((int64_t)B * GN_MAX_CHUNKS * groups * 2 + (int64_t)B * groups * 2) * (int64_t)sizeof(float);
}

int fie_groupnorm_nhwc_f16(fie_ctx* ctx, const void* X1, int C1, const void* X2, int C2, void* Y, int B,
                           int64_t rows_per_image, int groups, const void* gamma, const void* beta, float eps,
                           int silu, void* workspace) {
    return groupnorm_t<half_t>("fie_groupnorm_nhwc_f16", ctx, X1, C1, X2, C2, Y, B, rows_per_image, groups, gamma, beta, eps, silu, workspace);
}

int fie_groupnorm_nhwc_f32(fie_ctx* ctx, const void* X1, int C1, const void* X2, int C2, void* Y, int B,
                           int64_t rows_per_image, int groups, const void* gamma, const void* beta, float eps,
                           int silu, void* workspace) {
    return groupnorm_t<float>("fie_groupnorm_nhwc_f32", ctx, X1, C1, X2, C2, Y, B, rows_per_image, groups, gamma, beta, eps, silu, workspace);
}

int fie_layernorm_f16(fie_ctx* ctx, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t rows, int C,
                      const void* gamma, const void* beta, float eps) {
    return layernorm_t<half_t>("fie_layernorm_f16", ctx, X, ldx, Y, ldy, rows, C, gamma, beta, eps);
}

int fie_layernorm_f32(fie_ctx* ctx, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t rows, int C,
                      const void* gamma, const void* beta, float eps) {
    return layernorm_t<float>("fie_layernorm_f32", ctx, X, ldx, Y, ldy, rows, C, gamma, beta, eps);
}

// LayerNorm with an e4m3 output (BASELINE config 5: the q/k/v, cross-attention query and FF1 projections read fp8 activations): Y8 [rows, C]
// bytes, row stride ldy8 BYTES, value * inv_scale saturated to the e4m3 range.
int fie_layernorm_f16_o8(fie_ctx* ctx, const void* X, int64_t ldx, void* Y8, int64_t ldy8, int64_t rows, int C, const void* gamma, const void* beta,
                         float eps, float inv_scale) {
    const char* who = "fie_layernorm_f16_o8";
    FIE_REQUIRE(ctx && X && Y8 && gamma && beta, "%s: NULL argument", who);
    FIE_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * 8 * LN_MAXV && inv_scale > 0.f, "%s: C=%d unsupported", who, C);
    FIE_REQUIRE(ldx % 8 == 0 && ldy8 % 8 == 0 && ldx >= C && ldy8 >= C, "%s: bad strides", who);
    FIE_DESC(ctx, "layernorm->e4m3 rows=%lld C=%d bytes=%.0f", (long long)rows, C, 3.0 * rows * C);
    ln_launch<half_t, true>(ctx, rows, C, (const half_t*)X, ldx, (half_t*)Y8, ldy8, rows, C, (const half_t*)gamma, (const half_t*)beta, eps, inv_scale);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

}  // extern "C"

extern "C" int fie_debug_gn_onepass(fie_ctx* ctx, int enable) {
    FIE_REQUIRE(ctx != nullptr, "fie_debug_gn_onepass: NULL ctx");
    ctx->gn_onepass = enable;
    return FIE_OK;
}
