// ARCHIVE -- NOT BUILT, not part of libfie_hip.so.  The round-1 GEMM / conv translation unit as it stood at the end of round 1,
// kept for its unselected experiments (v2 zero-page global_load_lds ring `gemm2_kernel`, ping-pong `gemm4_kernel`, four-phase
// `gemm5_kernel`, producer waves `gemm6_kernel`, halo-reuse `conv_halo_kernel`, 160-wide tiles, the code-70 column split) and
// their measured verdicts (profiles/r01_microbench.md).  The production kernels live in
// fast-image-editing-with-generative-models_amd/csrc/{gemm_common.h, gemm_conv.hip, gemm8.hip}.
// K1/K2: fp16 MFMA GEMM and 3x3 implicit-GEMM convolution for gfx950 (include/fie.h: fie_gemm_f16,
// fie_conv3x3_nhwc_f16).
//
// One kernel template serves both: C[M,N] = epi(A[M,K] * W[N,K]^T) where the "A" operand is either a row-major
// matrix (optionally the column concatenation of two matrices) or the on-the-fly im2col view of an NHWC tensor
// (3x3 taps, stride 1/2, symmetric or VAE-style asymmetric padding, optional fused nearest-2x upsample).
//
// Tiling (CDNA4): 256 threads = 4 waves in a 2x2 grid, block tile BM x BN x 64, v_mfma_f32_16x16x32_f16.
// The MFMA is issued "swapped": the weight fragment is the A operand and the activation fragment the B operand,
// so each lane ends up with 4 CONSECUTIVE output channels of one output row -> 8-byte bias/residual loads and
// 8-byte stores, and GEGLU value/gate pairs sit in one lane.
// LDS: two buffers of (BM + BN) rows x 128 B, 16-byte chunks XOR-swizzled by (row & 7) so that the
// ds_read_b128 fragment reads of 16 different rows spread over the banks.  Global->LDS goes through registers
// (the conv gather needs per-lane zero fill), issued one K-step ahead of the MFMAs that consume it.
#include <cstdio>
#include <cstdlib>
#include "fie_internal.h"

namespace {

constexpr int BK = 64;

struct GemmArgs {
    const half_t* A1; int64_t lda1; int K1;
    const half_t* A2; int64_t lda2;
    // conv view of A1
    int H, W, Cin, OH, OW, stride, pt, pl, ups;
    const half_t* Wt; int64_t ldw;
    half_t* C; int64_t ldc;
    int M, N, K;
    const half_t* bias;
    const half_t* rowbias; int64_t ld_rowbias; int rows_per_batch;
    const half_t* res; int64_t ldr;
    float scale; int act;
    int nbm, nbn;
    int64_t a1_bytes, a2_bytes, w_bytes;   // operand extents for the v3 buffer descriptors
    int order;                              // 0: n-tiles fastest (an XCD owns a range of rows), 1: m-tiles fastest (an XCD owns a range of columns)
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * BK + ((chunk ^ (row & 7)) << 3); }

template <int FM, int FN, int WM, int WN, bool PATCH = false>
__device__ __forceinline__ void epilogue(const GemmArgs& p, f32x4 (&acc)[FN][FM], int m0, int n0, int wm, int wn, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    // ---- epilogue: lane holds C[m = .. + fr][n = .. + fq*4 + (0..3)]
    // PATCH: the block's 256 rows are a 16x16 spatial patch whose top-left output pixel has row index m0
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = PATCH ? m0 + (wm * (WM / 16) + j) * p.OW + fr : m0 + wm * WM + j * 16 + fr;
        if (m >= p.M) continue;
        const half_t* rb = nullptr;
        if (p.rowbias) rb = p.rowbias + (int64_t)(m / p.rows_per_batch) * p.ld_rowbias;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int n = n0 + wn * WN + i * 16 + fq * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if (p.bias) {
                const f16x4 b = *reinterpret_cast<const f16x4*>(p.bias + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)b[r];
            }
            if (rb) {
                const f16x4 b = *reinterpret_cast<const f16x4*>(rb + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)b[r];
            }
            if (p.act == FIE_ACT_GEGLU) {
                f16x2 o;
                o[0] = (half_t)(v[0] * fie_gelu(v[1]) * p.scale);
                o[1] = (half_t)(v[2] * fie_gelu(v[3]) * p.scale);
                *reinterpret_cast<f16x2*>(p.C + (int64_t)m * p.ldc + (n >> 1)) = o;
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = v[r];
                if (p.act == FIE_ACT_SILU) x = fie_silu(x);
                else if (p.act == FIE_ACT_GELU) x = fie_gelu(x);
                else if (p.act == FIE_ACT_QUICK_GELU) x = fie_qgelu(x);
                v[r] = x * p.scale;
            }
            if (p.res) {
                const f16x4 b = *reinterpret_cast<const f16x4*>(p.res + (int64_t)m * p.ldr + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)b[r];
            }
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
            *reinterpret_cast<f16x4*>(p.C + (int64_t)m * p.ldc + n) = o;
        }
    }
}

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int RA = BM / 32, RW = BN / 32;       // 16-byte chunks per thread per K-step
    __shared__ __attribute__((aligned(16))) half_t smem[2 * (BM + BN) * BK];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    // XCD-aware tile order: consecutive ids on one XCD share the activation rows (n fastest)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (p.order ? bid % p.nbm : bid / p.nbn) * BM;
    const int n0 = (p.order ? bid / p.nbm : bid % p.nbn) * BN;

    const int c8 = tid & 7;          // chunk column inside the K-step
    const int r0 = tid >> 3;         // 0..31

    // ---- per-thread A-row descriptors
    int64_t a_base[RA];
    int a_ih[RA], a_iw[RA];
    bool a_ok[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + r0 + 32 * i;
        a_ok[i] = m < p.M;
        if (MODE == 1) {
            const int hw = p.OH * p.OW;
            const int b = m / hw, rem = m - b * hw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_ih[i] = oh * p.stride - p.pt;
            a_iw[i] = ow * p.stride - p.pl;
            a_base[i] = (int64_t)b * p.H * p.W * p.Cin;
        } else {
            a_base[i] = (int64_t)m;
            a_ih[i] = a_iw[i] = 0;
        }
    }
    int tap = 0, ci = c8 * 8;        // conv: position of this thread's chunk in (tap, channel) space
    if (MODE == 1) {
        while (ci >= p.Cin) { ci -= p.Cin; ++tap; }
    }

    f16x8 ra[RA], rw[RW];
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_tiles = [&](int kt) {
        const int k = kt * BK + c8 * 8;
        if (MODE == 1) {
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            const int hlim = p.H << p.ups, wlim = p.W << p.ups;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int ih = a_ih[i] + ky, iw = a_iw[i] + kx;
                const bool ok = a_ok[i] && tap < 9 && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
                const int64_t off = a_base[i] + ((int64_t)((ih >> p.ups) * p.W + (iw >> p.ups))) * p.Cin + ci;
                ra[i] = ok ? *reinterpret_cast<const f16x8*>(p.A1 + off) : zero8;
            }
            ci += BK;
            while (ci >= p.Cin) { ci -= p.Cin; ++tap; }
        } else {
            const bool k1 = k < p.K1;
            const half_t* src = k1 ? p.A1 : p.A2;
            const int64_t ld = k1 ? p.lda1 : p.lda2;
            const int kk = k1 ? k : k - p.K1;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const bool ok = a_ok[i] && k < p.K;
                ra[i] = ok ? *reinterpret_cast<const f16x8*>(src + a_base[i] * ld + kk) : zero8;
            }
        }
#pragma unroll
        for (int i = 0; i < RW; ++i)
            rw[i] = *reinterpret_cast<const f16x8*>(p.Wt + (int64_t)(n0 + r0 + 32 * i) * p.ldw + k);
    };
    auto store_tiles = [&](int buf) {
        half_t* sa = smem + buf * (BM + BN) * BK;
        half_t* sw = sa + BM * BK;
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f16x8*>(sa + lds_off(r0 + 32 * i, c8)) = ra[i];
#pragma unroll
        for (int i = 0; i < RW; ++i) *reinterpret_cast<f16x8*>(sw + lds_off(r0 + 32 * i, c8)) = rw[i];
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const half_t* sa = smem + buf * (BM + BN) * BK;
        const half_t* sw = sa + BM * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 fw[FN], fa[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i)
                fw[i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * WN + i * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int j = 0; j < FM; ++j)
                fa[j] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * WM + j * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    epilogue<FM, FN, WM, WN>(p, acc, m0, n0, wm, wn, lane);
}


// =====================================================================================================================
// v2 mainloop: ST-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4), counted vmcnt + one raw s_barrier per
// K-step.  A wave-instruction writes 1 KiB linearly = 8 tile rows x 128 B, so the XOR swizzle is applied to the per-lane
// SOURCE chunk (lane l fetches logical chunk (l&7)^(l>>3) of row l>>3) and again on the fragment reads.  Out-of-range
// im2col / M / K lanes fetch from a zero page.  MODE 0 = GEMM, 1 = generic conv, 2 = conv with Cin % 64 == 0 (tap-major
// K-steps: per-row gather state is recomputed only when the tap changes).
__device__ __attribute__((aligned(64))) half_t g_zero_page[64];

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// wait until at most `later` K-steps' worth of loads (NP per step, issued after the stage about to be consumed) are in flight
template <int NP, int ST>
__device__ __forceinline__ void wait_stage(int later) {
    if (ST >= 5 && later == 3) wait_vm_barrier<(ST >= 5 ? 3 : 0) * NP>();
    else if (ST >= 4 && later == 2) wait_vm_barrier<(ST >= 4 ? 2 : 0) * NP>();
    else if (ST >= 3 && later >= 1) wait_vm_barrier<(ST >= 3 ? 1 : 0) * NP>();
    else wait_vm_barrier<0>();
}

__device__ __forceinline__ void glds16(const half_t* src, half_t* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int BM, int BN, int ST, int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void gemm2_kernel(GemmArgs p) {
    constexpr int WGN = NW / 2;                        // waves: 2 along m x NW/2 along n
    constexpr int WM = BM / 2, WN = BN / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int RA = BM / (8 * NW), RW = BN / (8 * NW);  // LDS-DMA pieces (8 rows x 128 B) per wave per stage
    constexpr int NP = RA + RW;
    constexpr int STAGE = (BM + BN) * BK;              // halfs per stage
    extern __shared__ __attribute__((aligned(16))) half_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (p.order ? bid % p.nbm : bid / p.nbn) * BM;
    const int n0 = (p.order ? bid / p.nbm : bid % p.nbn) * BN;

    const int lr = lane >> 3;                          // row inside a piece
    const int c8 = (lane & 7) ^ lr;                    // logical 16-byte chunk this lane fetches (source-side swizzle)

    // ---- per-thread A-row descriptors: piece i of this wave covers tile rows (wave + NW i) * 8 .. + 7
    int64_t a_base[RA];
    int a_ih[RA], a_iw[RA];
    bool a_ok[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + lr;
        a_ok[i] = m < p.M;
        if (MODE != 0) {
            const int hw = p.OH * p.OW;
            const int b = m / hw, rem = m - b * hw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_ih[i] = oh * p.stride - p.pt;
            a_iw[i] = ow * p.stride - p.pl;
            a_base[i] = (int64_t)b * p.H * p.W * p.Cin;
        } else {
            a_base[i] = (int64_t)m;
            a_ih[i] = a_iw[i] = 0;
        }
    }
    const half_t* w_src[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) w_src[i] = p.Wt + (int64_t)(n0 + (wave + NW * i) * 8 + lr) * p.ldw + c8 * 8;

    // gather state
    int tap = 0, ci = c8 * 8;                          // MODE 1
    if (MODE == 1) {
        while (ci >= p.Cin) { ci -= p.Cin; ++tap; }
    }
    int cs = 0, ftap = 0;                              // MODE 2: channel step inside the tap, current tap
    const int csteps = MODE == 2 ? p.Cin / BK : 1;
    int64_t t_off[RA];                                 // MODE 2: element offset of (row, tap) incl. this lane's chunk
    bool t_ok[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) { t_off[i] = 0; t_ok[i] = false; }

    auto issue = [&](int kt, int stage) {
        half_t* sa = smem + stage * STAGE;
        half_t* sw = sa + BM * BK;
        if (MODE == 2) {
            if (cs == 0) {
                const int ky = (ftap * 11) >> 5, kx = ftap - 3 * ky;
                const int hlim = p.H << p.ups, wlim = p.W << p.ups;
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const int ih = a_ih[i] + ky, iw = a_iw[i] + kx;
                    t_ok[i] = a_ok[i] && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
                    t_off[i] = a_base[i] + ((int64_t)((ih >> p.ups) * p.W + (iw >> p.ups))) * p.Cin + c8 * 8;
                }
            }
#pragma unroll
            for (int i = 0; i < RA; ++i)
                glds16(t_ok[i] ? p.A1 + t_off[i] + cs * BK : g_zero_page, sa + (wave + NW * i) * 512);
            if (++cs == csteps) { cs = 0; ++ftap; }
        } else if (MODE == 1) {
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            const int hlim = p.H << p.ups, wlim = p.W << p.ups;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int ih = a_ih[i] + ky, iw = a_iw[i] + kx;
                const bool ok = a_ok[i] && tap < 9 && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
                const int64_t off = a_base[i] + ((int64_t)((ih >> p.ups) * p.W + (iw >> p.ups))) * p.Cin + ci;
                glds16(ok ? p.A1 + off : g_zero_page, sa + (wave + NW * i) * 512);
            }
            ci += BK;
            while (ci >= p.Cin) { ci -= p.Cin; ++tap; }
        } else {
            const int k = kt * BK + c8 * 8;
            const bool k1 = k < p.K1;
            const half_t* src = k1 ? p.A1 : p.A2;
            const int64_t ld = k1 ? p.lda1 : p.lda2;
            const int kk = k1 ? k : k - p.K1;
#pragma unroll
            for (int i = 0; i < RA; ++i)
                glds16((a_ok[i] && k < p.K) ? src + a_base[i] * ld + kk : g_zero_page, sa + (wave + NW * i) * 512);
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) glds16(w_src[i] + kt * BK, sw + (wave + NW * i) * 512);
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s < nk) issue(s, s);

    const int fr = lane & 15, fq = lane >> 4;
    int stage = 0, fill = ST - 1;                      // stage being consumed, stage to refill
    for (int kt = 0; kt < nk; ++kt) {
        // groups still allowed in flight: the ones issued after stage kt's
        const int later = min(kt + ST - 2, nk - 1) - kt;
        wait_stage<NP, ST>(later);
        if (kt + ST - 1 < nk) issue(kt + ST - 1, fill);
        const half_t* sa = smem + stage * STAGE;
        const half_t* sw = sa + BM * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 fw[FN], fa[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i)
                fw[i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * WN + i * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int j = 0; j < FM; ++j)
                fa[j] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * WM + j * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        stage = stage + 1 == ST ? 0 : stage + 1;
        fill = fill + 1 == ST ? 0 : fill + 1;
    }
    epilogue<FM, FN, WM, WN>(p, acc, m0, n0, wm, wn, lane);
}

template <int BM, int BN, int ST, int MODE, int NW = 4>
void launch2_t(fie_ctx* ctx, const GemmArgs& a, dim3 grid) {
    constexpr int lds = ST * (BM + BN) * BK * (int)sizeof(half_t);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<BM, BN, ST, MODE, NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm2_kernel<BM, BN, ST, MODE, NW>), grid, dim3(NW * 64), lds, ctx->stream, a);
}


// =====================================================================================================================
// v3 mainloop: as v2 (ST-stage LDS ring, counted vmcnt, one raw barrier per K-step) but the tiles are fetched with
// `buffer_load_dwordx4 ... offen lds`: a wave-uniform buffer descriptor + a per-lane 32-bit byte offset that is computed
// ONCE (GEMM) or once per 3x3 tap (conv, Cin % 64 == 0) + a scalar offset that advances per K-step.  The K loop then
// carries no per-lane address arithmetic at all, and out-of-range lanes (im2col padding, rows >= M, K tail) use the
// descriptor's range check (offset >= num_records reads as 0) instead of a zero page.  Needs every operand < 2 GiB.
constexpr unsigned kOob = 0x80000000u;

__device__ __forceinline__ void bload16(__amdgpu_buffer_rsrc_t rsrc, half_t* lds_dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

template <int BM, int BN, int ST, int MODE, int NW>    // MODE 0 = GEMM, 2 = conv with Cin % 64 == 0
__global__ __launch_bounds__(NW * 64) void gemm3_kernel(GemmArgs p) {
    constexpr int WGN = NW / 2;
    constexpr int WM = BM / 2, WN = BN / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int RA = BM / (8 * NW), RW = BN / (8 * NW);
    constexpr int NP = RA + RW;
    constexpr int STAGE = (BM + BN) * BK;
    extern __shared__ __attribute__((aligned(16))) half_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (p.order ? bid % p.nbm : bid / p.nbn) * BM;
    const int n0 = (p.order ? bid / p.nbm : bid % p.nbn) * BN;
    const int lr = lane >> 3;
    const int c8 = (lane & 7) ^ lr;

    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)p.a1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, (int)p.a2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, (int)p.w_bytes, 0x00020000);

    // per-lane byte offsets
    unsigned a_off1[RA], a_off2[RA];       // GEMM: row offsets into A1 / A2;  conv: a_off1 = offset for the current tap
    int a_ih[RA], a_iw[RA];
    unsigned a_img[RA];
    bool a_ok[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + lr;
        a_ok[i] = m < p.M;
        if (MODE == 2) {
            const int hw = p.OH * p.OW;
            const int b = m / hw, rem = m - b * hw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_ih[i] = oh * p.stride - p.pt;
            a_iw[i] = ow * p.stride - p.pl;
            a_img[i] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.Cin * 2u;
            a_off1[i] = kOob;
            a_off2[i] = 0;
        } else {
            a_off1[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda1 * 2u + c8 * 16u : kOob;
            a_off2[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda2 * 2u + c8 * 16u : kOob;
            a_ih[i] = a_iw[i] = 0;
            a_img[i] = 0;
        }
    }
    unsigned w_off[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) w_off[i] = (unsigned)(n0 + (wave + NW * i) * 8 + lr) * (unsigned)p.ldw * 2u + c8 * 16u;

    int cs = 0, ftap = 0;
    const int csteps = MODE == 2 ? p.Cin / BK : 1;
    const int k1_steps = p.K1 / BK;            // GEMM: K-steps served by A1 (K1 % 64 == 0 unless K1 == K)
    const int nk = (p.K + BK - 1) / BK;
    const bool ktail = (p.K % BK) != 0;

    auto issue = [&](int kt, int stage) {
        half_t* sa = smem + stage * STAGE;
        if (MODE == 2) {
            if (cs == 0) {
                const int ky = (ftap * 11) >> 5, kx = ftap - 3 * ky;
                const int hlim = p.H << p.ups, wlim = p.W << p.ups;
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const int ih = a_ih[i] + ky, iw = a_iw[i] + kx;
                    const bool ok = a_ok[i] && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
                    a_off1[i] = ok ? a_img[i] + (unsigned)((ih >> p.ups) * p.W + (iw >> p.ups)) * (unsigned)p.Cin * 2u + c8 * 16u : kOob;
                }
            }
            const unsigned so = (unsigned)cs * (BK * 2);
#pragma unroll
            for (int i = 0; i < RA; ++i) bload16(rs_a1, sa + (wave + NW * i) * 512, a_off1[i], so);
            if (++cs == csteps) { cs = 0; ++ftap; }
        } else {
            if (ktail && kt == nk - 1) {               // last, partial K-step: columns >= K read as zero
                const bool in_k = kt * BK + c8 * 8 < p.K;
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a1, sa + (wave + NW * i) * 512, in_k ? a_off1[i] : kOob, (unsigned)kt * (BK * 2));
            } else if (kt < k1_steps || k1_steps == 0) {
                const unsigned so = (unsigned)kt * (BK * 2);
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a1, sa + (wave + NW * i) * 512, a_off1[i], so);
            } else {
                const unsigned so = (unsigned)(kt - k1_steps) * (BK * 2);
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a2, sa + (wave + NW * i) * 512, a_off2[i], so);
            }
        }
    };
    auto issue_w = [&](int kt, int stage) {
        half_t* sw = smem + stage * STAGE + BM * BK;
        const unsigned sow = (unsigned)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < RW; ++i) bload16(rs_w, sw + (wave + NW * i) * 512, w_off[i], sow);
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s < nk) { issue(s, s); issue_w(s, s); }

    const int fr = lane & 15, fq = lane >> 4;
    int stage = 0, fill = ST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const int later = min(kt + ST - 2, nk - 1) - kt;
        wait_stage<NP, ST>(later);
        const bool more = kt + ST - 1 < nk;
        const half_t* sa = smem + stage * STAGE;
        const half_t* sw = sa + BM * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // the refill of the freed stage is issued in two halves, one in front of each MFMA group, so that no wave
            // spends a whole K-step's worth of LDS-DMA issue slots before its first MFMA
            // (measured: pays for the 8-wave blocks, costs 10 % on the 4-wave ones, which keep one burst per K-step)
            if (more) {
                if (kk == 0) { issue(kt + ST - 1, fill); if (NW != 8) issue_w(kt + ST - 1, fill); }
                else if (NW == 8) issue_w(kt + ST - 1, fill);
            }
            f16x8 fw[FN], fa[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i)
                fw[i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * WN + i * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int j = 0; j < FM; ++j)
                fa[j] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * WM + j * 16 + fr, kk * 4 + fq));
            if (NW == 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
            if (NW == 8) __builtin_amdgcn_s_setprio(0);
        }
        stage = stage + 1 == ST ? 0 : stage + 1;
        fill = fill + 1 == ST ? 0 : fill + 1;
    }
    epilogue<FM, FN, WM, WN>(p, acc, m0, n0, wm, wn, lane);
}

// v6 = v3 with PRODUCER WAVES: the LDS-DMA pieces of a K-step are issued by NPW extra waves (one per SIMD) that do nothing
// else, the NW compute waves only ds_read + MFMA.  Reason: issuing a 1 KiB DMA piece stalls the issuing wave for 60-185
// cycles (guide, "LDS-DMA piece issue cost"); in v3 every compute wave pays that 6 times per K-step in front of its MFMAs.
// Protocol (one raw barrier per K-step, all NW + NPW waves): producers wait for their own loads of stage kt (counted vmcnt),
// everybody meets at the barrier, producers then refill the stage that was consumed in step kt - 1.
template <int BM, int BN, int ST, int MODE, int NW, int NPW>
__global__ __launch_bounds__((NW + NPW) * 64) void gemm6_kernel(GemmArgs p) {
    constexpr int WGN = NW / 2;
    constexpr int WM = BM / 2, WN = BN / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int RA = BM / (8 * NPW), RW = BN / (8 * NPW);      // pieces per producer wave per K-step
    constexpr int NP = RA + RW;
    constexpr int STAGE = (BM + BN) * BK;
    static_assert((ST - 1) * NP < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) half_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (p.order ? bid % p.nbm : bid / p.nbn) * BM;
    const int n0 = (p.order ? bid / p.nbm : bid % p.nbn) * BN;
    const int nk = (p.K + BK - 1) / BK;

    if (wave >= NW) {
        // ------------------------------------------------------------------ producer
        const int pw = wave - NW;
        const int lr = lane >> 3;
        const int c8 = (lane & 7) ^ lr;
        const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)p.a1_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, (int)p.a2_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, (int)p.w_bytes, 0x00020000);
        unsigned a_off1[RA], a_off2[RA], a_img[RA], w_off[RW];
        int a_ih[RA], a_iw[RA];
        bool a_ok[RA];
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0 + (pw + NPW * i) * 8 + lr;
            a_ok[i] = m < p.M;
            if (MODE == 2) {
                const int hw = p.OH * p.OW;
                const int b = m / hw, rem = m - b * hw;
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                a_ih[i] = oh * p.stride - p.pt;
                a_iw[i] = ow * p.stride - p.pl;
                a_img[i] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.Cin * 2u;
                a_off1[i] = kOob;
                a_off2[i] = 0;
            } else {
                a_off1[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda1 * 2u + c8 * 16u : kOob;
                a_off2[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda2 * 2u + c8 * 16u : kOob;
                a_ih[i] = a_iw[i] = 0;
                a_img[i] = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) w_off[i] = (unsigned)(n0 + (pw + NPW * i) * 8 + lr) * (unsigned)p.ldw * 2u + c8 * 16u;
        int cs = 0, ftap = 0;
        const int csteps = MODE == 2 ? p.Cin / BK : 1;
        const int k1_steps = p.K1 / BK;
        const bool ktail = (p.K % BK) != 0;
        auto issue = [&](int kt, int stage) {
            half_t* sa = smem + stage * STAGE;
            half_t* sw = sa + BM * BK;
            if (MODE == 2) {
                if (cs == 0) {
                    const int ky = (ftap * 11) >> 5, kx = ftap - 3 * ky;
                    const int hlim = p.H << p.ups, wlim = p.W << p.ups;
#pragma unroll
                    for (int i = 0; i < RA; ++i) {
                        const int ih = a_ih[i] + ky, iw = a_iw[i] + kx;
                        const bool ok = a_ok[i] && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
                        a_off1[i] = ok ? a_img[i] + (unsigned)((ih >> p.ups) * p.W + (iw >> p.ups)) * (unsigned)p.Cin * 2u + c8 * 16u : kOob;
                    }
                }
                const unsigned so = (unsigned)cs * (BK * 2);
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a1, sa + (pw + NPW * i) * 512, a_off1[i], so);
                if (++cs == csteps) { cs = 0; ++ftap; }
            } else if (ktail && kt == nk - 1) {
                const bool in_k = kt * BK + c8 * 8 < p.K;
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a1, sa + (pw + NPW * i) * 512, in_k ? a_off1[i] : kOob, (unsigned)kt * (BK * 2));
            } else if (kt < k1_steps || k1_steps == 0) {
                const unsigned so = (unsigned)kt * (BK * 2);
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a1, sa + (pw + NPW * i) * 512, a_off1[i], so);
            } else {
                const unsigned so = (unsigned)(kt - k1_steps) * (BK * 2);
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a2, sa + (pw + NPW * i) * 512, a_off2[i], so);
            }
            const unsigned sow = (unsigned)kt * (BK * 2);
#pragma unroll
            for (int i = 0; i < RW; ++i) bload16(rs_w, sw + (pw + NPW * i) * 512, w_off[i], sow);
        };
#pragma unroll
        for (int s_ = 0; s_ < ST - 1; ++s_)
            if (s_ < nk) issue(s_, s_);
        int fill = ST - 1;
        for (int kt = 0; kt < nk; ++kt) {
            wait_stage<NP, ST>(min(kt + ST - 2, nk - 1) - kt);
            if (kt + ST - 1 < nk) issue(kt + ST - 1, fill);
            fill = fill + 1 == ST ? 0 : fill + 1;
        }
        return;
    }
    // ---------------------------------------------------------------------- compute waves
    const int wm = wave & 1, wn = wave >> 1;
    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_barrier" ::: "memory");
        const half_t* sa = smem + stage * STAGE;
        const half_t* sw = sa + BM * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 fw[FN], fa[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i)
                fw[i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * WN + i * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int j = 0; j < FM; ++j)
                fa[j] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * WM + j * 16 + fr, kk * 4 + fq));
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        stage = stage + 1 == ST ? 0 : stage + 1;
    }
    epilogue<FM, FN, WM, WN>(p, acc, m0, n0, wm, wn, lane);
}

template <int BM, int BN, int ST, int MODE, int NW, int NPW>
void launch6_t(fie_ctx* ctx, const GemmArgs& a, dim3 grid) {
    constexpr int lds = ST * (BM + BN) * BK * (int)sizeof(half_t);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm6_kernel<BM, BN, ST, MODE, NW, NPW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm6_kernel<BM, BN, ST, MODE, NW, NPW>), grid, dim3((NW + NPW) * 64), lds, ctx->stream, a);
}

// v5 = 256x256 tile, FOUR PHASES PER K-STEP with half-tile slot recycling (the guide's "256^2 8-phase" idea restated for this
// data path).  v3's 256x256 build has room for two stages only, so every K-step ends on vmcnt(0): the DMA issued one step
// ago must land before anything proceeds.  Here the two 64 KiB buffers are recycled at half-tile granularity instead:
//   * a K-step is four phases of 16 MFMAs (one 64x32 quadrant of the wave's 128x64 tile each), each phase = its ds_reads,
//     ONE half-tile of DMA (2 pieces per wave), the MFMAs, one raw s_barrier;
//   * phase order (A0,W0) (A0,W1) (A1,W1) (A1,W0) with W0's fragments kept in registers: the step's W slots are dead after
//     phase 2 and its A slots after phase 3, so phases 3/4 already refill the W slots with step t+2 while phases 1/2 fill the
//     other buffer's A slots with step t+1;
//   * every half-tile is therefore issued >= 3 phases before its first read and ONE counted wait per K-step (vmcnt(4): the
//     two youngest half-tiles may still fly) + the phase-4 barrier orders it -- vmcnt never drains inside the loop.
// RAW: a half-tile is read only after the issuing waves' counted wait and a barrier (end of phase 4).  WAR: a slot is refilled
// only after a barrier that follows the phase holding its last ds_read (whose data the MFMAs of that phase consumed).
template <int MODE>    // 0 = GEMM, 2 = conv with Cin % 64 == 0
__global__ __launch_bounds__(512) void gemm5_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = 256, NW = 8, WM = 128, WN = 64, FM = 8, FN = 4;
    constexpr int BUF = (BM + BN) * BK;          // halves per buffer: [A 256 rows | W 256 rows]
    extern __shared__ __attribute__((aligned(16))) half_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (p.order ? bid % p.nbm : bid / p.nbn) * BM;
    const int n0 = (p.order ? bid / p.nbm : bid % p.nbn) * BN;
    const int lr = lane >> 3;
    const int c8 = (lane & 7) ^ lr;

    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)p.a1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, (int)p.a2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, (int)p.w_bytes, 0x00020000);

    // piece i of this wave = rows (wave + 8 i) * 8 + lr of the 256-row tile: i = 0, 1 lie in half 0, i = 2, 3 in half 1
    unsigned a_off1[4], a_off2[4], a_img[4], w_off[4];
    int a_ih[4], a_iw[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + lr;
        a_ok[i] = m < p.M;
        if (MODE == 2) {
            const int hw = p.OH * p.OW;
            const int b = m / hw, rem = m - b * hw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_ih[i] = oh * p.stride - p.pt;
            a_iw[i] = ow * p.stride - p.pl;
            a_img[i] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.Cin * 2u;
            a_off1[i] = kOob;
            a_off2[i] = 0;
        } else {
            a_off1[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda1 * 2u + c8 * 16u : kOob;
            a_off2[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda2 * 2u + c8 * 16u : kOob;
            a_ih[i] = a_iw[i] = 0;
            a_img[i] = 0;
        }
        w_off[i] = (unsigned)(n0 + (wave + NW * i) * 8 + lr) * (unsigned)p.ldw * 2u + c8 * 16u;
    }
    const int csteps = MODE == 2 ? p.Cin / BK : 1;
    const int k1_steps = p.K1 / BK;
    const int nk = (p.K + BK - 1) / BK;
    const bool ktail = (p.K % BK) != 0;

    // A half `h` of K-step kt -> buffer kt & 1.  Conv: the tap of K-step kt is fixed when its half 0 is issued.
    int a_cs = 0, a_tap = 0;           // channel block / tap of the NEXT K-step whose A half 0 will be issued
    unsigned a_so = 0;                 // scalar offset of the K-step being issued (set with half 0, reused by half 1)
    int a_src = 0;                     // GEMM: 0 = A1, 1 = A2, 2 = partial last step of A1
    auto issue_a = [&](int kt, int h) {
        half_t* dst = smem + (kt & 1) * BUF;
        if (h == 0) {
            if (MODE == 2) {
                if (a_cs == 0) {
                    const int ky = (a_tap * 11) >> 5, kx = a_tap - 3 * ky;
                    const int hlim = p.H << p.ups, wlim = p.W << p.ups;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ih = a_ih[i] + ky, iw = a_iw[i] + kx;
                        const bool ok = a_ok[i] && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
                        a_off1[i] = ok ? a_img[i] + (unsigned)((ih >> p.ups) * p.W + (iw >> p.ups)) * (unsigned)p.Cin * 2u + c8 * 16u : kOob;
                    }
                }
                a_so = (unsigned)a_cs * (BK * 2);
                if (++a_cs == csteps) { a_cs = 0; ++a_tap; }
            } else if (ktail && kt == nk - 1) {
                a_src = 2;
                a_so = (unsigned)kt * (BK * 2);
            } else if (kt < k1_steps || k1_steps == 0) {
                a_src = 0;
                a_so = (unsigned)kt * (BK * 2);
            } else {
                a_src = 1;
                a_so = (unsigned)(kt - k1_steps) * (BK * 2);
            }
        }
#pragma unroll
        for (int i = 2 * h; i < 2 * h + 2; ++i) {
            half_t* d = dst + (wave + NW * i) * 512;
            if (MODE == 2 || a_src == 0) bload16(rs_a1, d, a_off1[i], a_so);
            else if (a_src == 1) bload16(rs_a2, d, a_off2[i], a_so);
            else bload16(rs_a1, d, (int)(a_so >> 1) + c8 * 8 < p.K ? a_off1[i] : kOob, a_so);   // K tail: columns >= K read as zero
        }
    };
    auto issue_w = [&](int kt, int h) {
        half_t* dst = smem + (kt & 1) * BUF + BM * BK;
#pragma unroll
        for (int i = 2 * h; i < 2 * h + 2; ++i) bload16(rs_w, dst + (wave + NW * i) * 512, w_off[i], (unsigned)kt * (BK * 2));
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: A(0), W(0) into buffer 0, W(1) into buffer 1 (A(1) follows in phases 1-2 of step 0)
    issue_a(0, 0); issue_a(0, 1); issue_w(0, 0); issue_w(0, 1);
    if (nk > 1) { issue_w(1, 0); issue_w(1, 1); wait_vm_barrier<4>(); }
    else wait_vm_barrier<0>();

    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const half_t* sa = smem + (kt & 1) * BUF;
        const half_t* sw = sa + BM * BK;
        const bool next1 = kt + 1 < nk, next2 = kt + 2 < nk;
        f16x8 fa[2][4], fw0[2][2], fw1[2][2];
        // ---- phase 1: quadrant (A0, W0)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fw0[kk][i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * WN + i * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[kk][j] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * WM + j * 16 + fr, kk * 4 + fq));
        }
        if (next1) issue_a(kt + 1, 0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw0[kk][i], fa[kk][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_barrier" ::: "memory");
        // ---- phase 2: quadrant (A0, W1); last read of this step's W slots
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) fw1[kk][i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * WN + 32 + i * 16 + fr, kk * 4 + fq));
        if (next1) issue_a(kt + 1, 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw1[kk][i], fa[kk][j], acc[2 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_barrier" ::: "memory");
        // ---- phase 3: quadrant (A1, W1); last read of this step's A slots; the W slots are free: refill with step kt + 2
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[kk][j] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * WM + 64 + j * 16 + fr, kk * 4 + fq));
        if (next2) issue_w(kt + 2, 0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[2 + i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw1[kk][i], fa[kk][j], acc[2 + i][4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_barrier" ::: "memory");
        // ---- phase 4: quadrant (A1, W0) from registers; then the step's one counted wait
        if (next2) issue_w(kt + 2, 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw0[kk][i], fa[kk][j], acc[i][4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (next2) wait_vm_barrier<4>();      // A(kt+1) and W(kt+1) have landed; W(kt+2) may still be in flight
        else wait_vm_barrier<0>();
    }
    epilogue<FM, FN, WM, WN>(p, acc, m0, n0, wm, wn, lane);
}

template <int MODE>
void launch5_t(fie_ctx* ctx, const GemmArgs& a, dim3 grid) {
    constexpr int lds = 2 * 512 * BK * (int)sizeof(half_t);      // 128 KiB
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm5_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm5_kernel<MODE>), grid, dim3(512), lds, ctx->stream, a);
}

// v4 = v3's data path with a PING-PONG schedule: the block's 8 waves form two groups of 4 (one wave per SIMD each).  A K-step
// is split into a load phase (LDS-DMA issue for step kt+2, then ALL fragment ds_reads of step kt into registers) and an
// MFMA phase (the 2 x FN x FM MFMAs of step kt), each closed by a raw s_barrier; group 1 runs one barrier behind group 0, so
// on every SIMD one wave streams LDS while the other feeds the matrix pipe.  Stage reuse: the DMA for step kt+2 targets the
// stage both groups finished reading (lgkmcnt(0) before the barrier) at least one phase earlier.
template <int BM, int BN, int ST, int MODE, int NW>    // MODE 0 = GEMM, 2 = conv with Cin % 64 == 0; NW == 8, ST == 3
__global__ __launch_bounds__(NW * 64) void gemm4_kernel(GemmArgs p) {
    constexpr int WGN = NW / 2;
    constexpr int WM = BM / 2, WN = BN / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int RA = BM / (8 * NW), RW = BN / (8 * NW);
    constexpr int NP = RA + RW;
    constexpr int STAGE = (BM + BN) * BK;
    extern __shared__ __attribute__((aligned(16))) half_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (p.order ? bid % p.nbm : bid / p.nbn) * BM;
    const int n0 = (p.order ? bid / p.nbm : bid % p.nbn) * BN;
    const int lr = lane >> 3;
    const int c8 = (lane & 7) ^ lr;

    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)p.a1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, (int)p.a2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, (int)p.w_bytes, 0x00020000);

    // per-lane byte offsets
    unsigned a_off1[RA], a_off2[RA];       // GEMM: row offsets into A1 / A2;  conv: a_off1 = offset for the current tap
    int a_ih[RA], a_iw[RA];
    unsigned a_img[RA];
    bool a_ok[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + lr;
        a_ok[i] = m < p.M;
        if (MODE == 2) {
            const int hw = p.OH * p.OW;
            const int b = m / hw, rem = m - b * hw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_ih[i] = oh * p.stride - p.pt;
            a_iw[i] = ow * p.stride - p.pl;
            a_img[i] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.Cin * 2u;
            a_off1[i] = kOob;
            a_off2[i] = 0;
        } else {
            a_off1[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda1 * 2u + c8 * 16u : kOob;
            a_off2[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda2 * 2u + c8 * 16u : kOob;
            a_ih[i] = a_iw[i] = 0;
            a_img[i] = 0;
        }
    }
    unsigned w_off[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) w_off[i] = (unsigned)(n0 + (wave + NW * i) * 8 + lr) * (unsigned)p.ldw * 2u + c8 * 16u;

    int cs = 0, ftap = 0;
    const int csteps = MODE == 2 ? p.Cin / BK : 1;
    const int k1_steps = p.K1 / BK;            // GEMM: K-steps served by A1 (K1 % 64 == 0 unless K1 == K)
    const int nk = (p.K + BK - 1) / BK;
    const bool ktail = (p.K % BK) != 0;

    auto issue = [&](int kt, int stage) {
        half_t* sa = smem + stage * STAGE;
        if (MODE == 2) {
            if (cs == 0) {
                const int ky = (ftap * 11) >> 5, kx = ftap - 3 * ky;
                const int hlim = p.H << p.ups, wlim = p.W << p.ups;
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const int ih = a_ih[i] + ky, iw = a_iw[i] + kx;
                    const bool ok = a_ok[i] && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
                    a_off1[i] = ok ? a_img[i] + (unsigned)((ih >> p.ups) * p.W + (iw >> p.ups)) * (unsigned)p.Cin * 2u + c8 * 16u : kOob;
                }
            }
            const unsigned so = (unsigned)cs * (BK * 2);
#pragma unroll
            for (int i = 0; i < RA; ++i) bload16(rs_a1, sa + (wave + NW * i) * 512, a_off1[i], so);
            if (++cs == csteps) { cs = 0; ++ftap; }
        } else {
            if (ktail && kt == nk - 1) {               // last, partial K-step: columns >= K read as zero
                const bool in_k = kt * BK + c8 * 8 < p.K;
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a1, sa + (wave + NW * i) * 512, in_k ? a_off1[i] : kOob, (unsigned)kt * (BK * 2));
            } else if (kt < k1_steps || k1_steps == 0) {
                const unsigned so = (unsigned)kt * (BK * 2);
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a1, sa + (wave + NW * i) * 512, a_off1[i], so);
            } else {
                const unsigned so = (unsigned)(kt - k1_steps) * (BK * 2);
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a2, sa + (wave + NW * i) * 512, a_off2[i], so);
            }
        }
    };
    auto issue_w = [&](int kt, int stage) {
        half_t* sw = smem + stage * STAGE + BM * BK;
        const unsigned sow = (unsigned)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < RW; ++i) bload16(rs_w, sw + (wave + NW * i) * 512, w_off[i], sow);
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    static_assert(NW == 8 && ST == 3, "ping-pong schedule: 8 waves, 3 stages");
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s < nk) { issue(s, s); issue_w(s, s); }

    const int fr = lane & 15, fq = lane >> 4;
    const int group = wave >> 2;                       // wave-uniform: group 1 trails group 0 by one barrier
    if (nk > 1) wait_vm_barrier<NP>(); else wait_vm_barrier<0>();     // step 0 landed for every wave
    if (group == 1) asm volatile("s_barrier" ::: "memory");
    int stage = 0, fill = ST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // ---- load phase.  Entry condition (guaranteed by the previous barrier): stage kt has landed for every wave.
        const bool more = kt + ST - 1 < nk;
        if (more) { issue(kt + ST - 1, fill); issue_w(kt + ST - 1, fill); }
        const half_t* sa = smem + stage * STAGE;
        const half_t* sw = sa + BM * BK;
        f16x8 fw[2][FN], fa[2][FM];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < FN; ++i)
                fw[kk][i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * WN + i * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int j = 0; j < FM; ++j)
                fa[kk][j] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * WM + j * 16 + fr, kk * 4 + fq));
        }
        // close the phase: fragments in registers (so the stage may be refilled), and every step but the newest landed
        if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA phase
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[kk][i], fa[kk][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        stage = stage + 1 == ST ? 0 : stage + 1;
        fill = fill + 1 == ST ? 0 : fill + 1;
    }
    if (group == 0) asm volatile("s_barrier" ::: "memory");   // match group 1's extra barrier: equal barrier counts per wave
    epilogue<FM, FN, WM, WN>(p, acc, m0, n0, wm, wn, lane);
}


template <int BM, int BN, int MODE>
void launch4_t(fie_ctx* ctx, const GemmArgs& a, dim3 grid) {
    constexpr int lds = 3 * (BM + BN) * BK * (int)sizeof(half_t);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<BM, BN, 3, MODE, 8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm4_kernel<BM, BN, 3, MODE, 8>), grid, dim3(512), lds, ctx->stream, a);
}

int g_extra_lds = 0;    // tuning hook (fie_debug_extra_lds): pad the v3 kernels' dynamic LDS to lower their occupancy (A/B probe)

template <int BM, int BN, int ST, int MODE, int NW = 4>
void launch3_t(fie_ctx* ctx, const GemmArgs& a, dim3 grid) {
    constexpr int lds0 = ST * (BM + BN) * BK * (int)sizeof(half_t);
    const int lds = lds0 + g_extra_lds > 160 * 1024 ? 160 * 1024 : lds0 + g_extra_lds;
    static int attr = 0;
    if (attr < lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<BM, BN, ST, MODE, NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = lds;
    }
    hipLaunchKernelGGL((gemm3_kernel<BM, BN, ST, MODE, NW>), grid, dim3(NW * 64), lds, ctx->stream, a);
}


// =====================================================================================================================
// conv_halo_kernel: 3x3 / stride 1 / pad 1 convolution with INPUT REUSE ACROSS TAPS.  The im2col kernels above fetch
// every input pixel nine times (once per tap); here a block owns a 16x16 output patch, loads its 18x18x64-channel input
// halo ONCE per 64-channel block into LDS (rows = halo pixels, same 128-B swizzled rows) and runs the nine taps against
// it -- only the 16 KiB weight slice changes per K-step.  Global->LDS traffic per 9 K-steps drops from 9*(32+16) KiB to
// 48 + 9*16 KiB, which lifts the load-path cap that bounds the im2col kernels.
// 8 waves = 2 (patch halves of 8 rows) x 4 (32 output channels each); buffer-load LDS-DMA, offsets computed once per
// tile; weights in a 3-stage ring (2 steps ahead), halo in a 2-stage ring (one channel block ahead).
constexpr int HALO_ROWS = 384;                 // 18*18 = 324 halo pixels, padded to 48 pieces x 8 rows

template <int BN>
__global__ __launch_bounds__(512) void conv_halo_kernel(GemmArgs p) {
    constexpr int NW = 8, BM = 256, WM = 128, WN = BN / 4, FM = 8, FN = WN / 16;
    constexpr int RWP = BN / (8 * NW);         // weight pieces per wave per step
    constexpr int RAP = HALO_ROWS / (8 * NW);  // halo pieces per wave per channel block (6)
    constexpr int WSTAGE = BN * BK, ASTAGE = HALO_ROWS * BK;
    extern __shared__ __attribute__((aligned(16))) half_t smem[];
    half_t* const s_a = smem;                  // 2 halo stages
    half_t* const s_w = smem + 2 * ASTAGE;     // 3 weight stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int patch = bid / p.nbn, n0 = (bid % p.nbn) * BN;
    const int ppx = p.W >> 4, ppi = ppx * (p.H >> 4);
    const int b = patch / ppi, prem = patch - b * ppi;
    const int y0 = (prem / ppx) << 4, x0 = (prem - (prem / ppx) * ppx) << 4;
    const int m0 = (b * p.H + y0) * p.W + x0;          // output row index of the patch's top-left pixel

    const int lr = lane >> 3, c8 = (lane & 7) ^ lr;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)p.a1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, (int)p.w_bytes, 0x00020000);
    unsigned a_off[RAP], w_off[RWP];
#pragma unroll
    for (int i = 0; i < RAP; ++i) {
        const int r = (wave + NW * i) * 8 + lr;        // halo row = hy * 18 + hx
        const int hy = r / 18, hx = r - hy * 18;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = r < 324 && y >= 0 && y < p.H && x >= 0 && x < p.W;
        a_off[i] = ok ? ((unsigned)((b * p.H + y) * p.W + x) * (unsigned)p.Cin + c8 * 8u) * 2u : kOob;
    }
#pragma unroll
    for (int i = 0; i < RWP; ++i) w_off[i] = (unsigned)(n0 + (wave + NW * i) * 8 + lr) * (unsigned)p.ldw * 2u + c8 * 16u;

    const int ncb = p.Cin / BK, total = ncb * 9;
    auto issue_a = [&](int cb) {
        half_t* dst = s_a + (cb & 1) * ASTAGE;
#pragma unroll
        for (int i = 0; i < RAP; ++i) bload16(rs_x, dst + (wave + NW * i) * 512, a_off[i], (unsigned)cb * (BK * 2));
    };
    auto issue_w = [&](int s) {                        // step s = cb * 9 + tap; weight k-offset = tap * Cin + cb * 64
        const int cb = s / 9, tap = s - cb * 9;
        half_t* dst = s_w + (s % 3) * WSTAGE;
        const unsigned so = (unsigned)(tap * p.Cin + cb * BK) * 2u;
#pragma unroll
        for (int i = 0; i < RWP; ++i) bload16(rs_w, dst + (wave + NW * i) * 512, w_off[i], so);
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_a(0);
    issue_w(0);
    if (total > 1) issue_w(1);

    const int fr = lane & 15, fq = lane >> 4;
    int cb = 0, tap = 0;
    bool a_issued_prev = false;                        // halo DMA issued during the previous step (younger than W(s))
    for (int s = 0; s < total; ++s) {
        // W(s) (and the halo of this channel block) must have landed: allow only what was issued after W(s)
        const bool w_next = s + 1 < total;
        if (a_issued_prev) { if (w_next) wait_vm_barrier<RAP + RWP>(); else wait_vm_barrier<RAP>(); }
        else { if (w_next) wait_vm_barrier<RWP>(); else wait_vm_barrier<0>(); }
        a_issued_prev = false;
        if (tap == 0 && cb + 1 < ncb) { issue_a(cb + 1); a_issued_prev = true; }
        if (s + 2 < total) issue_w(s + 2);
        const half_t* sa = s_a + (cb & 1) * ASTAGE;
        const half_t* sw = s_w + (s % 3) * WSTAGE;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 fw[FN], fa[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i)
                fw[i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * WN + i * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int r = (wm * 8 + j + ky) * 18 + kx + fr;      // halo row of output pixel (ty = wm*8+j, tx = fr), this tap
                fa[j] = *reinterpret_cast<const f16x8*>(sa + lds_off(r, kk * 4 + fq));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        if (++tap == 9) { tap = 0; ++cb; }
    }
    epilogue<FM, FN, WM, WN, true>(p, acc, m0, n0, wm, wn, lane);
}

template <int BN>
void launch_halo(fie_ctx* ctx, const GemmArgs& a, dim3 grid) {
    constexpr int lds = (2 * HALO_ROWS + 3 * BN) * BK * (int)sizeof(half_t);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<BN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((conv_halo_kernel<BN>), grid, dim3(512), lds, ctx->stream, a);
}

// tuning hook (fie_debug_tile_override): per-shape tile codes for whole-pipeline A/B runs, "mode,M,N,K=code;..." (mode 0 GEMM, 1 conv)
struct TileOverride { int mode, M, N, K, code; };
TileOverride g_overrides[32];
int g_n_overrides = 0;

thread_local char g_last_kernel[96] = "";     // what the last launch<> of this thread chose (bench.py names the roofline kernel with it)
int g_force_order = -1;  // tuning hook: -1 = estimate, 0 / 1 = force the tile order
int g_force_tile = 0;   // tuning hook (fie_debug_force_tile): 0 = heuristic, 1 = 128x128, 2 = 128x64, 3 = 64x64

// tile codes: 1 = 128x128, 2 = 128x64, 3 = 64x64 (v1 register-staged kernel); 11/12/13 = v2 LDS-DMA ring, 3 stages;
// 21/22/23 = v2, 4 stages (128x128 has no 4-stage build: 128 KiB)
template <int MODE>
int launch(fie_ctx* ctx, GemmArgs& a) {
    auto blocks = [&](int bm, int bn) { return (int64_t)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
    // variant choice from tools/microbench.py on MI355X (profiles/r01_microbench.md).  v3 (buffer-load LDS-DMA ring) wins
    // wherever it is eligible; 128x64 x 3 stages is the workhorse, 64x64 for grids that would not fill the CUs, the
    // 8-wave 128x128 block for large N x K.  Ineligible shapes fall back to the v2 / v1 kernels.
    const int64_t cus = ctx->num_cus;
    const bool ok3 = a.a1_bytes < (1ll << 31) && a.a2_bytes < (1ll << 31) && a.w_bytes < (1ll << 31) &&
                     (MODE == 1 ? a.Cin % BK == 0 : (a.K1 == a.K || a.K1 % BK == 0));
    // Global->LDS bandwidth per CU (~30 B/clk measured) caps a tile at BM*BN/(BM+BN) FLOP per byte: 256x128 / 256x256
    // blocks (8 waves) where the grid still fills the chip, 128x64 otherwise, 64x64 for the smallest grids.
    const int64_t b62 = blocks(256, 128), b61 = blocks(256, 256);
    // 160-wide tiles (N = k*160 everywhere in SD): one or two exact rounds of one block per CU on the 32x32 / 64x64-latent layers
    auto rounds = [&](int64_t nb) { return (nb * 10 >= cus * 9 && nb <= cus) || (nb * 10 >= cus * 18 && nb <= 2 * cus); };
    static const bool no_160 = getenv("FIE_160_TILES") == nullptr;              // off by default: neutral in the real pipeline
    const bool n160 = !no_160 && ok3 && a.N % 160 == 0 && a.M <= 8192;
    // producer-wave kernels (gemm6, codes 65-69: the LDS-DMA issue moved off the MFMA waves) win where a CU holds one block
    // and K is long: 32x32-latent convs and FF2 (128x128, <= 1 block per CU), 64x64-latent convs (256x128), 77-token GEMMs (64x64)
    const int64_t b66 = blocks(128, 128);
    static const bool no_producer = getenv("FIE_PRODUCER_TILES") == nullptr;   // off by default: +2.5 % UNet time in the real pipeline (cold weights), see DESIGN.md
    int code;
    if (MODE == 1) {
        if (!ok3) code = a.Cin % BK == 0 ? 12 : 2;
        else if (!no_producer && a.N % 128 == 0 && b66 * 2 >= cus && b66 <= cus) code = 66;
        else if (!no_producer && a.N % 128 == 0 && a.K >= 8192 && b66 <= cus * 5 / 2) code = 66;
        else if (!no_producer && a.N % 128 == 0 && b62 * 2 >= cus && b62 <= cus) code = 65;
        else if (n160 && rounds(blocks(128, 160))) code = 92;
        else if (n160 && blocks(64, 160) * 10 >= cus * 9 && blocks(64, 160) <= cus) code = 91;
        else if (a.N % 256 == 0 && b61 >= 2 * cus) code = 61;
        else if (a.N % 128 == 0 && b62 >= 150) code = 62;
        else code = 42;
    } else if (!no_producer && ok3 && a.M <= 256 && a.K >= 1024) {
        code = 69;
    } else if (!no_producer && ok3 && a.N % 128 == 0 && a.K >= 1024 && b66 * 2 >= cus && b66 <= cus) {
        code = 66;
    } else if (n160 && a.K >= 2048 && a.N <= 1280 && rounds(blocks(128, 160))) {
        code = 92;
    } else if (ok3 && a.N % 256 == 0 && a.K >= 2048 && b61 >= 2 * cus) {
        code = 61;
    } else if (ok3 && a.N % 128 == 0 && a.N >= 1536 && a.K >= 512 && b62 >= 200) {
        code = 62;
    } else if (a.N >= 2048 && a.K >= 1024 && blocks(128, 128) >= cus) {
        code = ok3 ? 51 : 1;
    } else if (blocks(128, 64) >= cus * 7 / 2 || a.K >= 4096 || (a.K >= 1024 && blocks(128, 64) >= cus)) {
        code = ok3 ? 42 : 2;      // last clause: the M 2048 x N 1280 x K 1280 projections, -0.75 % UNet time in tools/tile_trials.py
    } else {
        code = ok3 ? 43 : 13;
    }
    for (int i = 0; i < g_n_overrides; ++i)
        if (g_overrides[i].mode == (MODE == 1) && g_overrides[i].M == a.M && g_overrides[i].N == a.N && g_overrides[i].K == a.K)
            code = g_overrides[i].code;
    const int override_order = code >= 1000 ? code / 1000 - 1 : -1;     // 1000 + code: order 0, 2000 + code: order 1 (overrides only)
    code %= 1000;
    if (g_force_tile) code = g_force_tile;
    if (code == 70) {
        // column split for GEMMs whose 256x256 grid is a little more than one round (FF1: 8 x 40 tiles on 256 CUs): the first
        // floor(CUs / row tiles) column tiles run as exactly one round of 256x256 tiles, the remaining columns as 256x128 tiles
        const int nbm256 = (a.M + 255) / 256;
        const int n1 = (int)(cus / nbm256) * 256;
        FIE_REQUIRE(MODE == 0 && ok3 && n1 > 0 && n1 < a.N && a.N % 256 == 0, "tile code 70: shape not eligible for the column split");
        const int save_force = g_force_tile;
        GemmArgs p1 = a, p2 = a;
        p1.N = n1;
        p2.N = a.N - n1;
        p2.Wt = a.Wt + (int64_t)n1 * a.ldw;
        p2.w_bytes = a.w_bytes - (int64_t)n1 * a.ldw * 2;
        if (a.bias) p2.bias = a.bias + n1;
        if (a.rowbias) p2.rowbias = a.rowbias + n1;
        if (a.res) p2.res = a.res + n1;
        p2.C = a.C + (a.act == FIE_ACT_GEGLU ? n1 / 2 : n1);
        g_force_tile = 61;
        int rc = launch<MODE>(ctx, p1);
        g_force_tile = 62;
        if (rc == FIE_OK) rc = launch<MODE>(ctx, p2);
        g_force_tile = save_force;
        return rc;
    }
    const int tile = code % 10, ver = code / 10;
    int bm = tile == 3 ? 64 : 128, bn = tile == 1 ? 128 : 64;
    if (ver == 4 && tile >= 4) { bm = tile == 4 ? 128 : 64; bn = 64; }               // 44 = 128x64 x 4 stages, 45 / 46 = 64x64 x 4 / 6 stages
    if (ver == 6) {      // 61-63 v3 8-wave tiles, 64 four-phase 256x256, 65-69 producer-wave kernels (gemm6)
        static const int bms[10] = {0, 256, 256, 256, 256, 256, 128, 128, 128, 64}, bns[10] = {0, 256, 128, 320, 256, 128, 128, 64, 128, 64};
        bm = bms[tile]; bn = bns[tile];
    }           // 64 = 256x256, four phases per K-step (gemm5)
    if (ver == 7) { bm = tile == 1 ? 128 : 256; bn = 128; }
    if (ver == 9) { bm = tile == 2 ? 128 : 64; bn = 160; }                          // 91 = 64x160, 92 = 128x160 (4 waves, 3 stages), 93 = 64x160 x 5 stages
    if (ver == 8) { bm = 256; bn = 128; }                                           // 82 = halo-reuse conv, 16x16 patch x 128 channels                        // 71 = ping-pong 128x128, 72 = ping-pong 256x128   // 61 = 256x256 x2, 62 = 256x128 x3, 63 = 256x320 x2 stages (8 waves)
    FIE_REQUIRE(ver <= 9 && tile >= 1 && (tile <= 3 || ver == 6 || (ver == 4 && tile <= 6)) && !((ver == 3 || ver == 5 || ver == 7) && tile == 3) && (ver != 8 || tile == 2),
                "bad tile code %d", code);
    a.nbm = (a.M + bm - 1) / bm;
    a.nbn = (a.N + bn - 1) / bn;
    snprintf(g_last_kernel, sizeof(g_last_kernel), "%s tile code %d (%dx%d)", MODE == 1 ? "conv3x3" : "gemm", code, bm, bn);
    {
        // Which operand should stay resident in an XCD's 4 MiB L2?  Consecutive tile ids run on one XCD, so the fastest
        // tile index decides what is re-streamed from the Infinity Cache / HBM.  Estimate both orders' traffic.
        const double l2 = 3.0e6, abytes = (double)a.M * a.K * 2 * (MODE == 1 ? 1.0 / 9 : 1.0), wbytes = (double)a.N * a.K * 2;
        const double row_major = abytes + (wbytes <= l2 ? 8 * wbytes : a.nbm * wbytes);
        const double col_major = wbytes + (abytes <= l2 ? 8 * abytes : (wbytes / 8 <= l2 ? 8 * abytes : a.nbn * abytes));
        (void)row_major; (void)col_major;   // measured (profiles/r01_microbench.md): the estimate does not pay; rows-per-XCD stays the default
        a.order = g_force_order >= 0 ? g_force_order : (override_order >= 0 ? override_order : 0);
    }
    const dim3 grid((unsigned)(a.nbm * a.nbn)), block(256);
    if (ver == 8) {
        const bool okh = MODE == 1 && ok3 && a.stride == 1 && a.ups == 0 && a.pt == 1 && a.H % 16 == 0 && a.W % 16 == 0;
        if (!okh) { fie_set_error("tile code %d: shape not eligible for the halo conv kernel", code); return FIE_EINVAL; }
        a.nbm = a.M / 256;
        launch_halo<128>(ctx, a, dim3((unsigned)(a.nbm * a.nbn)));
        FIE_LAUNCH_CHECK();
        return FIE_OK;
    }
    if (ver >= 4) {      // v3 (buffer-load LDS-DMA): needs < 2 GiB operands, tap-aligned / K1-aligned K-steps
        if (!ok3) {
            if (g_force_tile) { fie_set_error("tile code %d: shape not eligible for the v3 kernel", code); return FIE_EINVAL; }
        } else {
            constexpr int M3 = MODE == 1 ? 2 : 0;
            if (ver == 9) {
                if (tile == 1) launch3_t<64, 160, 3, M3>(ctx, a, grid);
                else if (tile == 2) launch3_t<128, 160, 3, M3>(ctx, a, grid);
                else launch3_t<64, 160, 5, M3>(ctx, a, grid);
            } else if (ver == 7) {
                if (tile == 1) launch4_t<128, 128, M3>(ctx, a, grid);
                else launch4_t<256, 128, M3>(ctx, a, grid);
            } else if (ver == 6) {
                if (tile == 4) launch5_t<M3>(ctx, a, grid);
                else if (tile == 5) launch6_t<256, 128, 3, M3, 8, 4>(ctx, a, grid);
                else if (tile == 6) launch6_t<128, 128, 3, M3, 8, 4>(ctx, a, grid);
                else if (tile == 7) launch6_t<128, 64, 3, M3, 4, 4>(ctx, a, grid);
                else if (tile == 8) launch6_t<128, 128, 3, M3, 4, 4>(ctx, a, grid);
                else if (tile == 9) launch6_t<64, 64, 3, M3, 4, 4>(ctx, a, grid);
                else if (tile == 1) launch3_t<256, 256, 2, M3, 8>(ctx, a, grid);
                else if (tile == 2) launch3_t<256, 128, 3, M3, 8>(ctx, a, grid);
                else launch3_t<256, 320, 2, M3, 8>(ctx, a, grid);
            } else if (ver == 4) {
                if (tile == 1) launch3_t<128, 128, 3, M3>(ctx, a, grid);
                else if (tile == 2) launch3_t<128, 64, 3, M3>(ctx, a, grid);
                else if (tile == 3) launch3_t<64, 64, 3, M3>(ctx, a, grid);
                else if (tile == 4) launch3_t<128, 64, 4, M3>(ctx, a, grid);
                else if (tile == 5) launch3_t<64, 64, 4, M3>(ctx, a, grid);
                else launch3_t<64, 64, 6, M3>(ctx, a, grid);
            } else {
                if (tile == 1) launch3_t<128, 128, 3, M3, 8>(ctx, a, grid);
                else launch3_t<128, 64, 3, M3, 8>(ctx, a, grid);
            }
            FIE_LAUNCH_CHECK();
            return FIE_OK;
        }
    }
    constexpr int M2 = MODE;     // v2 conv: fast path when a K-step never straddles a tap
    const bool fast = MODE == 1 && a.Cin % BK == 0;
    if (ver == 0) {
        if (tile == 1) hipLaunchKernelGGL((gemm_kernel<128, 128, MODE>), grid, block, 0, ctx->stream, a);
        else if (tile == 2) hipLaunchKernelGGL((gemm_kernel<128, 64, MODE>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((gemm_kernel<64, 64, MODE>), grid, block, 0, ctx->stream, a);
    } else if (ver == 1) {
        if (fast) {
            if (tile == 1) launch2_t<128, 128, 3, 2>(ctx, a, grid);
            else if (tile == 2) launch2_t<128, 64, 3, 2>(ctx, a, grid);
            else launch2_t<64, 64, 3, 2>(ctx, a, grid);
        } else {
            if (tile == 1) launch2_t<128, 128, 3, M2>(ctx, a, grid);
            else if (tile == 2) launch2_t<128, 64, 3, M2>(ctx, a, grid);
            else launch2_t<64, 64, 3, M2>(ctx, a, grid);
        }
    } else if (ver == 3) {          // 8 waves per block: two waves per SIMD inside ONE block (for grids of <= 1 block per CU)
        if (fast) {
            if (tile == 1) launch2_t<128, 128, 3, 2, 8>(ctx, a, grid);
            else launch2_t<128, 64, 3, 2, 8>(ctx, a, grid);
        } else {
            if (tile == 1) launch2_t<128, 128, 3, M2, 8>(ctx, a, grid);
            else launch2_t<128, 64, 3, M2, 8>(ctx, a, grid);
        }
    } else {
        if (fast) {
            if (tile == 1) launch2_t<128, 128, 3, 2>(ctx, a, grid);
            else if (tile == 2) launch2_t<128, 64, 4, 2>(ctx, a, grid);
            else launch2_t<64, 64, 4, 2>(ctx, a, grid);
        } else {
            if (tile == 1) launch2_t<128, 128, 3, M2>(ctx, a, grid);
            else if (tile == 2) launch2_t<128, 64, 4, M2>(ctx, a, grid);
            else launch2_t<64, 64, 4, M2>(ctx, a, grid);
        }
    }
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

int check_epilogue(const char* who, int N, int64_t ldc, const void* res, int64_t ldr, int act) {
    FIE_REQUIRE(act >= FIE_ACT_NONE && act <= FIE_ACT_GEGLU, "%s: unknown act %d", who, act);
    FIE_REQUIRE(N % 4 == 0, "%s: N=%d must be a multiple of 4", who, N);
    FIE_REQUIRE(ldc % 4 == 0 || act == FIE_ACT_GEGLU, "%s: ldc=%lld must be a multiple of 4", who, (long long)ldc);
    FIE_REQUIRE(!res || ldr % 4 == 0, "%s: ldr=%lld must be a multiple of 4", who, (long long)ldr);
    FIE_REQUIRE(!(res && act == FIE_ACT_GEGLU), "%s: GEGLU epilogue takes no residual", who);
    return FIE_OK;
}

// ---- weight repack kernels
__global__ void pack_rows_kernel(const half_t* src, int64_t ld_src, int N, int K, half_t* dst, int64_t ldw, int Npad,
                                 int interleave2) {
    const int64_t total = (int64_t)Npad * ldw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / ldw), k = (int)(i - (int64_t)n * ldw);
        half_t v = (half_t)0.f;
        if (n < N && k < K) {
            const int sn = interleave2 ? ((n & 1) ? (N / 2 + (n >> 1)) : (n >> 1)) : n;
            v = src[(int64_t)sn * ld_src + k];
        }
        dst[i] = v;
    }
}

__global__ void pack_conv_kernel(const half_t* src, int Cout, int Cin, int cin_pad, half_t* dst, int64_t ldw, int Npad) {
    const int64_t total = (int64_t)Npad * ldw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / ldw), k = (int)(i - (int64_t)n * ldw);
        half_t v = (half_t)0.f;
        const int tap = k / cin_pad, ci = k - tap * cin_pad;
        if (n < Cout && tap < 9 && ci < Cin) v = src[((int64_t)n * Cin + ci) * 9 + tap];
        dst[i] = v;
    }
}

}  // namespace

extern "C" {

int fie_debug_force_tile(int t) {
    g_force_order = t >= 1000 ? (t / 1000) - 1 : -1;      // 1000 + code: order 0, 2000 + code: order 1
    g_force_tile = t % 1000;
    return FIE_OK;
}

int fie_debug_tile_override(const char* spec) {
    g_n_overrides = 0;
    if (!spec) return FIE_OK;
    const char* q = spec;
    while (*q && g_n_overrides < 32) {
        TileOverride o;
        int used = 0;
        if (sscanf(q, "%d,%d,%d,%d=%d%n", &o.mode, &o.M, &o.N, &o.K, &o.code, &used) != 5) break;
        g_overrides[g_n_overrides++] = o;
        q += used;
        if (*q == ';') ++q;
    }
    return g_n_overrides;
}

const char* fie_debug_last_gemm_kernel(void) { return g_last_kernel; }

int fie_debug_extra_lds(int bytes) {
    g_extra_lds = bytes < 0 ? 0 : bytes;
    return FIE_OK;
}

int fie_gemm_f16(fie_ctx* ctx, const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2,
                 const void* Wpacked, int64_t ldw, void* C, int64_t ldc, int M, int N, int K, const void* bias,
                 const void* rowbias, int64_t ld_rowbias, int rows_per_batch, const void* residual, int64_t ldr,
                 float scale, int act) {
    FIE_REQUIRE(ctx && A1 && Wpacked && C, "fie_gemm_f16: NULL ctx/A1/W/C");
    FIE_REQUIRE(M > 0 && N > 0 && K > 0, "fie_gemm_f16: bad shape M=%d N=%d K=%d", M, N, K);
    FIE_REQUIRE(K % 8 == 0 && K1 % 8 == 0 && K1 > 0 && K1 <= K, "fie_gemm_f16: K=%d K1=%d must be multiples of 8", K, K1);
    FIE_REQUIRE(lda1 % 8 == 0 && lda1 >= K1, "fie_gemm_f16: lda1=%lld invalid", (long long)lda1);
    FIE_REQUIRE(K1 == K || (A2 && lda2 % 8 == 0 && lda2 >= K - K1), "fie_gemm_f16: A2/lda2 invalid for K1 < K");
    FIE_REQUIRE(ldw % BK == 0 && ldw >= K, "fie_gemm_f16: ldw=%lld must be a multiple of 64 covering K", (long long)ldw);
    FIE_REQUIRE(!rowbias || rows_per_batch > 0, "fie_gemm_f16: rowbias needs rows_per_batch");
    if (int e = check_epilogue("fie_gemm_f16", N, ldc, residual, ldr, act)) return e;
    GemmArgs a = {};
    a.A1 = (const half_t*)A1; a.lda1 = lda1; a.K1 = K1; a.A2 = (const half_t*)A2; a.lda2 = lda2;
    a.Wt = (const half_t*)Wpacked; a.ldw = ldw; a.C = (half_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.bias = (const half_t*)bias; a.rowbias = (const half_t*)rowbias; a.ld_rowbias = ld_rowbias;
    a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    a.res = (const half_t*)residual; a.ldr = ldr; a.scale = scale; a.act = act;
    a.a1_bytes = ((int64_t)(M - 1) * lda1 + K1) * 2;
    a.a2_bytes = A2 ? ((int64_t)(M - 1) * lda2 + (K - K1)) * 2 : 0;
    a.w_bytes = fie_roundup(N, 128) * ldw * 2;
    return launch<0>(ctx, a);
}

int fie_conv3x3_nhwc_f16(fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, int upsample2x, int stride,
                         int pad_mode, const void* Wpacked, int64_t ldw, void* Y, int64_t ldc, int Cout,
                         const void* bias, const void* rowbias, int64_t ld_rowbias, const void* residual,
                         int64_t ldr, float scale, int act) {
    FIE_REQUIRE(ctx && X && Wpacked && Y, "fie_conv3x3_nhwc_f16: NULL ctx/X/W/Y");
    FIE_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "fie_conv3x3_nhwc_f16: bad shape");
    FIE_REQUIRE(Cin % 8 == 0, "fie_conv3x3_nhwc_f16: Cin=%d must be a multiple of 8 (pad the tensor)", Cin);
    FIE_REQUIRE(stride == 1 || stride == 2, "fie_conv3x3_nhwc_f16: stride %d", stride);
    FIE_REQUIRE(pad_mode == 0 || pad_mode == 1, "fie_conv3x3_nhwc_f16: pad_mode %d", pad_mode);
    FIE_REQUIRE(act != FIE_ACT_GEGLU, "fie_conv3x3_nhwc_f16: GEGLU not supported");
    const int K = 9 * Cin;
    FIE_REQUIRE(ldw % BK == 0 && ldw >= K, "fie_conv3x3_nhwc_f16: ldw=%lld must be a multiple of 64 covering 9*Cin",
                (long long)ldw);
    if (int e = check_epilogue("fie_conv3x3_nhwc_f16", Cout, ldc, residual, ldr, act)) return e;
    const int ups = upsample2x ? 1 : 0;
    const int Hin = H << ups, Win = W << ups;
    const int pads = pad_mode == 0 ? 2 : 1;
    const int OH = (Hin + pads - 3) / stride + 1, OW = (Win + pads - 3) / stride + 1;
    FIE_REQUIRE((int64_t)B * OH * OW < (1ll << 31), "fie_conv3x3_nhwc_f16: too many output pixels");
    GemmArgs a = {};
    a.A1 = (const half_t*)X; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.stride = stride;
    a.pt = a.pl = pad_mode == 0 ? 1 : 0; a.ups = ups;
    a.Wt = (const half_t*)Wpacked; a.ldw = ldw; a.C = (half_t*)Y; a.ldc = ldc;
    a.M = B * OH * OW; a.N = Cout; a.K = K; a.K1 = K;
    a.bias = (const half_t*)bias; a.rowbias = (const half_t*)rowbias; a.ld_rowbias = ld_rowbias;
    a.rows_per_batch = OH * OW; a.res = (const half_t*)residual; a.ldr = ldr; a.scale = scale; a.act = act;
    a.a1_bytes = (int64_t)B * H * W * Cin * 2;
    a.a2_bytes = 0;
    a.w_bytes = fie_roundup(Cout, 128) * ldw * 2;
    return launch<1>(ctx, a);
}

int fie_pack_rows_f16(fie_ctx* ctx, const void* src, int64_t ld_src, int N, int K, void* dst, int64_t ldw,
                      int Npad, int interleave2) {
    FIE_REQUIRE(ctx && src && dst, "fie_pack_rows_f16: NULL argument");
    FIE_REQUIRE(N > 0 && K > 0 && Npad >= N && ldw >= K, "fie_pack_rows_f16: bad shape");
    FIE_REQUIRE(!interleave2 || N % 2 == 0, "fie_pack_rows_f16: interleave2 needs even N");
    hipLaunchKernelGGL(pack_rows_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const half_t*)src, ld_src, N, K,
                       (half_t*)dst, ldw, Npad, interleave2);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

int fie_pack_conv3x3_f16(fie_ctx* ctx, const void* src_oihw, int Cout, int Cin, int cin_pad, void* dst, int64_t ldw,
                         int Npad) {
    FIE_REQUIRE(ctx && src_oihw && dst, "fie_pack_conv3x3_f16: NULL argument");
    FIE_REQUIRE(cin_pad >= Cin && cin_pad % 8 == 0 && ldw >= 9 * cin_pad && Npad >= Cout, "fie_pack_conv3x3_f16: bad shape");
    hipLaunchKernelGGL(pack_conv_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const half_t*)src_oihw, Cout, Cin,
                       cin_pad, (half_t*)dst, ldw, Npad);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

}  // extern "C"
