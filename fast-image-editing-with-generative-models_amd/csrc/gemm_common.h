// Shared pieces of the fp16 GEMM / 3x3 implicit-GEMM convolution kernels (gemm_conv.hip: register-staged fallback + LDS-DMA
// ring kernels; gemm8.hip: the 256x256 phased kernel).  One problem description serves both entries of include/fie.h
// (fie_gemm_f16, fie_conv3x3_nhwc_f16):
//   C[M,N] = epi(A[M,K] * W[N,K]^T), "A" = a row-major matrix (optionally the column concatenation of two matrices) or the
//   on-the-fly im2col view of an NHWC tensor (3x3 taps, stride 1/2, symmetric or VAE-style asymmetric padding, optional fused
//   nearest-2x upsample).
// MFMA operands are "swapped": the weight fragment is the A operand and the activation fragment the B operand of
// v_mfma_f32_16x16x32_f16, so a lane owns 4 CONSECUTIVE output channels of one output row (8-byte bias / residual loads and
// stores; GEGLU value / gate pairs in one lane).  LDS images are rows of 128 B (one 64-deep K-step), 16-byte chunks
// XOR-swizzled by (row & 7): applied on the SOURCE offset of the LDS-DMA and again on the ds_read_b128 fragment reads.
#pragma once
#include "fie_internal.h"

namespace fie_gemm {

constexpr int BK = 64;

struct GemmArgs {
    const half_t* A1; int64_t lda1; int K1;
    const half_t* A2; int64_t lda2;
    // conv view with 1x1 side inputs (fie_conv3x3_plus_nhwc_f16: a resnet's conv2 + its 1x1 shortcut in one GEMM): after the taps, K-steps
    // C2x / 64 read row m of A2 and C3x / 64 row m of A3 (the shortcut's input, or its two halves [x | skip])
    const half_t* A3; int64_t lda3; int C2x, C3x; int64_t a3_bytes;
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
    // 2x-upsampling conv as four 2x2 convs (fie_conv_up2x_nhwc_f16): taps2 = the conv view has 2x2 taps (K = 4 * Cin) with pt / pl = 1 - parity;
    // oscat = output row m = (b, oh, ow) of the OH x OW view is stored at pixel (2 oh + opy, 2 ow + opx) of the [B, 2 OH, 2 OW] output
    int taps2, oscat, opy, opx;             // oscat 2: all four parities in ONE launch, tile id = 4 * tile + parity (take_parity below)
    int64_t w_par_stride;                   // oscat 2: elements between the parity weight matrices
    int gn_nch, gn_chunk0;                  // GroupNorm partials: granules per image in the buffer (0: gn_rows / 32) and this launch's first granule
    float* gn_partial; int gn_rows, gn_G, gn_cg;   // GroupNorm statistics of the OUTPUT (fie_gn_stats_target): per image and 32-row granule [b][gn_rows / 32][gn_G][2] = (sum, sum of squares) of the f16-rounded values, gn_cg = N / gn_G in {4, 8, 16} channels per group; NULL: none
    const float* w_scale;                   // fp8 weights (gemm_w8.hip): per-output-channel dequantisation scale [N], applied to the accumulator first; Wt then points at e4m3 bytes and ldw counts bytes
    unsigned* stamps;                       // tile codes 97 / 98: [tile][wave][8] cycle sums of the K-loop segments (fie_debug_gemm_stamps), else NULL
    int epi_prefetch;                       // ring kernels: bias row + residual tile loaded ahead of the K loop (EpiPre below); 0 = in the epilogue (fie_debug_epilogue_prefetch)
    int probe;                              // timing-only probes (fie_debug_gemm_probe; outputs are wrong): 1 = every DMA load dropped (zero-record descriptors), 2 = every tile fetches tile (0,0)'s operands (all L2 hits), 3 = ring kernels issue no DMA inside the K loop (MFMA + ds_read + barrier floor), 4 = no epilogue (nothing stored)
    // split-K (ring kernels, fie_splitk_workspace): the K-steps of a tile are dealt to `splitk` consecutive blocks (one XCD under the remap); each
    // writes its fp32 partial tile to sk_slabs[tile][slice] (write-through), draws a ticket from sk_tickets[tile]; the block that draws the last
    // ticket sums the slices IN SLICE ORDER (deterministic) and runs the epilogue.  splitk <= 1: off
    int splitk; float* sk_slabs; unsigned* sk_tickets;
    // fp8 activations (gemm_x8.hip, BASELINE config 5): a_scale = dequantisation scale of the e4m3 A operand (0 = none; multiplies the accumulator
    // together with w_scale); out_f8 = the output is written as e4m3 bytes, value * out_inv_scale, saturated to +-448 (C then points at bytes and
    // ldc counts bytes): the consumer GEMM reads it without any conversion
    float a_scale; int out_f8; float out_inv_scale;
    int order;                              // 0: n-tiles fastest (an XCD owns a range of rows), 1: m-tiles fastest (an XCD owns a range of columns)
    // halo-resident conv (conv_halo.hip): a tile is a 16x16 PATCH of one image's output pixels, not 256 consecutive rows.  Fragment j of a wave
    // (16 pixels of one patch row) then starts frag_ld = OW rows after fragment j - 1 instead of 16: the epilogue's row of fragment (wm, j), lane fr
    // is m0 + (wm * FM + j) * frag_ld + fr with m0 = the patch's first pixel.  0 = the ordinary consecutive-row tile (16).
    int frag_ld;
    // LayerNorm folded into the GEMM (fie_gemm_ln_f16; ring tiles 42 / 48 / 96 / 64, LEAN == 2): A1 is the UN-normalised tensor, Wt = W * gamma, and with
    // (mean, rstd) of row m -- summed over the activation fragments on their way to the MFMAs -- C[m, n] = rstd * (acc - mean * ln_tab[n][0]) + ln_tab[n][1],
    // ln_tab[n] = (sum_k Wt[n, k], (W beta)[n] + bias[n]) in fp32 (packed column order).  NULL: none
    const float* ln_tab; float ln_eps;
    const float* gna_tab; int gna_silu;        // conv_halo.hip, GNA: per-(image, channel) GroupNorm coefficients (sc, sh) of the conv's INPUT (fie_groupnorm_coef_f16), applied on the resident halo
};

// oscat == 2: the block's parity comes from its (remapped) tile id; the four parity tiles of one output tile are neighbours in time and
// share the input rows in L2.  Rewrites the by-value kernel argument and returns the tile id within the parity.
__device__ __forceinline__ int take_parity(GemmArgs& p, int bid) {
    if (p.oscat != 2) return bid;
    const int par = bid & 3;
    p.opy = par >> 1; p.opx = par & 1; p.pt = 1 - p.opy; p.pl = 1 - p.opx;
    p.Wt += (int64_t)par * p.w_par_stride;
    p.gn_chunk0 = par * ((p.OH * p.OW) >> 5);
    return bid >> 2;
}

// f(integral_constant<int, 0>), f(integral_constant<int, 1>), ...: a loop whose index is a compile-time constant in every iteration
template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * BK + ((chunk ^ (row & 7)) << 3); }

// Epilogue: bias / row bias / activation / scale / residual on the fp32 accumulators, then f16 stores.
// Shape of the code matters here: written as ONE loop over the fragments with every option tested per fragment it compiled to
// ~100 scalar branches, 64-bit address arithmetic per access and a load -> vmcnt(0) -> use chain per fragment (vmcnt counts the
// earlier fragments' stores as well): 13-30 % of a kernel's time (tools/gemm_probe.py, probe 4).  Now a few straight-line sweeps
// over the accumulators, one per optional term behind one uniform branch each, and with BUF (the LDS-DMA kernels; C / residual
// spans <= 1 GiB, checked by the launcher) every access is a buffer instruction on a 32-bit offset: lanes outside the matrix
// take an out-of-range offset and the descriptor's range check drops them, so there are no masks or branches at all.
// The MFMA layout gives a lane 4 consecutive columns of one row (8-B stores, 32 B per row and instruction); transposing the
// tile through LDS into full-row 16-B stores was built and measured equal (+-3 %) once the code above was lean, so it is not here.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k + 15), result in every lane: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// The epilogue's operands that do not depend on the accumulators (bias row, residual tile), loaded BEFORE the K loop: after it they are a
// cold HBM / L2 miss of 1-2 us on the critical path of a 15-20 us kernel (round 3, tools/fit_probe.py: 6 us of a K = 1280 projection's
// 15 us do not scale with K).  16-36 VGPRs, which the LDS-limited ring kernels have to spare.  Same lane mapping and range checks as the
// epilogue below (out-of-range lanes read zero through the descriptor).
template <int FM, int FN>
struct EpiPre {
    u32x2 bias[FN];
    u32x2 res[FM][FN];
    bool on;
};
template <int FM, int FN, int WM, int WN>
__device__ __forceinline__ void epilogue_prefetch(const GemmArgs& p, EpiPre<FM, FN>& pre, int m0, int n0, int wm, int wn, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    const int mrow = m0 + wm * WM + fr, ncol = n0 + wn * WN + fq * 4;
    pre.on = true;
#pragma unroll
    for (int i = 0; i < FN; ++i) pre.bias[i] = (u32x2){0u, 0u};                      // absent operands read as zero: the lean epilogue adds them unconditionally
#pragma unroll
    for (int j = 0; j < FM; ++j)
#pragma unroll
        for (int i = 0; i < FN; ++i) pre.res[j][i] = (u32x2){0u, 0u};
    if (p.bias) {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.bias), 0, p.N * 2, 0x00020000);
#pragma unroll
        for (int i = 0; i < FN; ++i) pre.bias[i] = __builtin_amdgcn_raw_buffer_load_b64(rb, (unsigned)(ncol + i * 16) * 2u, 0, 0);
    }
    if (p.res) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.res), 0, (int)(((int64_t)(p.M - 1) * p.ldr + p.N) * 2), 0x00020000);
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int m = mrow + j * 16;
            const unsigned ro = m < p.M ? (unsigned)m * (unsigned)p.ldr * 2u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int n = ncol + i * 16;
                pre.res[j][i] = __builtin_amdgcn_raw_buffer_load_b64(rs, ro + (n < p.N ? (unsigned)n << 1 : 0xC0000000u), 0, 0);
            }
        }
    }
}

// The LEAN epilogue: (accumulator + bias) + residual -> f16 stores on buffer offsets, the two operands taken from the EpiPre registers (zero when
// absent), nothing else compiled in.  The full epilogue below is ~25 KB of code behind a dozen uniform branches and runs once per block; for the plain
// Linear layers of the transformer blocks (no row bias, activation, scale, GEGLU, GroupNorm sums, fp8 or scattered output, split-K) the 6 KB kernel
// with this epilogue is 0.7-1.2 us per launch faster and gives the same values (tools/fit_probe.py, round 3).  Selected per launch by launch_ring.
template <int FM, int FN, int WM, int WN>
__device__ __forceinline__ void epilogue_lean(const GemmArgs& p, f32x4 (&acc)[FN][FM], int m0, int n0, int wm, int wn, int lane, const EpiPre<FM, FN>& pre) {
    const int fr = lane & 15, fq = lane >> 4;
    const int mrow = m0 + wm * WM + fr, ncol = n0 + wn * WN + fq * 4;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((int64_t)(p.M - 1) * p.ldc + p.N) * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = mrow + j * 16;
        const unsigned ro = m < p.M ? (unsigned)m * (unsigned)p.ldc * 2u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int n = ncol + i * 16;
            f16x4 b, r, o;
            __builtin_memcpy(&b, &pre.bias[i], 8);
            __builtin_memcpy(&r, &pre.res[j][i], 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (half_t)(acc[i][j][q] + (float)b[q] + (float)r[q]);
            u32x2 bits;
            __builtin_memcpy(&bits, &o, 8);
            __builtin_amdgcn_raw_buffer_store_b64(bits, rs_c, ro + (n < p.N ? (unsigned)n << 1 : 0xC0000000u), 0, 0);
        }
    }
}

// LEAN epilogue of the fp8-activation kernels (gemm_x8.hip): the same with the per-output-channel dequantisation scale in front (acc * (w_scale * a_scale))
template <int FM, int FN, int WM, int WN>
__device__ __forceinline__ void epilogue_lean_scaled(const GemmArgs& p, f32x4 (&acc)[FN][FM], int m0, int n0, int wm, int wn, int lane, const EpiPre<FM, FN>& pre,
                                                     const f32x4 (&ws)[FN]) {
    const int fr = lane & 15, fq = lane >> 4;
    const int mrow = m0 + wm * WM + fr, ncol = n0 + wn * WN + fq * 4;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((int64_t)(p.M - 1) * p.ldc + p.N) * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = mrow + j * 16;
        const unsigned ro = m < p.M ? (unsigned)m * (unsigned)p.ldc * 2u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int n = ncol + i * 16;
            f16x4 b, r, o;
            __builtin_memcpy(&b, &pre.bias[i], 8);
            __builtin_memcpy(&r, &pre.res[j][i], 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (half_t)((acc[i][j][q] * (ws[i][q] * p.a_scale) + (float)b[q]) + (float)r[q]);
            u32x2 bits;
            __builtin_memcpy(&bits, &o, 8);
            __builtin_amdgcn_raw_buffer_store_b64(bits, rs_c, ro + (n < p.N ? (unsigned)n << 1 : 0xC0000000u), 0, 0);
        }
    }
}

// ---- LayerNorm folded into the consumer GEMM (GemmArgs::ln_tab).  LN(x) W^T = rstd * (x (W gamma)^T - mean * colsum(W gamma)) + W beta: the
// normalised tensor is never written or read.  The row sums come from the activation fragments the MFMAs consume anyway: in the swapped operand
// layout lane (fr, fq) holds 8 k-values of row fr, the row it also owns in the epilogue; two v_dot2c_f32_f16 per register pair (x . 1, x . x).
__device__ __forceinline__ void ln_dot(const f16x8& f, float& s, float& q) {
    const f16x2 one = {(half_t)1.f, (half_t)1.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f16x2 v = {f[2 * t], f[2 * t + 1]};
        s = __builtin_amdgcn_fdot2(v, one, s, false);
        q = __builtin_amdgcn_fdot2(v, v, q, false);
    }
}

// (colsum, bias') pairs of a lane's 4 columns of fragment column i: 32 contiguous bytes, two 16-byte buffer loads; columns >= N read zero
template <int FN>
struct LnTab { f32x4 lo[FN], hi[FN]; };       // lo = (S0, b0, S1, b1), hi = (S2, b2, S3, b3)
__device__ __forceinline__ void ln_tab_load(const GemmArgs& p, int n, f32x4& lo, f32x4& hi) {
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ln_tab), 0, p.N * 8, 0x00020000);
    lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, (unsigned)n * 8u, 0, 0));
    hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, (unsigned)n * 8u + 16u, 0, 0));
}

template <int FM, int FN, int WM, int WN>
__device__ __forceinline__ void epilogue_lean_ln(const GemmArgs& p, f32x4 (&acc)[FN][FM], int m0, int n0, int wm, int wn, int lane, const LnTab<FN>& tab,
                                                 const float (&mean)[FM], const float (&rstd)[FM]) {
    const int fr = lane & 15, fq = lane >> 4;
    const int mrow = m0 + wm * WM + fr, ncol = n0 + wn * WN + fq * 4;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((int64_t)(p.M - 1) * p.ldc + p.N) * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = mrow + j * 16;
        const unsigned ro = m < p.M ? (unsigned)m * (unsigned)p.ldc * 2u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int n = ncol + i * 16;
            f16x4 o;
            o[0] = (half_t)((acc[i][j][0] - mean[j] * tab.lo[i][0]) * rstd[j] + tab.lo[i][1]);
            o[1] = (half_t)((acc[i][j][1] - mean[j] * tab.lo[i][2]) * rstd[j] + tab.lo[i][3]);
            o[2] = (half_t)((acc[i][j][2] - mean[j] * tab.hi[i][0]) * rstd[j] + tab.hi[i][1]);
            o[3] = (half_t)((acc[i][j][3] - mean[j] * tab.hi[i][2]) * rstd[j] + tab.hi[i][3]);
            u32x2 bits;
            __builtin_memcpy(&bits, &o, 8);
            __builtin_amdgcn_raw_buffer_store_b64(bits, rs_c, ro + (n < p.N ? (unsigned)n << 1 : 0xC0000000u), 0, 0);
        }
    }
}

// The LEAN GEGLU epilogue of the big tiles (256x320: the FF1 projection): bias + value * gelu(gate) -> f16 pairs, nothing else compiled in; same
// arithmetic as the full epilogue with scale 1.  The full path of these tiles inlines the generic epilogue three times.
// Stores (round 4): a lane's two outputs of a fragment are 4 bytes; stored as they are, every instruction leaves 16 bytes in each of 16 rows and the
// fabric sees partial sectors (round-3 PMC: WRITE_SIZE 55 MB for the 21 MB output, 2.6x).  A 4x4 transpose of dwords across the four fq lane rows (two
// v_permlane16_swap + two v_permlane32_swap) regroups them: over four fragment COLUMNS lane row q ends up with the 16 contiguous bytes of column I0 + q, so
// one 16-byte store per lane writes 64 contiguous bytes per row; the columns left over (FN % 4) are transposed over four ROW fragments instead (16-byte
// stores, 16 bytes per row: fewer instructions, same sectors).
// LN: the LayerNorm-folded form (above): value / gate = rstd * (acc - mean * S) + b' from the fp32 table instead of acc + bias
template <int FM, int FN, int WM, int WN, bool LN = false>
__device__ __forceinline__ void epilogue_geglu_lean(const GemmArgs& p, f32x4 (&acc)[FN][FM], int m0, int n0, int wm, int wn, int lane,
                                                    const float* mean = nullptr, const float* rstd = nullptr) {
    static_assert(FM % 4 == 0, "left-over fragment columns are transposed over four row fragments");
    const int fr = lane & 15, fq = lane >> 4;
    const int mrow = m0 + wm * WM + fr, nbase = n0 + wn * WN;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((int64_t)(p.M - 1) * p.ldc + (p.N >> 1)) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.bias), 0, p.bias ? p.N * 2 : 0, 0x00020000);
    auto bias4 = [&](int i) { return __builtin_amdgcn_raw_buffer_load_b64(rs_b, (unsigned)(nbase + i * 16 + fq * 4) * 2u, 0, 0); };   // no bias: zero-size descriptor, reads zero
    auto pair = [&](const f32x4& a, u32x2 bb) {
        f16x4 b;
        __builtin_memcpy(&b, &bb, 8);
        f16x2 o;
        o[0] = (half_t)((a[0] + (float)b[0]) * fie_gelu(a[1] + (float)b[1]) * 1.0f);
        o[1] = (half_t)((a[2] + (float)b[2]) * fie_gelu(a[3] + (float)b[3]) * 1.0f);
        unsigned bits;
        __builtin_memcpy(&bits, &o, 4);
        return bits;
    };
    auto transpose4 = [](unsigned& r0, unsigned& r1, unsigned& r2, unsigned& r3) {      // register k of lane row q <- register q of lane row k
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r0), "+v"(r1));
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r2), "+v"(r3));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r0), "+v"(r2));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r1), "+v"(r3));
    };
    auto rowoff = [&](int m) { return m < p.M ? (unsigned)m * (unsigned)p.ldc * 2u : 0x80000000u; };
    // LN: one fragment's four columns (value, gate, value, gate) un-normalised -> normalised, then the same product
    struct Tab { f32x4 lo, hi; };
    auto tab4 = [&](int i) { Tab t; ln_tab_load(p, nbase + i * 16 + fq * 4, t.lo, t.hi); return t; };
    auto pair_ln = [&](const f32x4& a, const Tab& t, int j) {
        const float mu = mean[j], rs = rstd[j];
        f16x2 o;
        o[0] = (half_t)(((a[0] - mu * t.lo[0]) * rs + t.lo[1]) * fie_gelu((a[1] - mu * t.lo[2]) * rs + t.lo[3]) * 1.0f);
        o[1] = (half_t)(((a[2] - mu * t.hi[0]) * rs + t.hi[1]) * fie_gelu((a[3] - mu * t.hi[2]) * rs + t.hi[3]) * 1.0f);
        unsigned bits;
        __builtin_memcpy(&bits, &o, 4);
        return bits;
    };
    // a value / gate column n is output column n / 2, two bytes each: byte offset n
    static_for([&](auto gc) {                                 // four fragment columns at a time: this lane stores column I0 + fq
        constexpr int I0 = 4 * decltype(gc)::value;
        const int nst = nbase + (I0 + fq) * 16;
        const unsigned co = nst < p.N ? (unsigned)nst : 0xC0000000u;
        if constexpr (LN) {
            const Tab t0 = tab4(I0), t1 = tab4(I0 + 1), t2 = tab4(I0 + 2), t3 = tab4(I0 + 3);
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                unsigned r0 = pair_ln(acc[I0][j], t0, j), r1 = pair_ln(acc[I0 + 1][j], t1, j), r2 = pair_ln(acc[I0 + 2][j], t2, j), r3 = pair_ln(acc[I0 + 3][j], t3, j);
                transpose4(r0, r1, r2, r3);
                const u32x4 v = {r0, r1, r2, r3};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_c, rowoff(mrow + j * 16) + co, 0, 0);
            }
        } else {
            const u32x2 b0 = bias4(I0), b1 = bias4(I0 + 1), b2 = bias4(I0 + 2), b3 = bias4(I0 + 3);
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                unsigned r0 = pair(acc[I0][j], b0), r1 = pair(acc[I0 + 1][j], b1), r2 = pair(acc[I0 + 2][j], b2), r3 = pair(acc[I0 + 3][j], b3);
                transpose4(r0, r1, r2, r3);
                const u32x4 v = {r0, r1, r2, r3};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_c, rowoff(mrow + j * 16) + co, 0, 0);
            }
        }
    }, std::make_integer_sequence<int, FN / 4>{});
    static_for([&](auto ic) {                                 // the columns left over: four row fragments at a time, this lane stores row fragment 4 h + fq
        constexpr int I = FN / 4 * 4 + decltype(ic)::value;
        const int n = nbase + I * 16;
        const unsigned co = n < p.N ? (unsigned)n : 0xC0000000u;
        if constexpr (LN) {
            const Tab t = tab4(I);
#pragma unroll
            for (int h = 0; h < FM / 4; ++h) {
                unsigned r0 = pair_ln(acc[I][4 * h], t, 4 * h), r1 = pair_ln(acc[I][4 * h + 1], t, 4 * h + 1), r2 = pair_ln(acc[I][4 * h + 2], t, 4 * h + 2), r3 = pair_ln(acc[I][4 * h + 3], t, 4 * h + 3);
                transpose4(r0, r1, r2, r3);
                const u32x4 v = {r0, r1, r2, r3};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_c, rowoff(mrow + (4 * h + fq) * 16) + co, 0, 0);
            }
        } else {
            const u32x2 b = bias4(I);
#pragma unroll
            for (int h = 0; h < FM / 4; ++h) {
                unsigned r0 = pair(acc[I][4 * h], b), r1 = pair(acc[I][4 * h + 1], b), r2 = pair(acc[I][4 * h + 2], b), r3 = pair(acc[I][4 * h + 3], b);
                transpose4(r0, r1, r2, r3);
                const u32x4 v = {r0, r1, r2, r3};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_c, rowoff(mrow + (4 * h + fq) * 16) + co, 0, 0);
            }
        }
    }, std::make_integer_sequence<int, FN % 4>{});
}

// PATCH (conv_halo.hip only): the tile is a 16x16 output patch, fragment rows p.frag_ld = OW apart (GemmArgs::frag_ld); every other kernel compiles
// the constant 16 (a runtime stride in here costs the big tiles their register allocation: measured, round 4)
template <int FM, int FN, int WM, int WN, bool BUF = false, bool PATCH = false>
__device__ __forceinline__ void epilogue(const GemmArgs& p, f32x4 (&acc)[FN][FM], int m0, int n0, int wm, int wn, int lane, const EpiPre<FM, FN>* pre = nullptr) {
    const int fr = lane & 15, fq = lane >> 4;
    const bool have_pre = pre != nullptr && pre->on;
    if (p.probe == 4 && p.M > 0) return;                    // timing probe: no epilogue at all (p.M > 0 keeps the accumulators live)
    const bool geglu = p.act == FIE_ACT_GEGLU;
    const int fld = PATCH ? p.frag_ld : 16;                 // rows between consecutive fragments
    const int mrow = m0 + wm * FM * fld + fr, ncol = n0 + wn * WN + fq * 4;
    // clamped coordinates for the loads of the pointer form; BUF: byte offsets, out of range where the lane is outside the matrix
    auto col = [&](int i) { const int n = ncol + i * 16; return n < p.N ? n : p.N - 4; };       // N % 4 == 0
    auto row = [&](int j) { const int m = mrow + j * fld; return m < p.M ? m : p.M - 1; };
    constexpr unsigned kRowOut = 0x80000000u, kColOut = 0xC0000000u;    // any sum with a span <= 1 GiB stays out of range
    auto coff = [&](int i, bool half) { const int n = ncol + i * 16; return n < p.N ? (unsigned)n << (half ? 0 : 1) : kColOut; };   // bytes; half (GEGLU): output column n / 2
    auto roff = [&](int j, int64_t ld) { const int m = mrow + j * fld; return m < p.M ? (unsigned)m * (unsigned)ld * 2u : kRowOut; };
    auto coff_row = [&](int j) {                            // byte offset of output row j in C: scattered for the parity convs of a 2x upsampling
        const int m = mrow + j * fld;
        if (m >= p.M) return kRowOut;
        if (!p.oscat) return (unsigned)m * (unsigned)p.ldc * 2u;
        const int hw = p.OH * p.OW, b = m / hw, rem = m - b * hw, oh = rem / p.OW, ow = rem - oh * p.OW;
        return (unsigned)((b * 2 * p.OH + 2 * oh + p.opy) * (2 * p.OW) + 2 * ow + p.opx) * (unsigned)p.ldc * 2u;
    };
    auto rsrc = [&](const void* ptr, int64_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000); };
    auto as_h4 = [](u32x2 v) { f16x4 h; __builtin_memcpy(&h, &v, 8); return h; };

    if (p.w_scale) {
        const float as = p.a_scale != 0.f ? p.a_scale : 1.f;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            float4 ws = *reinterpret_cast<const float4*>(p.w_scale + col(i));
            ws.x *= as; ws.y *= as; ws.z *= as; ws.w *= as;
#pragma unroll
            for (int j = 0; j < FM; ++j) { acc[i][j][0] *= ws.x; acc[i][j][1] *= ws.y; acc[i][j][2] *= ws.z; acc[i][j][3] *= ws.w; }
        }
    }
    if (p.bias) {
        f16x4 b[FN];
#pragma unroll
        for (int i = 0; i < FN; ++i) b[i] = have_pre ? as_h4(pre->bias[i]) : *reinterpret_cast<const f16x4*>(p.bias + col(i));
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)b[i][r];
    }
    if (p.rowbias) {
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const half_t* rb = p.rowbias + (int64_t)(row(j) / p.rows_per_batch) * p.ld_rowbias;
            f16x4 b[FN];
#pragma unroll
            for (int i = 0; i < FN; ++i) b[i] = *reinterpret_cast<const f16x4*>(rb + col(i));
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)b[i][r];
        }
    }
    auto sweep = [&](auto f) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = f(acc[i][j][r]) * p.scale;
    };
    if (geglu) {                                            // (value, gate) column pairs -> acc[.][.][0..1]
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                acc[i][j][0] = acc[i][j][0] * fie_gelu(acc[i][j][1]) * p.scale;
                acc[i][j][1] = acc[i][j][2] * fie_gelu(acc[i][j][3]) * p.scale;
            }
    } else if (p.act == FIE_ACT_SILU) sweep([](float x) { return fie_silu(x); });
    else if (p.act == FIE_ACT_GELU) sweep([](float x) { return fie_gelu(x); });
    else if (p.act == FIE_ACT_QUICK_GELU) sweep([](float x) { return fie_qgelu(x); });
    else if (p.scale != 1.f) sweep([](float x) { return x; });
    if (p.res) {                                            // never with GEGLU (check_epilogue); may alias C: read before this tile's stores
        const __amdgpu_buffer_rsrc_t rs = rsrc(p.res, BUF ? ((int64_t)(p.M - 1) * p.ldr + p.N) * 2 : 0);
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            f16x4 b[FN];
            if (have_pre) {
#pragma unroll
                for (int i = 0; i < FN; ++i) b[i] = as_h4(pre->res[j][i]);
            } else if constexpr (BUF) {
                const unsigned ro = roff(j, p.ldr);
#pragma unroll
                for (int i = 0; i < FN; ++i) b[i] = as_h4(__builtin_amdgcn_raw_buffer_load_b64(rs, ro + coff(i, false), 0, 0));
            } else {
                const half_t* rr = p.res + (int64_t)row(j) * p.ldr;
#pragma unroll
                for (int i = 0; i < FN; ++i) b[i] = *reinterpret_cast<const f16x4*>(rr + col(i));
            }
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)b[i][r];
            if (FM * FN > 16) __builtin_amdgcn_sched_barrier(0);        // 256x256: no room to keep every row's loads in flight
        }
    }
    if (p.gn_partial) {
        // GroupNorm partial sums of what is about to be stored (the consumer normalises the f16 tensor, so the sums run over the
        // ROUNDED values): a lane's 4 columns lie in one group (cg = 4: are one group); rows reduce over the 16 fr lanes and over the
        // two fragments of a 32-row granule, columns over 1 / 2 / 4 fq lanes.  One lane per (granule, group) stores: every slot has
        // exactly one writer, so the sums are deterministic.  WM % 32 == 0 for every tile.
        static_assert(FM % 2 == 0, "row granules are two fragments");
        const int nch = p.gn_nch ? p.gn_nch : p.gn_rows >> 5;
#pragma unroll
        for (int jj = 0; jj < FM / 2; ++jj) {
            float sum[FN], sq[FN];
#pragma unroll
            for (int i = 0; i < FN; ++i) sum[i] = sq[i] = 0.f;
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                const bool ok = mrow + (2 * jj + dj) * fld < p.M;
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = ok ? (float)(half_t)acc[i][2 * jj + dj][r] : 0.f;
                        sum[i] += x;
                        sq[i] += x * x;
                    }
            }
            const int mg = m0 + (wm * FM + 2 * jj) * fld;   // first row of the granule (wave-uniform)
            const int b = mg / p.gn_rows, rem = mg - b * p.gn_rows;
            // granule id within the image: 32 consecutive rows, or (patch tiles) two 16-pixel runs one image row apart -- any one-to-one numbering
            // of an image's rows / 32 granules serves: the consumer only sums over them
            const int chunk = p.gn_chunk0 + (PATCH ? ((rem / p.frag_ld) >> 1) * (p.frag_ld >> 4) + ((rem % p.frag_ld) >> 4) : rem >> 5);
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                // 16 rows: four DPP adds (xor 1, xor 2, mirror in 8, mirror in 16) leave the row-of-16 total in every lane, VALU only;
                // the four fq totals are then read as scalars from lanes 0 / 16 / 32 / 48
                const float rs = row16_sum(sum[i]), rq = row16_sum(sq[i]);
                float ts[4], tq[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ts[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs), 16 * k));
                    tq[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rq), 16 * k));
                }
                if (p.gn_cg == 16) { ts[0] = (ts[0] + ts[1]) + (ts[2] + ts[3]); tq[0] = (tq[0] + tq[1]) + (tq[2] + tq[3]); }
                else if (p.gn_cg == 8) { ts[0] += ts[1]; tq[0] += tq[1]; ts[1] = ts[2] + ts[3]; tq[1] = tq[2] + tq[3]; }
                const int ngrp = 16 / p.gn_cg;                // groups in this fragment's 16 columns: 1, 2 or 4
                const int nfrag = n0 + wn * WN + i * 16;      // wave-uniform
                if (lane < ngrp && mg < p.M && nfrag + lane * p.gn_cg < p.N) {
                    const float vs = lane == 0 ? ts[0] : lane == 1 ? ts[1] : lane == 2 ? ts[2] : ts[3];
                    const float vq = lane == 0 ? tq[0] : lane == 1 ? tq[1] : lane == 2 ? tq[2] : tq[3];
                    float2* dst = reinterpret_cast<float2*>(p.gn_partial) + ((int64_t)b * nch + chunk) * p.gn_G + nfrag / p.gn_cg + lane;
                    *dst = make_float2(vs, vq);
                }
            }
        }
    }
    if constexpr (BUF) {
        if (p.out_f8) {                                     // e4m3 bytes for an fp8-activation consumer: 4 (GEGLU: 2) bytes per lane and fragment
            const __amdgpu_buffer_rsrc_t rs8 = rsrc(p.C, (int64_t)(p.M - 1) * p.ldc + (geglu ? p.N >> 1 : p.N));
            const float inv = p.out_inv_scale;
            auto q = [&](float x) { return fie_sat448(x * inv); };
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const int m = mrow + j * fld;
                const unsigned ro = m < p.M ? (unsigned)m * (unsigned)p.ldc : kRowOut;
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    const int n = ncol + i * 16;
                    if (geglu) {
                        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q(acc[i][j][0]), q(acc[i][j][1]), 0, false);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pk & 0xffff), rs8, ro + (n < p.N ? (unsigned)(n >> 1) : kColOut), 0, 0);
                    } else {
                        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q(acc[i][j][0]), q(acc[i][j][1]), 0, false);
                        pk = __builtin_amdgcn_cvt_pk_fp8_f32(q(acc[i][j][2]), q(acc[i][j][3]), pk, true);
                        __builtin_amdgcn_raw_buffer_store_b32((unsigned)pk, rs8, ro + (n < p.N ? (unsigned)n : kColOut), 0, 0);
                    }
                }
            }
            return;
        }
    }
    // ---- stores: lane holds C[m = .. + fr][n = .. + fq*4 + (0..3)]
    const __amdgpu_buffer_rsrc_t rs_c = rsrc(p.C, BUF ? ((int64_t)(p.oscat ? 4 * (int64_t)p.M : p.M) - 1) * p.ldc * 2 + (geglu ? p.N >> 1 : p.N) * 2 : 0);
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = mrow + j * fld;
        const unsigned ro = coff_row(j);
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int n = ncol + i * 16;
            if (geglu) {
                f16x2 o;
                o[0] = (half_t)acc[i][j][0]; o[1] = (half_t)acc[i][j][1];
                if constexpr (BUF) {
                    unsigned bits; __builtin_memcpy(&bits, &o, 4);
                    __builtin_amdgcn_raw_buffer_store_b32(bits, rs_c, ro + coff(i, true), 0, 0);
                } else if (m < p.M && n < p.N) {
                    *reinterpret_cast<f16x2*>(p.C + (int64_t)m * p.ldc + (n >> 1)) = o;
                }
            } else {
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (half_t)acc[i][j][r];
                if constexpr (BUF) {
                    u32x2 bits; __builtin_memcpy(&bits, &o, 8);
                    __builtin_amdgcn_raw_buffer_store_b64(bits, rs_c, ro + coff(i, false), 0, 0);   // default cache policy: nt stores +16-24 % per kernel, sc0 sc1 -5 % per kernel but +1.3 % on the whole edit (the consumer reads further away)
                } else if (m < p.M && n < p.N) {
                    *reinterpret_cast<f16x4*>(p.C + (int64_t)m * p.ldc + n) = o;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                  // one row of fragments at a time: converting all of them first costs 2 VGPRs each (256x256: spills)
    }
}

// ---- LDS-DMA helpers (buffer_load_dwordx4 ... offen lds): a wave-uniform descriptor + a per-lane 32-bit byte offset; lanes
// whose offset is >= num_records (im2col padding, rows >= M, the K tail: kOob) read zeros through the descriptor's range check.
constexpr unsigned kOob = 0x80000000u;

__device__ __forceinline__ void bload16(__amdgpu_buffer_rsrc_t rsrc, half_t* lds_dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ---- in-launch split-K reduction (guide: "Projection GEMM at M = 256" item 2, the write-through form).  Every slice block stores its
// accumulators as whole 16-byte lane vectors ([fragment][thread]: 1 KiB per wave instruction) with sc1 (write-through) stores, every wave
// drains its stores, the workgroup meets at a barrier and ONE lane adds to the tile's arrival counter (relaxed, agent scope).  Nobody
// ever waits for another block (no residency assumption, no deadlock with other streams' kernels on the chip): the block whose add
// returns splitk - 1 (mod splitk) is the reducer.  It subtracts splitk from the counter (counters start zeroed: fie_splitk_workspace; every
// launch leaves them zero), makes
// one agent-scope acquire and reads ALL slices (its own included: no per-slice branch) in slice order with sc1 loads, so the sum
// does not depend on which block came last.  Returns true in the reducer, whose accumulators then hold the full sums.
template <int FM, int FN, int NW>
__device__ __forceinline__ bool splitk_reduce(const GemmArgs& p, f32x4 (&acc)[FN][FM], int tile, int slice, int tid, void* lds) {
    constexpr int kTile = FM * FN * NW * 64 * 4;                  // floats per slab = BM * BN
    const int S = p.splitk;
    float* base = p.sk_slabs + (size_t)tile * S * kTile;
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)slice * kTile, 0, kTile * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, (unsigned)((i * FM + j) * NW * 64 + tid) * 16u, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // EVERY storing wave drains before the workgroup signals
    __syncthreads();
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(lds);   // the ring is drained: reuse its first word (one LDS object only)
    if (tid == 0) *flag = __hip_atomic_fetch_add(p.sk_tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // Every S-th arrival reduces, and it takes its S arrivals back OFF the counter with one atomic subtract (round 4; was: "== S - 1, then
    // store 0").  A subtract of S never changes the counter mod S, so arrival k always sees k mod S whatever interleaves with it, and the
    // counter returns to zero after any whole number of launches: a counter can no longer be left stuck by a caller that breaks the
    // one-launch-in-flight-per-workspace rule (include/fie.h, "split-K workspace") -- the sums of THOSE overlapping launches are still
    // undefined, but every later launch on the workspace is sound again.  (The store-0 form left the counter at 1 forever after two
    // interleaved copies of one launch: GPUTEST r03.)
    if (*flag % (unsigned)S != (unsigned)(S - 1)) return false;
    if (tid == 0) {
        (void)__hip_atomic_fetch_sub(p.sk_tickets + tile, (unsigned)S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)s * kTile, 0, kTile * 4, 0x00020000);
        u32x4 v[FN][FM];
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) v[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((i * FM + j) * NW * 64 + tid) * 16u, 0, 16);
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] += __builtin_bit_cast(f32x4, v[i][j]);
    }
    return true;
}

// XCD-aware bijective tile remap: consecutive tile ids run on one XCD (blocks b and b + 8 share an XCD's L2)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

}  // namespace fie_gemm

// gemm8.hip: the 256x256 phased kernel.  conv != 0: im2col view with Cin % 64 == 0.  Shapes must satisfy the LDS-DMA
// eligibility rule of gemm_conv.hip (operands < 2 GiB, K1 == K or K1 % 64 == 0).
int fie_launch_gemm8(fie_ctx* ctx, const fie_gemm::GemmArgs& a, int conv, int split);
int fie_gemm8_init(void);          // per-device function attributes (dynamic LDS size); called from fie_ctx_create
int fie_gemm_init(void);           // same for the kernels of gemm_conv.hip
// conv_halo.hip: the halo-resident 3x3 conv (tile code 71; 73 = with cycle stamps): a block = a 16x16 output patch x 128 channels
int fie_launch_conv_halo(fie_ctx* ctx, fie_gemm::GemmArgs& a, int stamped);
bool fie_conv_halo_ok(const fie_gemm::GemmArgs& a);
// conv_thin.hip: 3x3 convs with at most 16 output channels (tile code 77): direct global -> VGPR operands, a wave = a 16-pixel strip x all channels
int fie_launch_conv_thin(fie_ctx* ctx, fie_gemm::GemmArgs& a);
bool fie_conv_thin_ok(const fie_gemm::GemmArgs& a);
bool fie_conv_halo_gna_ok(const fie_gemm::GemmArgs& a);
int fie_conv_halo_init(void);
// gemm_w8.hip: fp8-weight ring kernels; code 62 = 256x128 (8 waves), 42 = 128x64, 43 = 64x64
int fie_launch_gemm_w8(fie_ctx* ctx, const fie_gemm::GemmArgs& a, int conv, int code);
int fie_gemm_w8_init(void);
// gemm_x8.hip: e4m3 activations x e4m3 weights on the block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales), GEMM view only
int fie_launch_gemm_x8(fie_ctx* ctx, const fie_gemm::GemmArgs& a, int code, int conv);
int fie_gemm_x8_init(void);
