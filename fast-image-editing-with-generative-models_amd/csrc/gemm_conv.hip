// K1/K2: fp16 MFMA GEMM and 3x3 implicit-GEMM convolution for gfx950 (include/fie.h: fie_gemm_f16, fie_conv3x3_nhwc_f16):
// the launch table and two of the three kernel families (the third, the 256x256 phased kernel, lives in gemm8.hip; shared
// pieces in gemm_common.h).
//
//   gemm_kernel<BM,BN,MODE>          register-staged double buffer, 4 waves.  Serves EVERY shape (generic im2col gather with
//                                    per-lane zero fill): the fallback for operands the LDS-DMA kernels cannot address
//                                    (Cin % 64 != 0: conv_in / ControlNet conditioning embedding; >= 2 GiB operands).
//   gemm3_kernel<BM,BN,ST,MODE,NW>   ST-stage LDS ring filled by `buffer_load_dwordx4 ... offen lds`, counted vmcnt + ONE raw
//                                    s_barrier per K-step, no per-lane address arithmetic in the K loop.  Tiles 256x128 (8 waves),
//                                    128x128 (8 waves), 128x64 and 64x64 (4 waves): the grids that cannot give every CU a
//                                    256x256 tile.
//   gemm8_kernel<MODE> (gemm8.hip)   256x256, four phases per K-tile, two wave groups one barrier apart.
//
// Unselected experiments of round 1 (zero-page global_load_lds ring, ping-pong, four-phase, producer waves, halo-reuse conv,
// 160-wide tiles, column split): measured no better, removed (git history has them; the numbers are in profiles/r01_microbench.md).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "gemm_common.h"

using namespace fie_gemm;

namespace {

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int RA = BM / 32, RW = BN / 32;       // 16-byte chunks per thread per K-step
    __shared__ __attribute__((aligned(16))) half_t smem[2 * (BM + BN) * BK];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);     // consecutive ids on one XCD share the activation rows (n fastest)
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

// wait until at most `later` K-steps' worth of loads (NP per step, issued after the stage about to be consumed) are in flight
template <int NP, int ST>
__device__ __forceinline__ void wait_stage(int later) {
    if (ST >= 3 && later >= 1) wait_vm_barrier<(ST >= 3 ? 1 : 0) * NP>();
    else wait_vm_barrier<0>();
}

// LDS fragment reads as asm: asynchronous, valid after an explicit lgkmcnt wait (prefetching ring kernels only)
template <int OFF>
__device__ __forceinline__ f16x8 lds_read16_async(unsigned addr) {
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N, int... I>
__device__ __forceinline__ void read_frags(f16x8 (&f)[N], unsigned addr, std::integer_sequence<int, I...>) {
    ((f[I] = lds_read16_async<I * 16 * BK * 2>(addr)), ...);
}

// (Round 3 negative result: issuing the K-step's DMA pieces one at a time BETWEEN the MFMAs of the 4-wave tiles, instead of as one burst in
// front of them, is 15-30 % slower on every hot-path GEMM -- profiles/r03_interleaved_dma_issue_negative.log -- and was removed again.)
// ALT (8-wave tiles, GEMM view): the two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) take TURNS issuing a K-step's whole LDS-DMA refill.
// A K-step of these kernels is a serial sum -- DMA issue (the waves sit in the 64 B/clk load path) + fragment reads + MFMAs + barrier -- because every
// wave does the same thing at the same time; with turns, the group that issues nothing goes straight to its reads and MFMAs while the other one feeds the
// load path, and in the next K-step they swap.
// LEAN: 0 = the full epilogue, 1 = the lean one (plain Linear), 2 = LayerNorm folded in (GemmArgs::ln_tab: row sums from the activation fragments, lean LN epilogues)
template <int BM, int BN, int ST, int MODE, int NW, bool PF = false, bool STAMP = false, bool ALT = false, int LEAN = 0>    // MODE 0 = GEMM, 2 = conv with Cin % 64 == 0
__global__ __launch_bounds__(NW * 64) void gemm3_kernel(GemmArgs p) {
    constexpr bool LNF = LEAN == 2;
    static_assert(!LNF || (MODE == 0 && !STAMP), "LayerNorm fold: GEMM view only");
    static_assert(!ALT || (NW == 8 && MODE == 0 && !PF && !STAMP), "ALT: 8-wave GEMM-view tiles without the fragment prefetch");
    // waves as 2 (rows) x NW/2 (columns) wherever that leaves whole 16-column fragments; otherwise (128x80) all NW waves stacked along the rows.
    // LayerNorm fold (small tiles): stacked along the rows too -- every activation row then belongs to ONE wave, which takes its statistics (ln_take below)
    constexpr int WGN = (LEAN == 2 && BM * BN / (NW * 256) <= 16 && (BM / NW) % 16 == 0) ? 1 : (BN / (NW / 2)) % 16 == 0 ? NW / 2 : 1, WGM = NW / WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int NPW = BN / 8;                            // weight pieces (8 rows x 128 B) per K-step
    constexpr int RA = BM / (8 * NW), RW = (NPW + NW - 1) / NW;
    constexpr int NP = RA + RW;                            // EVERY wave issues NP pieces per K-step (the counted waits rely on it): see issue_w for ragged NPW
    constexpr int STAGE = (BM + BN) * BK;
    extern __shared__ __attribute__((aligned(16))) half_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WGM, wn = wave / WGM;

    // split-K: consecutive (remapped) block ids = the slices of one tile, so they share an XCD's L2 for the tile's operands and slabs
    const int bid_all = xcd_remap(blockIdx.x, gridDim.x);
    const int nsplit = p.splitk > 1 ? p.splitk : 1;
    const int tile_all = bid_all / nsplit, slice = bid_all - tile_all * nsplit;
    const int bid = take_parity(p, tile_all);
    const int m0 = (p.order ? bid % p.nbm : bid / p.nbn) * BM;
    const int n0 = (p.order ? bid / p.nbm : bid % p.nbn) * BN;
    const int lr = lane >> 3;
    const int c8 = (lane & 7) ^ lr;

    const int live = p.probe == 1 ? 0 : 1;                 // timing probes, see GemmArgs::probe
    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)p.a1_bytes * live, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, (int)p.a2_bytes * live, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, (int)p.w_bytes * live, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A3, 0, (int)p.a3_bytes * live, 0x00020000);
    const int lm0 = p.probe == 2 ? 0 : m0, ln0 = p.probe == 2 ? 0 : n0;

    // per-lane byte offsets
    unsigned a_off1[RA], a_off2[RA];       // GEMM: row offsets into A1 / A2;  conv: a_off1 = offset for the current tap, a_off2 = row m of the side input A2
    unsigned a_off3[RA];                   // conv: row m of the second side input A3
    int a_ih[RA], a_iw[RA];
    unsigned a_img[RA];
    bool a_ok[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = lm0 + (wave + NW * i) * 8 + lr;
        a_ok[i] = m < p.M;
        if (MODE == 2) {
            const int hw = p.OH * p.OW;
            const int b = m / hw, rem = m - b * hw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_ih[i] = oh * p.stride - p.pt;
            a_iw[i] = ow * p.stride - p.pl;
            a_img[i] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.Cin * 2u;
            a_off1[i] = kOob;
            a_off2[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda2 * 2u + c8 * 16u : kOob;
            a_off3[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda3 * 2u + c8 * 16u : kOob;
        } else {
            a_off1[i] = a_off2[i] = a_off3[i] = a_img[i] = 0;          // GEMM: a_base1 / a_base2 below
            a_ih[i] = a_iw[i] = 0;
        }
    }
    // GEMM view: a lane's row groups are NW * 8 rows apart, so ONE per-lane offset (+ a uniform stride per group) serves all of them.
    // Rows >= M need no mask: their offset m * lda * 2 is >= the descriptor's extent ((M - 1) * lda + K1) * 2 (lda >= K1), so the range
    // check returns zeros (and (M + 255) * lda * 2 stays far below 2^32 for operands < 2 GiB).
    const unsigned a_base1 = (unsigned)(lm0 + wave * 8 + lr) * (unsigned)p.lda1 * 2u + c8 * 16u;
    const unsigned a_base2 = (unsigned)(lm0 + wave * 8 + lr) * (unsigned)p.lda2 * 2u + c8 * 16u;
    const unsigned a_step1 = (unsigned)(NW * 8) * (unsigned)p.lda1 * 2u, a_step2 = (unsigned)(NW * 8) * (unsigned)p.lda2 * 2u;
    const unsigned w_base = (unsigned)(ln0 + wave * 8 + lr) * (unsigned)p.ldw * 2u + c8 * 16u;       // weights are padded to whole tiles: no tail
    const unsigned w_step = (unsigned)(NW * 8) * (unsigned)p.ldw * 2u;

    const int csteps = MODE == 2 ? p.Cin / BK : 1;
    const int k1_steps = p.K1 / BK;            // GEMM: K-steps served by A1 (K1 % 64 == 0 unless K1 == K)
    const int nk_all = (p.K + BK - 1) / BK;
    const int kbeg = (int)((int64_t)nk_all * slice / nsplit);          // this block's K-steps: [kbeg, kbeg + nk); all of them without split-K
    const int nk = (int)((int64_t)nk_all * (slice + 1) / nsplit) - kbeg;
    const bool ktail = (p.K % BK) != 0;
    int ftap = kbeg / csteps, cs = kbeg - ftap * csteps;              // conv: tap and channel step of the next K-step to issue
    bool tap_fresh = true;                                            // a slice may start in the middle of a tap

    const int ntaps = p.taps2 ? 4 : 9;
    auto issue = [&](int kt, int stage) {
        half_t* sa = smem + stage * STAGE;
        if (MODE == 2 && ftap >= ntaps) {                   // past the taps: the 1x1 side inputs, row m itself
            const int e = kt - ntaps * csteps, x2 = p.C2x / BK;
            if (e < x2) {
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a2, sa + (wave + NW * i) * 512, a_off2[i], (unsigned)e * (BK * 2));
            } else {
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a3, sa + (wave + NW * i) * 512, a_off3[i], (unsigned)(e - x2) * (BK * 2));
            }
        } else if (MODE == 2) {
            if (cs == 0 || tap_fresh) {
                tap_fresh = false;
                const int ky = p.taps2 ? ftap >> 1 : (ftap * 11) >> 5, kx = p.taps2 ? ftap & 1 : ftap - 3 * ky;
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
            if (ktail && kt == nk_all - 1) {           // last, partial K-step: columns >= K read as zero
                const bool in_k = kt * BK + c8 * 8 < p.K;
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a1, sa + (wave + NW * i) * 512, in_k ? a_base1 + (unsigned)i * a_step1 : kOob, (unsigned)kt * (BK * 2));
            } else if (kt < k1_steps || k1_steps == 0) {
                const unsigned so = (unsigned)kt * (BK * 2);
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a1, sa + (wave + NW * i) * 512, a_base1 + (unsigned)i * a_step1, so);
            } else {
                const unsigned so = (unsigned)(kt - k1_steps) * (BK * 2);
#pragma unroll
                for (int i = 0; i < RA; ++i) bload16(rs_a2, sa + (wave + NW * i) * 512, a_base2 + (unsigned)i * a_step2, so);
            }
        }
    };
    auto issue_w = [&](int kt, int stage) {
        half_t* sw = smem + stage * STAGE + BM * BK;
        const unsigned sow = (unsigned)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            // ragged piece count (128x80: 10 pieces over 4 waves): a wave without an i-th piece re-issues its first one (same data, same place)
            const int ii = (NPW % NW == 0 || wave + NW * i < NPW) ? i : 0;
            bload16(rs_w, sw + (wave + NW * ii) * 512, w_base + (unsigned)ii * w_step, sow);
        }
    };

    // ALT: this wave's share when its group refills a whole stage alone: pieces lw + 4 i of the 2 RA activation and 2 RW weight pieces
    const int grp = wave >> 2, lw = wave & 3;
    const unsigned alt_a1 = (unsigned)(lm0 + lw * 8 + lr) * (unsigned)p.lda1 * 2u + c8 * 16u, alt_a2 = (unsigned)(lm0 + lw * 8 + lr) * (unsigned)p.lda2 * 2u + c8 * 16u;
    const unsigned alt_sa1 = 32u * (unsigned)p.lda1 * 2u, alt_sa2 = 32u * (unsigned)p.lda2 * 2u;
    const unsigned alt_w = (unsigned)(ln0 + lw * 8 + lr) * (unsigned)p.ldw * 2u + c8 * 16u, alt_sw = 32u * (unsigned)p.ldw * 2u;
    auto issue_alt = [&](int kt, int stage) {
        half_t* sa = smem + stage * STAGE;
        half_t* sw = sa + BM * BK;
        if (ktail && kt == nk_all - 1) {
            const bool in_k = kt * BK + c8 * 8 < p.K;
#pragma unroll
            for (int i = 0; i < 2 * RA; ++i) bload16(rs_a1, sa + (lw + 4 * i) * 512, in_k ? alt_a1 + (unsigned)i * alt_sa1 : kOob, (unsigned)kt * (BK * 2));
        } else if (kt < k1_steps || k1_steps == 0) {
            const unsigned so = (unsigned)kt * (BK * 2);
#pragma unroll
            for (int i = 0; i < 2 * RA; ++i) bload16(rs_a1, sa + (lw + 4 * i) * 512, alt_a1 + (unsigned)i * alt_sa1, so);
        } else {
            const unsigned so = (unsigned)(kt - k1_steps) * (BK * 2);
#pragma unroll
            for (int i = 0; i < 2 * RA; ++i) bload16(rs_a2, sa + (lw + 4 * i) * 512, alt_a2 + (unsigned)i * alt_sa2, so);
        }
        const unsigned sow = (unsigned)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < 2 * RW; ++i) bload16(rs_w, sw + (lw + 4 * i) * 512, alt_w + (unsigned)i * alt_sw, sow);
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // bias row + residual tile go out ahead of the ring's first stage and are complete (loads return in order) by the first counted wait
    EpiPre<FM, FN> pre;
    pre.on = false;
    LnTab<(LNF && FM * FN <= 16) ? FN : 1> ln_pre;
    if constexpr (FM * FN <= 16) {
        if constexpr (LNF) {
            // the (colsum, bias') table is loaded AFTER the K loop (below), not here with the other epilogue operands: see the note there
        } else if (LEAN || (nsplit == 1 && p.epi_prefetch)) epilogue_prefetch<FM, FN, WM, WN>(p, pre, m0, n0, wm, wn, lane);
    }
    // LNF: every wave sums the rows of ITS OWN activation fragments, all of them, straight-line: v_dot2c (x . 1, x . x), 8 per fragment and half K-step.
    // (Round 4, profiles/r04_ln_fold_row_sum_variants.log: dealing the fragments to the wave columns that share them -- a wave-uniform `if` per fragment
    // inside the K loop -- cost 5-8 us per launch whatever did the sums, VALU or two extra MFMAs per fragment (Gram diagonal + ones row): the branches
    // split the K loop into basic blocks and every block boundary drains the LDS / MFMA pipeline.  Hence the row-stacked wave layout above: each row
    // fragment belongs to exactly one wave, nothing is summed twice and nothing is traded through LDS.)
    constexpr int LJ = LNF ? FM : 1;
    float ln_s[LJ], ln_q[LJ];
#pragma unroll
    for (int t = 0; t < LJ; ++t) ln_s[t] = ln_q[t] = 0.f;
    // The sums ride BETWEEN the MFMAs, a few per MFMA (issued behind the whole MFMA block they simply add their issue time: +11 us on the FF1 tile):
    // a fragment is four register pairs = four units of two v_dot2c; ln_slice<I, NI, J0, CNT>(fa) issues the units of fragments fa[0 .. CNT) (row fragments
    // J0 ..) that fall to column iteration I of NI
    auto ln_unit = [&](auto jc, auto tc, const f16x8& f) {
        constexpr int j = decltype(jc)::value, t = decltype(tc)::value;
        const f16x2 one = {(half_t)1.f, (half_t)1.f};
        const f16x2 v = {f[2 * t], f[2 * t + 1]};
        ln_s[j] = __builtin_amdgcn_fdot2(v, one, ln_s[j], false);
        ln_q[j] = __builtin_amdgcn_fdot2(v, v, ln_q[j], false);
    };
    auto ln_slice = [&](auto ic, auto nic, auto j0c, auto cntc, const f16x8* fa) {
        if constexpr (LNF) {
            constexpr int I = decltype(ic)::value, NI = decltype(nic)::value, J0 = decltype(j0c)::value, U = 4 * decltype(cntc)::value;
            constexpr int u0 = I * U / NI, u1 = (I + 1) * U / NI;
            static_for([&](auto uc) {
                constexpr int u = u0 + decltype(uc)::value;
                ln_unit(std::integral_constant<int, J0 + u / 4>{}, std::integral_constant<int, u % 4>{}, fa[u / 4]);
            }, std::make_integer_sequence<int, u1 - u0>{});
        }
    };
    auto ln_pin = [&]() {
        if constexpr (LNF) {
#pragma unroll
            for (int t = 0; t < LJ; ++t) asm volatile("" : "+v"(ln_s[t]), "+v"(ln_q[t]));
        }
    };
    using c0_t = std::integral_constant<int, 0>;

#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s < nk) { issue(kbeg + s, s); issue_w(kbeg + s, s); }

    const int fr = lane & 15, fq = lane >> 4;
    int stage = 0, fill = ST - 1;
    auto frags = [&](f16x8* fw, f16x8* fa, const half_t* sa, int kk) {
        const half_t* sw = sa + BM * BK;
#pragma unroll
        for (int i = 0; i < FN; ++i) fw[i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * WN + i * 16 + fr, kk * 4 + fq));
#pragma unroll
        for (int j = 0; j < FM; ++j) fa[j] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * WM + j * 16 + fr, kk * 4 + fq));
    };
    auto mfmas = [&](const f16x8* fw, const f16x8* fa) {
        if (NW == 8) __builtin_amdgcn_s_setprio(1);
        static_for([&](auto ic) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
            ln_slice(ic, std::integral_constant<int, FN>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, FM>{}, fa);
        }, std::make_integer_sequence<int, FN>{});
        if (NW == 8) __builtin_amdgcn_s_setprio(0);
    };
    // STAMP instances (tile codes 97 / 98, fie_debug_gemm_stamps): per-wave cycle sums of the K-loop segments, s_memtime deltas
    unsigned seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;      // [7] prologue, [8] epilogue, [9] / [10] entry / exit time
    auto stamp = [&](int i) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
            seg[i] += t - tprev;
            tprev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (STAMP) tprev = seg[9] = (unsigned)__builtin_amdgcn_s_memtime();
    if constexpr (PF) {
        // Fragment reads run one half K-step ahead of the MFMAs that use them: the block's waves leave each barrier together, so
        // without this every wave reads LDS at the same time (matrix cores idle) and then all issue MFMAs (LDS idle).
        // The reads are inline asm with explicit counted waits: the compiler's own wait insertion falls back to lgkmcnt(0) in
        // this loop (scalar loads pending at loop entry share the counter), which serialises exactly what this path overlaps.
        static_assert(FM + FN <= 15, "lgkmcnt is a 4-bit counter");
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) half_t*)smem;
        unsigned a_rd[2], w_rd[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            a_rd[kk] = lds0 + 2u * (unsigned)lds_off(wm * WM + fr, kk * 4 + fq);
            w_rd[kk] = lds0 + 2u * (unsigned)(BM * BK + lds_off(wn * WN + fr, kk * 4 + fq));
        }
        f16x8 fw0[FN], fa0[FM], fw1[FN], fa1[FM];
        auto reads = [&](f16x8 (&fw)[FN], f16x8 (&fa)[FM], int st, int kk) {
            const unsigned so = (unsigned)st * (STAGE * 2);
            read_frags(fw, w_rd[kk] + so, std::make_integer_sequence<int, FN>{});
            read_frags(fa, a_rd[kk] + so, std::make_integer_sequence<int, FM>{});
        };
        auto landed = [&](f16x8 (&fw)[FN], f16x8 (&fa)[FM]) {     // MFMAs on these fragments stay behind the wait in front
#pragma unroll
            for (int i = 0; i < FN; ++i) asm volatile("" : "+v"(fw[i]));
#pragma unroll
            for (int j = 0; j < FM; ++j) asm volatile("" : "+v"(fa[j]));
        };
        wait_stage<NP, ST>(min(ST - 2, nk - 1));
        reads(fw0, fa0, 0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            stamp(kt == 0 ? 7 : 0);                        // whole K-steps only: a stamp drains lgkmcnt
            const bool more = kt + ST - 1 < nk && p.probe != 3;
            if (more) { issue(kbeg + kt + ST - 1, fill); if (NW != 8) issue_w(kbeg + kt + ST - 1, fill); }
            // LNF: every use of a fragment set stays on its side of the asm reads that refill it.  The row sums are plain VALU uses of fa: the scheduler
            // sank them below the next `reads`, the old fragments then outlived the statement, the new ones got other registers and a COPY at the loop's
            // back edge -- a v_mov of asm-loaded registers ahead of their lgkmcnt wait (cdna_hip_programming.md, "What hipcc does not do" 1): stale B
            // operands whenever LDS was slow (round 4: a whole edit differed between replays of one graph).  ln_pin: an asm statement that takes the sums, so
            // the v_dot2c's that feed them stand in front of it, and asm volatile statements keep their order.  Audit: no v_mov_b64 of fragments in the loop
            ln_pin();
            reads(fw1, fa1, stage, 1);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FM + FN) : "memory");
            landed(fw0, fa0);
            mfmas(fw0, fa0);
            if (more && NW == 8) issue_w(kbeg + kt + ST - 1, fill);
            stage = stage + 1 == ST ? 0 : stage + 1;
            fill = fill + 1 == ST ? 0 : fill + 1;
            // every read of the stage consumed in this K-step has returned before any wave may refill it (next iteration)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            landed(fw1, fa1);
            if (kt + 1 < nk) {
                wait_stage<NP, ST>(min(kt + ST - 1, nk - 1) - (kt + 1));
                ln_pin();
                reads(fw0, fa0, stage, 0);
            }
            mfmas(fw1, fa1);
        }
    } else
    for (int kt = 0; kt < nk; ++kt) {
        const int later = min(kt + ST - 2, nk - 1) - kt;
        if constexpr (ALT && ST >= 3) {
            // stage kt was issued (whole) by group kt & 1 two K-steps ago, stage kt + 1 by the other group one K-step ago: the group that owns
            // stage kt drains, the other one may keep its 2 NP pieces of stage kt + 1 in flight (the prologue is issued by all waves: kt < ST - 1 drains)
            if (later >= 1 && kt >= ST - 1 && (kt & 1) != grp) wait_vm_barrier<2 * NP>();
            else wait_vm_barrier<0>();
        } else
        wait_stage<NP, ST>(later);
        stamp(kt == 0 ? 7 : 0);                            // 0: MFMA drain + barrier wait
        const bool more = kt + ST - 1 < nk;
        const half_t* sa = smem + stage * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // the refill of the freed stage is issued in two halves, one in front of each MFMA group, so that no wave
            // spends a whole K-step's worth of LDS-DMA issue slots before its first MFMA
            // (measured: pays for the 8-wave blocks, costs 10 % on the 4-wave ones, which keep one burst per K-step)
            if (more && p.probe != 3) {                // probe 3: no DMA issued inside the K loop at all (stale LDS data)
                if constexpr (ALT) {
                    if (kk == 0 && ((kt + ST - 1) & 1) == grp) issue_alt(kbeg + kt + ST - 1, fill);      // stage s is refilled by group s & 1
                } else {
                    if (kk == 0) { issue(kbeg + kt + ST - 1, fill); if (NW != 8) issue_w(kbeg + kt + ST - 1, fill); }
                    else if (NW == 8) issue_w(kbeg + kt + ST - 1, fill);
                }
            }
            stamp(1 + 3 * kk);                             // 1 / 4: DMA issue
            if constexpr (FM * FN > 16) {
                // 256x256 / 256x320: the activation fragments in two halves (13 fragments + 160 accumulators would not fit 256 VGPRs)
                constexpr int HM = FM / 2;
                const half_t* sw = sa + BM * BK;
                f16x8 fw[FN];
#pragma unroll
                for (int i = 0; i < FN; ++i) fw[i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * WN + i * 16 + fr, kk * 4 + fq));
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    f16x8 fa[HM];
#pragma unroll
                    for (int j = 0; j < HM; ++j) fa[j] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * WM + (jh * HM + j) * 16 + fr, kk * 4 + fq));
                    __builtin_amdgcn_s_setprio(1);
                    static_for([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
#pragma unroll
                        for (int j = 0; j < HM; ++j) acc[i][jh * HM + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][jh * HM + j], 0, 0, 0);
                        if (jh == 0) ln_slice(ic, std::integral_constant<int, FN>{}, c0_t{}, std::integral_constant<int, HM>{}, fa);
                        else ln_slice(ic, std::integral_constant<int, FN>{}, std::integral_constant<int, HM>{}, std::integral_constant<int, HM>{}, fa);
                    }, std::make_integer_sequence<int, FN>{});
                    __builtin_amdgcn_s_setprio(0);
                }
            } else {
                f16x8 fw[FN], fa[FM];
                frags(fw, fa, sa, kk);
                if constexpr (STAMP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                stamp(2 + 3 * kk);                         // 2 / 5: fragment reads issued and landed
                mfmas(fw, fa);
                stamp(3 + 3 * kk);                         // 3 / 6: 16 MFMAs issued
            }
        }
        stage = stage + 1 == ST ? 0 : stage + 1;
        fill = fill + 1 == ST ? 0 : fill + 1;
    }
    stamp(11);                                             // last MFMA group issued
    float ln_mean[LNF ? FM : 1], ln_rstd[LNF ? FM : 1];
    if constexpr (LNF && FM * FN <= 16) {
        // loaded behind the K loop: 16-64 VGPRs that nothing in the loop needs
#pragma unroll
        for (int i = 0; i < FN; ++i) ln_tab_load(p, n0 + wn * WN + i * 16 + (lane >> 4) * 4, ln_pre.lo[i], ln_pre.hi[i]);
    }
    if constexpr (LNF) {
        // a lane holds the sums over its own k-slices (fq) of row fr, the row it owns in the epilogue: total over the four fq lane rows
        const float inv_k = 1.f / (float)p.K;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            float sx = ln_s[j], sq = ln_q[j];
            sx += __shfl_xor(sx, 16); sx += __shfl_xor(sx, 32);
            sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
            const float mu = sx * inv_k;
            ln_mean[j] = mu;
            ln_rstd[j] = rsqrtf(fmaxf(sq * inv_k - mu * mu, 0.f) + p.ln_eps);
        }
    }
    if constexpr (FM * FN <= 16 && !LEAN) {                // split-K: only the block that draws the tile's last ticket goes on, with the summed slices
        if (nsplit > 1 && !splitk_reduce<FM, FN, NW>(p, acc, tile_all, slice, tid, smem)) return;
    }
    if constexpr (LNF && FM * FN > 16) {
        epilogue_geglu_lean<FM, FN, WM, WN, true>(p, acc, m0, n0, wm, wn, lane, ln_mean, ln_rstd);      // host: GEGLU only on the big tile
    } else if constexpr (LNF) {
        epilogue_lean_ln<FM, FN, WM, WN>(p, acc, m0, n0, wm, wn, lane, ln_pre, ln_mean, ln_rstd);
    } else
    if constexpr (FM * FN > 16) {                          // 256x256 / 256x320: column chunks of two fragments (register pressure, see gemm8.hip)
        if (p.act == FIE_ACT_GEGLU && p.epi_prefetch && !p.rowbias && !p.res && p.scale == 1.f && !p.gn_partial && !p.out_f8 && !p.w_scale && !p.oscat && p.probe == 0) {
            epilogue_geglu_lean<FM, FN, WM, WN>(p, acc, m0, n0, wm, wn, lane);         // the FF1 projection: compact code instead of three generic epilogues
        } else {
        // column chunks of two fragments, the chunk index a COMPILE-TIME constant: as a `#pragma unroll` loop the two huge inlined bodies exceed the
        // pragma-unroll threshold as soon as the epilogue grows by a few instructions, the loop stays rolled, acc[2 * c] becomes a runtime index and the
        // whole accumulator array moves to scratch -- K loop included (round 4: FF1 on tile 64 went from 55 to 87 us that way)
        static_for([&](auto cc) {
            constexpr int C = decltype(cc)::value;
            epilogue<FM, 2, WM, WN, true>(p, reinterpret_cast<f32x4(&)[2][FM]>(acc[2 * C]), m0, n0 + 32 * C, wm, wn, lane);
        }, std::make_integer_sequence<int, FN / 2>{});
        if constexpr (FN & 1)
            epilogue<FM, 1, WM, WN, true>(p, reinterpret_cast<f32x4(&)[1][FM]>(acc[FN - 1]), m0, n0 + 16 * (FN - 1), wm, wn, lane);
        }
    } else {
        if constexpr (LEAN == 1) epilogue_lean<FM, FN, WM, WN>(p, acc, m0, n0, wm, wn, lane, pre);
        else epilogue<FM, FN, WM, WN, true>(p, acc, m0, n0, wm, wn, lane, &pre);
    }
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile's stores have left the wave
        stamp(8);
        seg[10] = tprev;
        if (p.stamps && lane == 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) p.stamps[((size_t)bid * NW + wave) * 12 + i] = seg[i];
        }
    }
}

template <int BM, int BN, int ST, int NW>
constexpr int ring_lds() { return ST * (BM + BN) * BK * (int)sizeof(half_t); }

// tiles with <= 16 accumulator fragments per lane have a LEAN instantiation for the GEMM view (gemm_common.h: epilogue_lean)
template <int BM, int BN, int MODE, int NW, bool STAMP>
constexpr bool has_lean() { return MODE == 0 && !STAMP && BM * BN / (NW * 64) <= 64; }

template <int BM, int BN, int ST, int MODE, int NW, bool PF = false, bool STAMP = false, bool ALT = false>
hipError_t ring_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<BM, BN, ST, MODE, NW, PF, STAMP, ALT, 0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, ring_lds<BM, BN, ST, NW>());
    if constexpr (has_lean<BM, BN, MODE, NW, STAMP>()) {
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<BM, BN, ST, MODE, NW, PF, STAMP, ALT, 1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, ring_lds<BM, BN, ST, NW>());
    }
    return e;
}

template <int BM, int BN, int ST, int MODE, int NW, bool PF = false, bool STAMP = false, bool ALT = false>
void launch_ring(fie_ctx* ctx, const GemmArgs& a, dim3 grid) {
    constexpr int lds = ring_lds<BM, BN, ST, NW>();
    if constexpr (has_lean<BM, BN, MODE, NW, STAMP>()) {
        // a plain Linear (optional bias, optional residual, f16 out): the kernel with the lean epilogue
        if (a.epi_prefetch && a.splitk <= 1 && !a.rowbias && a.act == FIE_ACT_NONE && a.scale == 1.f && !a.gn_partial && !a.out_f8 && !a.w_scale && !a.oscat &&
            a.probe == 0) {
            fie_launch(ctx, (gemm3_kernel<BM, BN, ST, MODE, NW, PF, STAMP, ALT, 1>), grid, dim3(NW * 64), lds, a);
            return;
        }
    }
    fie_launch(ctx, (gemm3_kernel<BM, BN, ST, MODE, NW, PF, STAMP, ALT, 0>), grid, dim3(NW * 64), lds, a);
}

// LayerNorm folded into the GEMM (GemmArgs::ln_tab, LEAN == 2): built for the four tiles the transformer blocks' LN consumers run on
template <int BM, int BN, int ST, int NW, bool PF = false, bool ALT = false>
void launch_ring_ln(fie_ctx* ctx, const GemmArgs& a, dim3 grid) {
    fie_launch(ctx, (gemm3_kernel<BM, BN, ST, 0, NW, PF, false, ALT, 2>), grid, dim3(NW * 64), (ring_lds<BM, BN, ST, NW>()), a);
}
template <int BM, int BN, int ST, int NW, bool PF = false, bool ALT = false>
hipError_t ring_attr_ln() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<BM, BN, ST, 0, NW, PF, false, ALT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               ring_lds<BM, BN, ST, NW>());
}
// (128x80, tile 48, was built too and is NOT offered: its instantiation returned wrong values in 16-row x 1-column spots on a loaded chip, in every form
// tried, while 42 / 96 / 64 never did: profiles/r04_ln_fold_tile48_anomaly.md; tests/test_ops_gpu.py screens the three that ship)
constexpr bool is_ln_code(int code) { return code == 42 || code == 96 || code == 64; }

// ---- tile codes (also the values fie_debug_force_tile / fie_debug_tile_override take)
//   1 / 2 / 3      gemm_kernel   128x128 / 128x64 / 64x64 (any shape)
//   42 / 43        gemm3_kernel  128x64 / 64x64, 3 stages, 4 waves (two / three blocks per CU)
//   44 / 46        the same tiles with 2 stages (three / five blocks per CU)
//   52 / 54        gemm3_kernel  128x128 / 192x128, 2 stages, 8 waves: two blocks per CU
//   47             gemm3_kernel  128x96, 3 stages, 4 waves: 224 tiles for M 2048 x N 1280 (one per CU, each streaming a 96-row weight tile:
//                  the 32x32-latent convs lose nothing on cold weights with it, 7-9 % with 128x128)
//   48             gemm3_kernel  128x80, 3 stages, 4 waves stacked along the rows (wave tile 32x80); M 2048 x N 1280 = 256 tiles, one per CU
//   51             gemm3_kernel  128x128, 3 stages, 8 waves
//   61 / 62        gemm3_kernel  256x256 x 2 stages / 256x128 x 3 stages, 8 waves
//   63             gemm3_kernel  256x320 x 2 stages, 8 waves (wave tile 128x80): exactly ONE tile per CU for the FF1 projection M 2048 x N 10240
//                  (256 tiles) and two rounds for M 8192 x N 5120; 142 FLOP per staged byte pair against 85 for 256x128
//   64             63 with the two wave groups taking turns at a K-step's LDS-DMA refill (ALT above): FF1 56.5 us against 59.5-63.4 (the same idea on
//                  256x128 x 3 stages was 3 % SLOWER than 62 and is not built: the barrier still paces both groups by the slower one)
//   + 1000 / + 2000  force the tile order (n-tiles / m-tiles fastest); plain codes estimate it
//   95 / 96        gemm3_kernel 51 / 62 with fragment reads one half K-step ahead of the MFMAs (the 4-wave and 2-stage tiles gain nothing from it)
//   97 / 98 / 94   62 / 96 / 42 with in-kernel cycle stamps (fie_debug_gemm_stamps; slower, for tools/kstep_stamps.py only)
//   81 / 82        gemm8_kernel  256x256 phased (82: second DMA piece of each phase inside the MFMA cluster, A/B: slower)
//   71 / 73        conv_halo_kernel (conv_halo.hip): a 16x16 output patch x 128 channels per block, the patch's 18x18 halo resident in LDS per
//                  64-channel chunk, weights through a ring; stride-1 same-size convs with H, W % 16 == 0 only (73: with cycle stamps).
//                  K is summed chunk-major (the im2col kernels: tap-major): equal to the other codes to rounding, not bit for bit
//   72 / 74 / 76   conv_halo2_kernel: the same K loop in persistent blocks that prefetch their next tile and defer a tile's stores into the next
//                  tile's first chunk (74: with cycle stamps, 76: the 8-byte-store form, A/B); THE RULE for eligible convs (heuristic_code)
struct TileDim { int code, bm, bn; };
constexpr TileDim kTiles[] = {{1, 128, 128}, {2, 128, 64}, {3, 64, 64}, {42, 128, 64}, {43, 64, 64},
                              {51, 128, 128}, {61, 256, 256}, {62, 256, 128}, {81, 256, 256}, {82, 256, 256},
                              {63, 256, 320}, {95, 128, 128}, {96, 256, 128}, {97, 256, 128}, {98, 256, 128}, {94, 128, 64},
                              {52, 128, 128}, {47, 128, 96}, {54, 192, 128}, {46, 64, 64}, {44, 128, 64}, {48, 128, 80}, {64, 256, 320},
                              {71, 256, 128}, {73, 256, 128}, {72, 256, 128}, {74, 256, 128}, {76, 256, 128}, {77, 64, 16}};

// Heuristic tile code for a shape (the default; the autotuner below and the debug hooks can replace it).
template <int MODE>
int heuristic_code(const fie_ctx* ctx, const GemmArgs& a, bool dma_ok) {
    auto blocks = [&](int bm, int bn) { return (int64_t)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
    const int64_t cus = ctx->num_cus;
    // Selection: tools/tile_table.py (interleaved rounds per shape, profiles/r02_tile_table.md) + tools/tile_trials.py inside the
    // UNet.  The K loop is paced by the per-CU global->LDS issue path (~100 cycles per 1 KiB piece), so a tile's FLOP per
    // staged byte BM*BN/(BM+BN) decides its ceiling -- but only while the grid still fills the CUs: the largest tile that does wins.
    const int64_t b256 = blocks(256, 256), b62 = blocks(256, 128), b51 = blocks(128, 128), b42 = blocks(128, 64);
    int code;
    if (!dma_ok) {
        code = b42 >= cus ? 2 : 3;
    } else if (MODE == 1 && fie_conv_halo_ok(a) && 4 * (int64_t)(a.M / 256) * ((a.N + 127) / 128) >= 3 * cus && a.C2x + a.C3x <= 4 * BK &&      // (1x1 side inputs: each 64 channels cost a drained K-step: the rule takes few of them, the tuner may take more)
               std::find(std::begin(ctx->tune_exclude), std::end(ctx->tune_exclude), 72) == std::end(ctx->tune_exclude)) {      // fie_debug_tune_exclude("72"): the A/B switch
        // stride-1 same-size convs on maps of whole 16x16 patches with enough (patch, 128-channel) tiles to fill the chip: the halo-resident kernel
        // (conv_halo.hip; persistent, deferred stores).  Measured against every im2col code on the VAE's and the UNet's 64x64 / 128x128-latent
        // shapes: -15 to -35 % (profiles/r04_halo_conv.md)
        code = 72;
    } else if (MODE == 1) {
        if (!a.A2 && ((b256 >= cus && (a.N % 256 == 0 || (a.N % 128 != 0 && a.K >= 5760))) ||        // 256x256 phased: VAE 256/512-ch maps, 128x128-latent convs into 320 ch (not with 1x1 side inputs: ring kernels only)
            (a.N % 256 == 0 && a.K >= 8192 && 2 * b256 >= cus))) code = 81;                  // ... and the long-K upsampling convs of the 32x32 level
        else if (a.N % 128 == 0 && b62 >= 150) code = 96;
        else if (a.N % 128 == 0 && a.K >= 5760 && b51 >= 100) code = 51;                     // 32x32-latent convs: 160 x (128x128), 8 waves
        else if (2 * b42 < 3 * cus) code = 43;                                               // stride-2 convs and other small grids
        else code = 42;
    } else if (MODE == 0 && a.N % 320 == 0 && a.N >= 5120 && 4 * blocks(256, 320) >= 3 * cus) {
        // the FF1 projections (M 2048 x N 10240: 256 tiles of 256x320 = one per CU; M 8192 x N 5120: two exact rounds): tile 64 measured 52-58 us
        // on every box of round 3 against 65-80 us for 96 -- but the cold-timed tuner, starting from 96 as the rule, kept 96 on some boxes (one
        // profile run of round 4: 79.7 us x 112 launches = +2.4 ms per edit).  As the RULE it needs a 3 % better challenger to be displaced.
        code = 64;
    } else if (a.N % 256 == 0 && a.K >= 4096 && b256 >= cus) {
        code = 81;
    } else if (a.N % 128 == 0 && b62 >= 150 && (a.N >= 1536 || a.K >= 2048 || a.M >= 16384)) {
        code = 96;
    } else if (a.N >= 2048 && a.K >= 1024 && b51 >= cus) {
        code = 51;
    } else if (a.K <= 640 && a.N <= 640) {
        code = 43;                                                                           // 64x64 / 128x128-latent projections: many small tiles
    } else if (b42 >= cus * 7 / 2 || a.K >= 4096 || (a.K >= 1024 && b42 >= cus)) {
        code = 42;
    } else {
        code = 43;
    }
    return code;
}

// One launch of `code` (order: 0 n-tiles fastest, 1 m-tiles fastest, -1 estimate).
constexpr int64_t kSkTickets = 4096;        // arrival counters at the head of the split-K workspace

// Largest usable split for a tile code: ring kernels with <= 16 accumulator fragments per lane (not 256x256), at least 4 K-steps per
// slice, counters and slabs inside the bound workspace.  Returns 1 when split-K cannot run.
inline int splitk_fit(const fie_ctx* ctx, const GemmArgs& a, int code, int bm, int bn, int want) {
    const bool ring = (code >= 42 && code <= 54) || code == 62 || code == 95 || code == 96;
    if (want <= 1 || !ring || !ctx->sk_ws || !ctx->splitk_mode || (a.w_scale && a.a_scale == 0.f)) return 1;
    const int64_t tiles = (int64_t)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * (a.oscat == 2 ? 4 : 1);
    const int nk = (a.K + BK - 1) / BK / (a.a_scale != 0.f ? 2 : 1);      // fp8 activations: 128 k-values per K-step
    int s = want;
    while (s > 1 && (nk / s < 4 || tiles > kSkTickets || (int64_t)(kSkTickets * 4 + tiles * s * bm * bn * 4) > ctx->sk_bytes)) --s;
    return s;
}

template <int MODE>
int run_code(fie_ctx* ctx, GemmArgs& a, int code, int order, bool dma_ok, int split = 1) {
    const TileDim* t = nullptr;
    for (const TileDim& d : kTiles)
        if (d.code == code) t = &d;
    FIE_REQUIRE(t != nullptr, "unknown tile code %d", code);
    FIE_REQUIRE(code < 40 || code == 77 || dma_ok, "tile code %d: shape not eligible for the LDS-DMA kernels (operands >= 2 GiB, Cin %% 64 != 0 or K1 %% 64 != 0)", code);
    FIE_REQUIRE(!(a.taps2 && (code < 40 || a.w_scale)), "tile code %d: the 2x2 parity convs run on the f16 LDS-DMA kernels only", code);
    FIE_REQUIRE(!(MODE == 1 && a.A2 && (code < 40 || code == 81 || code == 82 || a.w_scale)), "tile code %d: conv + 1x1 side inputs run on the f16 ring kernels and the halo-resident kernel (72) only", code);
    FIE_REQUIRE(!((code >= 71 && code <= 76) && (MODE != 1 || !dma_ok || !fie_conv_halo_ok(a))), "tile code %d (halo-resident conv): stride-1 same-size 3x3 conv with H, W %% 16 == 0, Cin %% 64 == 0, f16 weights only", code);
    if (order < 0) {
        // Tile order = which operand an XCD re-streams past its 4 MiB L2.  Consecutive tile ids run on one XCD (xcd_remap), so an
        // XCD owns T/8 consecutive tiles: with n fastest that is `dm` row blocks x up to all column tiles, with m fastest the
        // transpose.  Fabric bytes per XCD ~ (row blocks touched) x (A bytes per row block) + (column tiles touched) x (W bytes per
        // column tile); pick the smaller (FF1 at M 2048: 8 XCDs x all 26 MB of W with n fastest, 247 MB measured by PMC in round 1,
        // against 8 x (3.3 MB of W + all 5.2 MB of A) with m fastest).  The 3x3 im2col view re-reads an input row block ~once.
        const int64_t nbm = (a.M + t->bm - 1) / t->bm, nbn = (a.N + t->bn - 1) / t->bn;
        const int64_t per_xcd = (nbm * nbn + 7) / 8;
        const double a_tile = (double)t->bm * a.K * 2 * (MODE == 1 ? 1.0 / 9 : 1.0), w_tile = (double)t->bn * a.K * 2;
        auto fabric = [&](int64_t fast, int64_t slow, double fast_tile, double slow_tile) {       // `fast` tiles per row of the id space
            const int64_t rows = (per_xcd + fast - 1) / fast;                                       // slow-index values an XCD touches
            return (double)(rows < slow ? rows : slow) * slow_tile + (double)(per_xcd < fast ? per_xcd : fast) * fast_tile;
        };
        order = fabric(nbm, nbn, a_tile, w_tile) < fabric(nbn, nbm, w_tile, a_tile) ? 1 : 0;
    }
    const bool x8 = a.w_scale && a.a_scale != 0.f;           // e4m3 activations x e4m3 weights (gemm_x8.hip): its own tile set
    if (x8) {
        FIE_REQUIRE(dma_ok && !a.A2 && !a.taps2, "fp8 activations: plain GEMM / 3x3 conv views on the LDS-DMA kernels only");
        if (code == 64) code = 63;                          // no alternating-refill form of the fp8 kernels
        if (MODE == 1 && code == 63) code = 62;
        code = code == 96 || code == 81 || code == 61 ? 62 : code == 95 ? 51 : code == 44 || code == 2 ? 42 : code == 46 || code == 3 || code == 1 ? 43 : code;
        FIE_REQUIRE(code == 42 || code == 43 || code == 47 || code == 51 || code == 52 || code == 54 || code == 62 || code == 63, "tile code %d has no fp8-activation kernel", code);
        for (const TileDim& d : kTiles)
            if (d.code == code) t = &d;
    } else if (a.w_scale) {          // fp8 weights: the three W8 ring tiles (gemm_w8.hip)
        FIE_REQUIRE(dma_ok, "fp8 weights: shape not eligible for the LDS-DMA kernels (operands >= 2 GiB, Cin %% 64 != 0 or K1 %% 64 != 0)");
        code = (code == 43 || code == 46 || code == 3) ? 43 : (code == 42 || code == 44 || code == 2) ? 42 : (code == 52 || code == 54) ? code : 62;
        for (const TileDim& d : kTiles)
            if (d.code == code) t = &d;
    }
    a.nbm = (a.M + t->bm - 1) / t->bm;
    a.nbn = (a.N + t->bn - 1) / t->bn;
    split = splitk_fit(ctx, a, code, t->bm, t->bn, split);
    a.splitk = split;
    a.sk_tickets = static_cast<unsigned*>(ctx->sk_ws);
    a.sk_slabs = reinterpret_cast<float*>(static_cast<char*>(ctx->sk_ws) + kSkTickets * 4);
    a.order = order;
    a.probe = ctx->gemm_probe;
    a.epi_prefetch = ctx->epi_prefetch;
    a.stamps = (code == 97 || code == 98 || code == 94 || code == 73 || code == 74) ? ctx->gemm_stamps : nullptr;
    snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "%s<%dx%d> (%s, tile code %d)", code >= 90 ? "gemm3_kernel+prefetch" : code >= 80 ? "gemm8_kernel" : code == 77 ? "conv_thin_kernel" : code >= 71 && code <= 76 ? (code == 71 || code == 73 ? "conv_halo_kernel" : "conv_halo2_kernel") : code >= 40 ? "gemm3_kernel" : "gemm_kernel",
             t->bm, t->bn, MODE == 1 ? "conv3x3" : "gemm", code);
    if (split > 1) snprintf(ctx->last_kernel + strlen(ctx->last_kernel) - 1, 24, ", split-K %d)", split);
    if (MODE == 1)
        FIE_DESC(ctx, "conv M=%d N=%d K=%d in=%dx%dx%d s%d u%d%s%s code=%d flop=%.0f", a.M * (a.oscat == 2 ? 4 : 1), a.N, a.K, a.H, a.W, a.Cin, a.stride, a.ups,
                 a.taps2 ? " up2x-parity" : "", a.A2 ? " +1x1" : "", code + 10000 * (split > 1 ? split : 0), 2.0 * a.M * a.N * a.K * (a.oscat == 2 ? 4 : 1));
    else
        FIE_DESC(ctx, "gemm M=%d N=%d K=%d act=%d%s%s%s code=%d flop=%.0f", a.M, a.N, a.K, a.act, a.res ? " +res" : "", a.ln_tab ? " +ln" : "", a.w_scale ? (a.a_scale != 0.f ? " a8w8" : " w8") : "", code + 10000 * (split > 1 ? split : 0), 2.0 * a.M * a.N * a.K);
    if (x8) {
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "gemm3x8_kernel<%dx%d> (gemm, fp8 activations x fp8 weights, tile code %d%s", t->bm, t->bn, code, split > 1 ? "" : ")");
        if (split > 1) snprintf(ctx->last_kernel + strlen(ctx->last_kernel), 24, ", split-K %d)", split);
        return fie_launch_gemm_x8(ctx, a, code, MODE == 1);
    }
    if (a.w_scale) {
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "gemm3w8_kernel<%dx%d> (%s, fp8 weights, tile code %d)", t->bm, t->bn, MODE == 1 ? "conv3x3" : "gemm", code);
        return fie_launch_gemm_w8(ctx, a, MODE == 1, code);
    }
    const dim3 grid((unsigned)(a.nbm * a.nbn * (a.oscat == 2 ? 4 : 1) * split)), block(256);
    constexpr int M3 = MODE == 1 ? 2 : 0;
    if (a.ln_tab) {
        FIE_REQUIRE(MODE == 0 && is_ln_code(code) && split == 1, "LayerNorm-folded GEMM: tile code %d not built for it", code);
        FIE_REQUIRE((code == 64) == (a.act == FIE_ACT_GEGLU), "LayerNorm-folded GEMM: GEGLU runs on tile 64 and nothing else does (code %d, act %d)", code, a.act);
        switch (code) {
            case 42: launch_ring_ln<128, 64, 3, 4>(ctx, a, grid); break;
            case 96: launch_ring_ln<256, 128, 3, 8, true>(ctx, a, grid); break;
            case 64: launch_ring_ln<256, 320, 2, 8, false, true>(ctx, a, grid); break;
        }
        FIE_LAUNCH_CHECK();
        return FIE_OK;
    }
    switch (code) {
        case 1: fie_launch(ctx, (gemm_kernel<128, 128, MODE>), grid, block, 0, a); break;
        case 2: fie_launch(ctx, (gemm_kernel<128, 64, MODE>), grid, block, 0, a); break;
        case 3: fie_launch(ctx, (gemm_kernel<64, 64, MODE>), grid, block, 0, a); break;
        case 42: launch_ring<128, 64, 3, M3, 4>(ctx, a, grid); break;
        case 43: launch_ring<64, 64, 3, M3, 4>(ctx, a, grid); break;
        case 51: launch_ring<128, 128, 3, M3, 8>(ctx, a, grid); break;
        case 61: launch_ring<256, 256, 2, M3, 8>(ctx, a, grid); break;
        case 62: launch_ring<256, 128, 3, M3, 8>(ctx, a, grid); break;
        case 63:
            FIE_REQUIRE(MODE == 0, "tile code 63 (256x320) is built for the GEMM view only");
            launch_ring<256, 320, 2, 0, 8>(ctx, a, grid);
            break;
        case 64:
            FIE_REQUIRE(MODE == 0, "tile code 64 (256x320, alternating refill) is built for the GEMM view only");
            launch_ring<256, 320, 2, 0, 8, false, false, true>(ctx, a, grid);
            break;
        case 52: launch_ring<128, 128, 2, M3, 8>(ctx, a, grid); break;
        case 47: launch_ring<128, 96, 3, M3, 4>(ctx, a, grid); break;
        case 48: launch_ring<128, 80, 3, M3, 4>(ctx, a, grid); break;       // weight rows past the packed matrix (80 does not divide Npad) read as zero through the descriptor
        case 54: launch_ring<192, 128, 2, M3, 8>(ctx, a, grid); break;
        case 46: launch_ring<64, 64, 2, M3, 4>(ctx, a, grid); break;
        case 44: launch_ring<128, 64, 2, M3, 4>(ctx, a, grid); break;
        case 95: launch_ring<128, 128, 3, M3, 8, true>(ctx, a, grid); break;
        case 96: launch_ring<256, 128, 3, M3, 8, true>(ctx, a, grid); break;
        case 97: launch_ring<256, 128, 3, M3, 8, false, true>(ctx, a, grid); break;
        case 98: launch_ring<256, 128, 3, M3, 8, true, true>(ctx, a, grid); break;
        case 94: launch_ring<128, 64, 3, M3, 4, false, true>(ctx, a, grid); break;
        case 71: return fie_launch_conv_halo(ctx, a, 0);
        case 73: return fie_launch_conv_halo(ctx, a, 1);
        case 72: return fie_launch_conv_halo(ctx, a, 2);
        case 76: return fie_launch_conv_halo(ctx, a, 5);
        case 74: return fie_launch_conv_halo(ctx, a, 4);
        case 77: return fie_launch_conv_thin(ctx, a);
        case 81: return fie_launch_gemm8(ctx, a, MODE == 1, 0);
        case 82: return fie_launch_gemm8(ctx, a, MODE == 1, 1);   // A/B: second DMA piece of a phase issued from inside the MFMA cluster (measured slower)
    }
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

// ---- per-shape autotune (fie_gemm_autotune): the first eager launch of a shape times every eligible tile on a scratch output
// and remembers the fastest; later launches (and stream captures, which never tune) use the remembered code.  The candidates
// differ in tile shape, ring depth and blocks per CU; every im2col tile accumulates K in the same order, so the choice among them does not change
// the result (the halo-resident conv, code 72, sums chunk-major and a split-K code in slices: those two move the last f16 bit).  Selection by
// measurement instead of by rule: which tile wins depends on how the grid quantises onto 256 CUs
// and on whether two blocks share a CU (their epilogues and DMA issue overlap), see DESIGN.md.
// Every timed launch sees what a launch inside the network sees: weights COLD (the 256 MB Infinity Cache is flushed by a
// 384 MB memset; a UNet evaluation streams 2.6 GB of weights, so no layer finds its own in cache) and activations WARM (they
// were just written by the previous layer; re-read here after the flush).  Timed warm, the two-stage tiles win almost
// everywhere and then lose inside the network, where their one K-step of prefetch does not cover an HBM miss
// (tools/cold_weights.py).
constexpr size_t kFlushBytes = 384u << 20;

template <int MODE>
int autotune(fie_ctx* ctx, GemmArgs& a, int guess, bool dma_ok) {
    static const int kRing[] = {43, 46, 42, 44, 51, 52, 54, 96, 81, 63, 47, 48, 64, 72};      // 47 (128x96): FIE_TUNE_47=0 leaves it out
    static const bool use47 = !(getenv("FIE_TUNE_47") && getenv("FIE_TUNE_47")[0] == '0');
    static const int kW8[] = {43, 42, 62, 52, 54};
    static const int kX8[] = {43, 42, 47, 51, 52, 54, 62, 63};
    const size_t bytes = (size_t)a.M * (a.oscat ? 4 : 1) * (size_t)a.ldc * sizeof(half_t);     // a parity conv scatters its M rows over 4 M output rows
    if (bytes > ctx->tune_bytes) {
        if (ctx->tune_buf) (void)hipFree(ctx->tune_buf);
        ctx->tune_buf = nullptr;
        ctx->tune_bytes = 0;
        if (hipMalloc(&ctx->tune_buf, bytes) != hipSuccess) { (void)hipGetLastError(); return guess; }
        ctx->tune_bytes = bytes;
    }
    if (!ctx->tune_flush && hipMalloc(&ctx->tune_flush, kFlushBytes) != hipSuccess) { (void)hipGetLastError(); ctx->tune_flush = nullptr; return guess; }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return guess;
    (void)hipDeviceSynchronize();                          // nothing else on the device while the candidates are timed
    GemmArgs t = a;
    t.C = static_cast<half_t*>(ctx->tune_buf);           // a residual that aliases C is still read from the caller's buffer
    auto blocks = [&](int bm, int bn) { return (int64_t)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
    auto time_of = [&](int code, int split = 1) -> float {
        float ms[7];
        for (int rep = -1; rep < 7; ++rep) {               // rep -1: untimed (first use of the kernel)
            (void)hipMemsetAsync(ctx->tune_flush, rep & 1, kFlushBytes, ctx->stream);
            if ((size_t)a.a1_bytes <= kFlushBytes) (void)hipMemcpyAsync(ctx->tune_flush, a.A1, (size_t)a.a1_bytes, hipMemcpyDeviceToDevice, ctx->stream);
            if (a.A2 && a.A2 != a.A1 && (size_t)a.a2_bytes <= kFlushBytes) (void)hipMemcpyAsync(ctx->tune_flush, a.A2, (size_t)a.a2_bytes, hipMemcpyDeviceToDevice, ctx->stream);
            (void)hipEventRecord(e0, ctx->stream);
            if (run_code<MODE>(ctx, t, code, -1, dma_ok, split) != FIE_OK) return 1e30f;
            (void)hipEventRecord(e1, ctx->stream);
            if (hipEventSynchronize(e1) != hipSuccess) return 1e30f;
            if (rep >= 0) (void)hipEventElapsedTime(&ms[rep], e0, e1);
        }
        std::sort(ms, ms + 7);
        return (ms[2] + ms[3] + ms[4]) / 3.f;              // mean of the middle three of seven
    };
    static const bool verbose = getenv("FIE_TUNE_VERBOSE") && getenv("FIE_TUNE_VERBOSE")[0] == '1';
    const float t_guess = time_of(guess);
    if (verbose) fprintf(stderr, "[fie tune] %s M=%d N=%d K=%d: rule %d %.1f us", MODE == 1 ? "conv" : "gemm", a.M, a.N, a.K, guess, t_guess * 1e3f);
    int best = guess;
    float t_best = t_guess * 0.97f;                          // a challenger has to win by 3 %
    const bool x8 = a.w_scale && a.a_scale != 0.f;
    const int* cand = x8 ? kX8 : a.w_scale ? kW8 : kRing;
    const int ncand = x8 ? 8 : a.w_scale ? 5 : 14;
    auto excluded = [&](int c) {
        for (int e : ctx->tune_exclude)
            if (e == c) return e != 0;
        return false;
    };
    for (int i = 0; i < ncand; ++i) {
        const int c = cand[i];
        if (c == guess || excluded(c)) continue;
        if ((c == 43 || c == 46) && blocks(64, 64) > 64 * ctx->num_cus) continue;       // tens of thousands of tiny tiles: never wins
        if ((c == 81 || c == 96 || c == 62) && 2 * blocks(256, 128) < ctx->num_cus) continue;
        if ((c == 63 || c == 64) && (MODE != 0 || a.N % 320 != 0 || 2 * blocks(256, 320) < ctx->num_cus)) continue;
        if (c == 47 && !use47) continue;
        if (c == 48 && (a.N % 80 != 0 || blocks(128, 80) > 2 * ctx->num_cus)) continue;      // the exact-fit tile of the N = 1280 projections: small grids only
        if (x8 && MODE == 1 && c == 63) continue;
        if (x8 && (c == 62 || c == 51) && 2 * blocks(c == 62 ? 256 : 128, 128) < ctx->num_cus) continue;
        if (c == 81 && MODE == 1 && a.A2) continue;            // side inputs: ring kernels only
        if (c == 72 && (MODE != 1 || !fie_conv_halo_ok(a) || 2 * (int64_t)(a.M / 256) * ((a.N + 127) / 128) < ctx->num_cus)) continue;
        const float tc = time_of(c);
        if (verbose) fprintf(stderr, ", %d %.1f", c, tc * 1e3f);
        if (tc < t_best) { t_best = tc; best = c; }
    }
    // split-K: big tiles whose grid leaves CUs idle (the M = 2048 class: 80 tiles of 256x128 on 256 CUs) with the K-steps of a tile dealt
    // to 2-4 blocks, reduced in the launch (gemm_common.h: splitk_reduce).  Changes the fp32 summation order, so unlike the tile choice
    // it is visible in the last bit of some f16 outputs; fixed per (shape, choice), hence deterministic within a process.
    if ((!a.w_scale || x8) && ctx->sk_ws && ctx->splitk_mode && !excluded(10000)) {
        struct SplitCand { int code, bm, bn, per_cu; };
        static const SplitCand kSplitF16[] = {{96, 256, 128, 1}, {95, 128, 128, 1}, {47, 128, 96, 1}, {54, 192, 128, 2}, {52, 128, 128, 2}};
        static const SplitCand kSplitX8[] = {{62, 256, 128, 1}, {51, 128, 128, 1}, {47, 128, 96, 1}, {54, 192, 128, 2}, {52, 128, 128, 2}};
        const SplitCand* sc = x8 ? kSplitX8 : kSplitF16;
        for (int ci = 0; ci < 5; ++ci) {
            const SplitCand& c = sc[ci];
            if (excluded(c.code)) continue;
            const int64_t nb = blocks(c.bm, c.bn) * (a.oscat == 2 ? 4 : 1);
            if (nb >= ctx->num_cus * c.per_cu) continue;                 // the grid already fills the chip
            for (int sp = 2; sp <= 4; ++sp) {
                if (nb * sp > (int64_t)ctx->num_cus * c.per_cu * 21 / 20) break;
                if (splitk_fit(ctx, a, c.code, c.bm, c.bn, sp) != sp) continue;
                const float tc = time_of(c.code, sp);
                if (verbose) fprintf(stderr, ", %d/s%d %.1f", c.code, sp, tc * 1e3f);
                if (tc < t_best) { t_best = tc; best = c.code + 10000 * sp; }
            }
        }
    }
    if (verbose) fprintf(stderr, " -> %d\n", best);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return best;
}

template <int MODE>
int launch(fie_ctx* ctx, GemmArgs& a) {
    // LDS-DMA kernels (codes >= 40): 32-bit buffer offsets (operands < 2 GiB) and K-steps that never straddle a 3x3 tap / the A1|A2 seam
    // ... and an epilogue on 32-bit buffer offsets (gemm_common.h): output / residual spans of at most 1 GiB
    const int64_t c_span = ((int64_t)((a.oscat ? 4 * (int64_t)a.M : a.M) - 1) * a.ldc + a.N) * 2, r_span = a.res ? ((int64_t)(a.M - 1) * a.ldr + a.N) * 2 : 0;
    const bool dma_ok = a.a1_bytes < (1ll << 31) && a.a2_bytes < (1ll << 31) && a.w_bytes < (1ll << 31) && c_span <= (1ll << 30) && r_span <= (1ll << 30) &&
                        (MODE == 1 ? a.Cin % (a.a_scale != 0.f ? 2 * BK : BK) == 0 : (a.K1 == a.K || a.K1 % BK == 0));
    FIE_REQUIRE(!(MODE == 1 && a.A2 && !dma_ok), "conv + 1x1 side inputs: tensors too large for the LDS-DMA kernels");
    if (MODE == 1 && a.gna_tab) {                          // the input's GroupNorm applied on the resident halo: one kernel family, no choice to make
        FIE_REQUIRE(dma_ok && fie_conv_halo_gna_ok(a), "fie_conv3x3_gn_nhwc_f16: shape / epilogue not built for the fused form (ask fie_conv3x3_gn_ok first)");
        return run_code<MODE>(ctx, a, 72, -1, dma_ok, 1);
    }
    if (MODE == 0 && a.ln_tab) {
        // LayerNorm folded in: four tiles, chosen by rule (no tuner: the choice among them does not change the sums, every one adds K in the same order)
        FIE_REQUIRE(dma_ok, "fie_gemm_ln_f16: operands too large for the LDS-DMA kernels");
        auto blocks = [&](int bm, int bn) { return (int64_t)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
        int code;
        if (a.act == FIE_ACT_GEGLU) code = 64;
        else if (a.N % 128 == 0 && blocks(256, 128) >= 150 && (a.N >= 1536 || a.M >= 16384)) code = 96;
        else code = 42;
        const int forced = ctx->force_tile % 1000;
        if (forced && is_ln_code(forced) && (forced == 64) == (a.act == FIE_ACT_GEGLU)) code = forced;
        return run_code<MODE>(ctx, a, code, -1, dma_ok, 1);
    }
    int code = heuristic_code<MODE>(ctx, a, dma_ok);
    int order = -1;                // 0: n-tiles fastest (an XCD owns a range of activation rows), 1: m-tiles fastest; -1: estimate
    int split = 1;                 // encoded choices: split-K factor * 10000 + (1000 / 2000: forced tile order) + tile code
    bool pinned = false;
    auto decode = [&](int v) { split = v / 10000 > 1 ? v / 10000 : 1; v %= 10000; order = v >= 2000 ? 1 : v >= 1000 ? 0 : -1; code = v % 1000; };
    for (int i = 0; i < ctx->n_overrides; ++i) {
        const fie_tile_override& o = ctx->overrides[i];
        if (o.mode == (MODE == 1) && o.M == a.M && o.N == a.N && o.K == a.K) { decode(o.code); pinned = true; }
    }
    if (ctx->force_tile) { decode(ctx->force_tile); pinned = true; }
    // at most 16 output channels (conv_out of the VAE / UNet, the conditioning embedding's first convs): the direct-load strip kernel of conv_thin.hip,
    // by rule and without the tuner (one kernel family; profiles/r04_conv_thin.md).  fie_debug_tune_exclude("77") is the A/B switch.
    if (MODE == 1 && !pinned && !ctx->gemm_probe && fie_conv_thin_ok(a) && a.M >= 131072 &&      // (the 128x128-latent conv_outs, K 2880 / 4608 on few strips: the ring tiles are faster)
        std::find(std::begin(ctx->tune_exclude), std::end(ctx->tune_exclude), 77) == std::end(ctx->tune_exclude))
        return run_code<MODE>(ctx, a, 77, 0, dma_ok, 1);
    if (ctx->autotune && !pinned && dma_ok && !ctx->gemm_probe) {      // 1: tune shapes not met before, 2: remembered shapes only
        const fie_tune_key key{MODE, a.M, a.N, a.K, a.K1, MODE == 1 ? a.stride * 2 + a.ups + 8 * a.taps2 + 16 * (a.A2 != nullptr) + 32 * (a.A3 != nullptr) : 0, a.w_scale == nullptr ? 0 : a.a_scale != 0.f ? 2 : 1};
        auto it = ctx->tuned.find(key);
        if (it != ctx->tuned.end()) {
            decode(it->second);
        } else if (ctx->autotune == 1 && !ctx->recording) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
                const int best = autotune<MODE>(ctx, a, code, dma_ok);
                ctx->tuned[key] = best;
                decode(best);
            }
        }
    }
    return run_code<MODE>(ctx, a, code, order, dma_ok, split);
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

template <int MODE>
hipError_t ring_attrs() {
    hipError_t e = ring_attr<128, 64, 3, MODE, 4>();
    if (e == hipSuccess) e = ring_attr<64, 64, 3, MODE, 4>();
    if (e == hipSuccess) e = ring_attr<128, 128, 3, MODE, 8>();
    if (e == hipSuccess) e = ring_attr<256, 256, 2, MODE, 8>();
    if (e == hipSuccess) e = ring_attr<256, 128, 3, MODE, 8>();
    if (e == hipSuccess && MODE == 0) e = ring_attr<256, 320, 2, 0, 8>();
    if (e == hipSuccess && MODE == 0) e = ring_attr<256, 320, 2, 0, 8, false, false, true>();
    if (e == hipSuccess) e = ring_attr<128, 128, 2, MODE, 8>();
    if (e == hipSuccess) e = ring_attr<128, 96, 3, MODE, 4>();
    if (e == hipSuccess) e = ring_attr<128, 80, 3, MODE, 4>();
    if (e == hipSuccess) e = ring_attr<192, 128, 2, MODE, 8>();
    if (e == hipSuccess) e = ring_attr<64, 64, 2, MODE, 4>();
    if (e == hipSuccess) e = ring_attr<128, 64, 2, MODE, 4>();
    if (e == hipSuccess) e = ring_attr<128, 128, 3, MODE, 8, true>();
    if (e == hipSuccess) e = ring_attr<256, 128, 3, MODE, 8, true>();
    if (e == hipSuccess) e = ring_attr<256, 128, 3, MODE, 8, false, true>();
    if (e == hipSuccess) e = ring_attr<256, 128, 3, MODE, 8, true, true>();
    if (e == hipSuccess) e = ring_attr<128, 64, 3, MODE, 4, false, true>();
    return e;
}

}  // namespace

int fie_gemm_init(void) {
    hipError_t e = ring_attrs<0>();
    if (e == hipSuccess) e = ring_attrs<2>();
    if (e == hipSuccess) e = ring_attr_ln<128, 64, 3, 4>();
    if (e == hipSuccess) e = ring_attr_ln<256, 128, 3, 8, true>();
    if (e == hipSuccess) e = ring_attr_ln<256, 320, 2, 8, false, true>();
    if (e != hipSuccess) {
        fie_set_error("gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return FIE_EHIP;
    }
    return FIE_OK;
}

extern "C" {

int fie_debug_force_tile(fie_ctx* ctx, int t) {
    FIE_REQUIRE(ctx != nullptr, "fie_debug_force_tile: ctx is NULL");
    ctx->force_tile = t;                     // 0 = heuristic; 1000 + code: n-tiles fastest, 2000 + code: m-tiles fastest
    return FIE_OK;
}

int fie_debug_tile_override(fie_ctx* ctx, const char* spec) {
    FIE_REQUIRE(ctx != nullptr, "fie_debug_tile_override: ctx is NULL");
    ctx->n_overrides = 0;
    if (!spec) return 0;
    const char* q = spec;
    while (*q && ctx->n_overrides < 32) {
        fie_tile_override o;
        int used = 0;
        if (sscanf(q, "%d,%d,%d,%d=%d%n", &o.mode, &o.M, &o.N, &o.K, &o.code, &used) != 5) break;
        ctx->overrides[ctx->n_overrides++] = o;
        q += used;
        if (*q == ';') ++q;
    }
    return ctx->n_overrides;
}

int fie_debug_gemm_probe(fie_ctx* ctx, int mode) {
    FIE_REQUIRE(ctx != nullptr && mode >= 0 && mode <= 4, "fie_debug_gemm_probe: bad argument");
    ctx->gemm_probe = mode;
    return FIE_OK;
}

int fie_debug_epilogue_prefetch(fie_ctx* ctx, int on) {
    FIE_REQUIRE(ctx != nullptr, "fie_debug_epilogue_prefetch: ctx is NULL");
    ctx->epi_prefetch = on ? 1 : 0;
    return FIE_OK;
}

int fie_gn_stats_target(fie_ctx* ctx, void* partial, int64_t rows_per_image, int groups) {
    FIE_REQUIRE(ctx != nullptr && (partial == nullptr || (rows_per_image > 0 && groups > 0)), "fie_gn_stats_target: bad argument");
    ctx->gn_target = static_cast<float*>(partial);
    ctx->gn_target_rows = rows_per_image;
    ctx->gn_target_groups = groups;
    return FIE_OK;
}

int fie_splitk_workspace(fie_ctx* ctx, void* ws, int64_t bytes) {
    FIE_REQUIRE(ctx != nullptr && (ws == nullptr || bytes >= kSkTickets * 4 + (1 << 20)), "fie_splitk_workspace: bad argument (at least 1 MiB + 16 KiB)");
    ctx->sk_ws = ws;
    ctx->sk_bytes = ws ? bytes : 0;
    return FIE_OK;
}

int fie_debug_splitk(fie_ctx* ctx, int mode) {
    FIE_REQUIRE(ctx != nullptr && (mode == 0 || mode == 1), "fie_debug_splitk: bad argument");
    ctx->splitk_mode = mode;
    return FIE_OK;
}

int fie_gemm_autotune(fie_ctx* ctx, int on) {
    FIE_REQUIRE(ctx != nullptr && on >= 0 && on <= 2, "fie_gemm_autotune: bad argument");
    ctx->autotune = on;
    if (on != 1 && (ctx->tune_buf || ctx->tune_flush)) {   // the timing scratch is only needed while shapes are still being met
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->tune_buf) (void)hipFree(ctx->tune_buf);
        if (ctx->tune_flush) (void)hipFree(ctx->tune_flush);
        ctx->tune_buf = ctx->tune_flush = nullptr;
        ctx->tune_bytes = 0;
    }
    return FIE_OK;
}

int fie_debug_tune_exclude(fie_ctx* ctx, const char* codes) {
    FIE_REQUIRE(ctx != nullptr, "fie_debug_tune_exclude: ctx is NULL");
    memset(ctx->tune_exclude, 0, sizeof(ctx->tune_exclude));
    ctx->tuned.clear();                                   // choices made under the previous list are forgotten
    int n = 0;
    for (const char* q = codes; q && *q && n < 15;) {
        char* end = nullptr;
        const long v = strtol(q, &end, 10);
        if (end == q) break;
        if (v > 0) ctx->tune_exclude[n++] = (int)v;
        q = *end == ',' ? end + 1 : end;
    }
    return n;
}

int fie_gemm_autotune_report(fie_ctx* ctx, char* buf, int cap) {
    FIE_REQUIRE(ctx != nullptr && buf != nullptr && cap > 0, "fie_gemm_autotune_report: bad argument");
    int n = 0;
    buf[0] = 0;
    for (const auto& kv : ctx->tuned) {
        const fie_tune_key& k = kv.first;
        const int w = snprintf(buf + n, (size_t)(cap - n), "%s M=%d N=%d K=%d K1=%d geom=%d w8=%d -> %d\n", k.mode ? "conv" : "gemm", k.M, k.N, k.K, k.K1, k.geom, k.w8, kv.second);
        if (w < 0 || w >= cap - n) break;
        n += w;
    }
    return (int)ctx->tuned.size();
}

// The inverse of fie_gemm_autotune_report: remembered choices from its text (one "gemm|conv M= N= K= K1= geom= w8= -> code" line per problem; other
// lines are skipped).  A host saves the report of a tuned process and loads it into the next one: no timing launches at start-up, and the SAME kernels
// (a split-K choice is visible in the last f16 bit) on every box.  Entries replace remembered ones for the same problem.  Returns the number loaded.
int fie_gemm_autotune_load(fie_ctx* ctx, const char* text) {
    FIE_REQUIRE(ctx != nullptr && text != nullptr, "fie_gemm_autotune_load: NULL argument");
    int n = 0;
    for (const char* q = text; *q;) {
        char kind[8] = "";
        fie_tune_key k{};
        int code = 0;
        if (sscanf(q, "%7s M=%d N=%d K=%d K1=%d geom=%d w8=%d -> %d", kind, &k.M, &k.N, &k.K, &k.K1, &k.geom, &k.w8, &code) == 8 &&
            (!strcmp(kind, "gemm") || !strcmp(kind, "conv")) && code > 0) {
            k.mode = kind[0] == 'c';
            ctx->tuned[k] = code;
            ++n;
        }
        const char* nl = strchr(q, '\n');
        if (!nl) break;
        q = nl + 1;
    }
    return n;
}

int fie_debug_gemm_stamps(fie_ctx* ctx, void* buf) {
    FIE_REQUIRE(ctx != nullptr, "fie_debug_gemm_stamps: NULL ctx");
    ctx->gemm_stamps = static_cast<unsigned*>(buf);
    return FIE_OK;
}

const char* fie_debug_last_gemm_kernel(fie_ctx* ctx) { return ctx ? ctx->last_kernel : ""; }

// The one-shot GroupNorm target (fie_gn_stats_target) is taken -- and the context disarmed -- at the very TOP of every GEMM / conv entry,
// before any argument check can return: a call that fails validation must not leave it armed for an unrelated later launch (whose M
// might not fit the buffer the target was sized for).
struct GnTarget { float* partial; int64_t rows; int groups; };
static GnTarget grab_gn_target(fie_ctx* ctx) {
    GnTarget t = {nullptr, 0, 0};
    if (ctx) { t = {ctx->gn_target, ctx->gn_target_rows, ctx->gn_target_groups}; ctx->gn_target = nullptr; }
    return t;
}

static int take_gn_target(const char* who, const GnTarget& t, GemmArgs& a) {
    if (!t.partial) return FIE_OK;
    a.gn_partial = t.partial;
    a.gn_rows = (int)t.rows;
    a.gn_G = t.groups;
    FIE_REQUIRE(a.act != FIE_ACT_GEGLU && a.N % a.gn_G == 0, "%s: GroupNorm statistics: N=%d not divisible into %d groups", who, a.N, a.gn_G);
    a.gn_cg = a.N / a.gn_G;
    FIE_REQUIRE(a.gn_cg == 4 || a.gn_cg == 8 || a.gn_cg == 16, "%s: GroupNorm statistics need 4, 8 or 16 channels per group (got %d)", who, a.gn_cg);
    FIE_REQUIRE(a.gn_rows > 0 && a.gn_rows % 32 == 0 && a.M % a.gn_rows == 0, "%s: GroupNorm statistics: M=%d rows, %d per image (must be a multiple of 32)", who, a.M, a.gn_rows);
    return FIE_OK;
}

static int gemm_impl(const char* who, fie_ctx* ctx, const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2,
                     const void* Wpacked, int64_t ldw, const float* w_scale, void* C, int64_t ldc, int M, int N, int K, const void* bias,
                     const void* rowbias, int64_t ld_rowbias, int rows_per_batch, const void* residual, int64_t ldr, float scale, int act) {
    const GnTarget gn = grab_gn_target(ctx);
    FIE_REQUIRE(ctx && A1 && Wpacked && C, "%s: NULL ctx/A1/W/C", who);
    FIE_REQUIRE(M > 0 && N > 0 && K > 0, "%s: bad shape M=%d N=%d K=%d", who, M, N, K);
    FIE_REQUIRE(K % 8 == 0 && K1 % 8 == 0 && K1 > 0 && K1 <= K, "%s: K=%d K1=%d must be multiples of 8", who, K, K1);
    FIE_REQUIRE(lda1 % 8 == 0 && lda1 >= K1, "%s: lda1=%lld invalid", who, (long long)lda1);
    FIE_REQUIRE(K1 == K || (A2 && lda2 % 8 == 0 && lda2 >= K - K1), "%s: A2/lda2 invalid for K1 < K", who);
    FIE_REQUIRE(ldw % BK == 0 && ldw >= K, "%s: ldw=%lld must be a multiple of 64 covering K", who, (long long)ldw);
    FIE_REQUIRE(!rowbias || rows_per_batch > 0, "%s: rowbias needs rows_per_batch", who);
    if (int e = check_epilogue(who, N, ldc, residual, ldr, act)) return e;
    GemmArgs a = {};
    a.A1 = (const half_t*)A1; a.lda1 = lda1; a.K1 = K1; a.A2 = (const half_t*)A2; a.lda2 = lda2;
    a.Wt = (const half_t*)Wpacked; a.ldw = ldw; a.w_scale = w_scale; a.C = (half_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.bias = (const half_t*)bias; a.rowbias = (const half_t*)rowbias; a.ld_rowbias = ld_rowbias;
    a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    a.res = (const half_t*)residual; a.ldr = ldr; a.scale = scale; a.act = act;
    a.a1_bytes = ((int64_t)(M - 1) * lda1 + K1) * 2;
    a.a2_bytes = A2 ? ((int64_t)(M - 1) * lda2 + (K - K1)) * 2 : 0;
    a.w_bytes = fie_roundup(N, 128) * ldw * (w_scale ? 1 : 2);
    if (int rc = take_gn_target(who, gn, a)) return rc;
    return launch<0>(ctx, a);
}

int fie_gemm_f16(fie_ctx* ctx, const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2,
                 const void* Wpacked, int64_t ldw, void* C, int64_t ldc, int M, int N, int K, const void* bias,
                 const void* rowbias, int64_t ld_rowbias, int rows_per_batch, const void* residual, int64_t ldr,
                 float scale, int act) {
    return gemm_impl("fie_gemm_f16", ctx, A1, lda1, K1, A2, lda2, Wpacked, ldw, nullptr, C, ldc, M, N, K, bias, rowbias, ld_rowbias,
                     rows_per_batch, residual, ldr, scale, act);
}

int fie_gemm_ln_f16(fie_ctx* ctx, const void* X, int64_t ldx, const void* Wfolded, int64_t ldw, const float* ln_tab, float eps, void* C, int64_t ldc,
                    int M, int N, int K, int act) {
    const GnTarget gn = grab_gn_target(ctx);
    (void)gn;
    FIE_REQUIRE(ctx && X && Wfolded && ln_tab && C, "fie_gemm_ln_f16: NULL ctx/X/W/ln_tab/C");
    FIE_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0, "fie_gemm_ln_f16: bad shape M=%d N=%d K=%d (K must be a multiple of 64)", M, N, K);
    FIE_REQUIRE(ldx % 8 == 0 && ldx >= K, "fie_gemm_ln_f16: ldx=%lld invalid", (long long)ldx);
    FIE_REQUIRE(ldw % BK == 0 && ldw >= K, "fie_gemm_ln_f16: ldw=%lld must be a multiple of 64 covering K", (long long)ldw);
    FIE_REQUIRE(act == FIE_ACT_NONE || (act == FIE_ACT_GEGLU && N % 320 == 0), "fie_gemm_ln_f16: act %d (none, or GEGLU with N %% 320 == 0)", act);
    FIE_REQUIRE(eps > 0.f, "fie_gemm_ln_f16: eps must be positive");
    if (int e = check_epilogue("fie_gemm_ln_f16", N, ldc, nullptr, 0, act)) return e;
    GemmArgs a = {};
    a.A1 = (const half_t*)X; a.lda1 = ldx; a.K1 = K;
    a.Wt = (const half_t*)Wfolded; a.ldw = ldw; a.C = (half_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.rows_per_batch = 1; a.scale = 1.f; a.act = act;
    a.a1_bytes = ((int64_t)(M - 1) * ldx + K) * 2;
    a.w_bytes = fie_roundup(N, 128) * ldw * 2;
    a.ln_tab = ln_tab; a.ln_eps = eps;
    return launch<0>(ctx, a);
}

int fie_gemm_w8_f16(fie_ctx* ctx, const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2, const void* W8packed,
                    int64_t ldw, const float* w_scale, void* C, int64_t ldc, int M, int N, int K, const void* bias, const void* rowbias,
                    int64_t ld_rowbias, int rows_per_batch, const void* residual, int64_t ldr, float scale, int act) {
    FIE_REQUIRE(w_scale != nullptr, "fie_gemm_w8_f16: w_scale is NULL");
    return gemm_impl("fie_gemm_w8_f16", ctx, A1, lda1, K1, A2, lda2, W8packed, ldw, w_scale, C, ldc, M, N, K, bias, rowbias, ld_rowbias,
                     rows_per_batch, residual, ldr, scale, act);
}

// e4m3 activations x e4m3 weights (gemm_x8.hip).  A8: [M, K] e4m3 bytes written by a producer with out_f8 (fie_layernorm_f16_o8,
// fie_attention_f16_o8, this entry with out_f8, fie_quantize_f8), row stride lda BYTES; W8packed / w_scale as fie_gemm_w8_f16 with ldw % 128 == 0.
// C = epi(a_scale * w_scale[n] * (A8 . W8^T)): f16 [M, N] (ldc in elements) or, out_f8 != 0, e4m3 bytes of value * out_inv_scale (ldc in bytes).
int fie_gemm_x8_f16(fie_ctx* ctx, const void* A8, int64_t lda, const void* W8packed, int64_t ldw, const float* w_scale, float a_scale, void* C, int64_t ldc,
                    int M, int N, int K, const void* bias, const void* rowbias, int64_t ld_rowbias, int rows_per_batch, const void* residual, int64_t ldr,
                    float scale, int act, int out_f8, float out_inv_scale) {
    const char* who = "fie_gemm_x8_f16";
    (void)grab_gn_target(ctx);
    FIE_REQUIRE(ctx && A8 && W8packed && w_scale && C, "%s: NULL ctx/A/W/scale/C", who);
    FIE_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0 && lda % 16 == 0 && lda >= K, "%s: bad shape M=%d N=%d K=%d lda=%lld (K, lda %% 16 == 0)", who, M, N, K, (long long)lda);
    FIE_REQUIRE(ldw % 128 == 0 && ldw >= K, "%s: ldw=%lld must be a multiple of 128 covering K", who, (long long)ldw);
    FIE_REQUIRE(a_scale > 0.f && (!out_f8 || out_inv_scale > 0.f), "%s: scales must be positive", who);
    FIE_REQUIRE(!rowbias || rows_per_batch > 0, "%s: rowbias needs rows_per_batch", who);
    if (int e = check_epilogue(who, N, out_f8 ? 4 : ldc, residual, ldr, act)) return e;
    FIE_REQUIRE(!out_f8 || (ldc % 4 == 0 && !residual), "%s: fp8 output: ldc %% 4 == 0, no residual", who);
    GemmArgs a = {};
    a.A1 = (const half_t*)A8; a.lda1 = lda; a.K1 = K;
    a.Wt = (const half_t*)W8packed; a.ldw = ldw; a.w_scale = w_scale; a.a_scale = a_scale; a.C = (half_t*)C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.bias = (const half_t*)bias; a.rowbias = (const half_t*)rowbias; a.ld_rowbias = ld_rowbias;
    a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    a.res = (const half_t*)residual; a.ldr = ldr; a.scale = scale; a.act = act;
    a.out_f8 = out_f8; a.out_inv_scale = out_inv_scale;
    a.a1_bytes = (int64_t)(M - 1) * lda + K;
    a.w_bytes = fie_roundup(N, 128) * ldw;
    return launch<0>(ctx, a);
}

static int conv_impl(const char* who, fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, int upsample2x, int stride, int pad_mode,
                     const void* Wpacked, int64_t ldw, const float* w_scale, void* Y, int64_t ldc, int Cout, const void* bias,
                     const void* rowbias, int64_t ld_rowbias, const void* residual, int64_t ldr, float scale, int act,
                     const void* X2 = nullptr, int64_t ld2 = 0, int C2 = 0, const void* X3 = nullptr, int64_t ld3 = 0, int C3 = 0,
                     const float* gna_tab = nullptr, int gna_silu = 0) {
    const GnTarget gn = grab_gn_target(ctx);
    FIE_REQUIRE(ctx && X && Wpacked && Y, "%s: NULL ctx/X/W/Y", who);
    FIE_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "%s: bad shape", who);
    FIE_REQUIRE(Cin % 8 == 0, "%s: Cin=%d must be a multiple of 8 (pad the tensor)", who, Cin);
    FIE_REQUIRE(stride == 1 || stride == 2, "%s: stride %d", who, stride);
    FIE_REQUIRE(pad_mode == 0 || pad_mode == 1, "%s: pad_mode %d", who, pad_mode);
    FIE_REQUIRE(act != FIE_ACT_GEGLU, "%s: GEGLU not supported", who);
    FIE_REQUIRE((X2 != nullptr) == (C2 > 0) && (X3 != nullptr) == (C3 > 0) && (!X3 || X2) && C2 % BK == 0 && C3 % BK == 0 && (!X2 || (Cin % BK == 0 && !w_scale && ld2 >= C2 && (!X3 || ld3 >= C3))),
                "%s: side inputs need Cin, C2, C3 %% 64 == 0 (C2=%d C3=%d)", who, C2, C3);
    const int K = 9 * Cin + C2 + C3;
    FIE_REQUIRE(ldw % BK == 0 && ldw >= K, "%s: ldw=%lld must be a multiple of 64 covering 9*Cin (+ side inputs)", who, (long long)ldw);
    if (int e = check_epilogue(who, Cout, ldc, residual, ldr, act)) return e;
    const int ups = upsample2x ? 1 : 0;
    const int Hin = H << ups, Win = W << ups;
    const int pads = pad_mode == 0 ? 2 : 1;
    const int OH = (Hin + pads - 3) / stride + 1, OW = (Win + pads - 3) / stride + 1;
    FIE_REQUIRE((int64_t)B * OH * OW < (1ll << 31), "%s: too many output pixels", who);
    GemmArgs a = {};
    a.A1 = (const half_t*)X; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.stride = stride;
    a.pt = a.pl = pad_mode == 0 ? 1 : 0; a.ups = ups;
    a.Wt = (const half_t*)Wpacked; a.ldw = ldw; a.w_scale = w_scale; a.C = (half_t*)Y; a.ldc = ldc;
    a.M = B * OH * OW; a.N = Cout; a.K = K; a.K1 = K;
    a.bias = (const half_t*)bias; a.rowbias = (const half_t*)rowbias; a.ld_rowbias = ld_rowbias;
    a.rows_per_batch = OH * OW; a.res = (const half_t*)residual; a.ldr = ldr; a.scale = scale; a.act = act;
    a.a1_bytes = (int64_t)B * H * W * Cin * 2;
    a.a2_bytes = 0;
    if (X2) {                                               // 1x1 side inputs: row m of X2 (and X3) after the nine taps
        FIE_REQUIRE(stride == 1 && !ups && pad_mode == 0, "%s: side inputs need the plain same-size conv", who);
        a.A2 = (const half_t*)X2; a.lda2 = ld2; a.C2x = C2; a.a2_bytes = ((int64_t)(a.M - 1) * ld2 + C2) * 2;
        a.A3 = (const half_t*)X3; a.lda3 = ld3; a.C3x = C3; a.a3_bytes = X3 ? ((int64_t)(a.M - 1) * ld3 + C3) * 2 : 0;
        FIE_REQUIRE(a.a2_bytes < (1ll << 31) && a.a3_bytes < (1ll << 31), "%s: side inputs too large", who);
    }
    a.w_bytes = fie_roundup(Cout, 128) * ldw * (w_scale ? 1 : 2);
    a.gna_tab = gna_tab; a.gna_silu = gna_silu;
    if (int rc = take_gn_target(who, gn, a)) return rc;
    return launch<1>(ctx, a);
}

// 3x3 conv on e4m3 activations (written by fie_groupnorm_nhwc_f16_o8 / fie_groupnorm_stats_nhwc_f16_o8) and e4m3 weights (fie_pack_conv3x3_f8 with
// cin_pad == Cin, Cin % 128 == 0): the conv view of gemm_x8.hip, a K-step = 128 channels of one tap.  Y is f16 (ldc in elements).
int fie_conv3x3_x8_nhwc_f16(fie_ctx* ctx, const void* X8, int B, int H, int W, int Cin, int upsample2x, int stride, int pad_mode, const void* W8packed,
                            int64_t ldw, const float* w_scale, float a_scale, void* Y, int64_t ldc, int Cout, const void* bias, const void* rowbias,
                            int64_t ld_rowbias, const void* residual, int64_t ldr, float scale, int act) {
    const char* who = "fie_conv3x3_x8_nhwc_f16";
    const GnTarget gn = grab_gn_target(ctx);
    FIE_REQUIRE(ctx && X8 && W8packed && w_scale && Y, "%s: NULL ctx/X/W/scale/Y", who);
    FIE_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 128 == 0 && Cout > 0 && a_scale > 0.f, "%s: bad shape (Cin %% 128 == 0)", who);
    FIE_REQUIRE((stride == 1 || stride == 2) && (pad_mode == 0 || pad_mode == 1) && act != FIE_ACT_GEGLU, "%s: stride / pad_mode / act", who);
    const int K = 9 * Cin;
    FIE_REQUIRE(ldw % 128 == 0 && ldw >= K, "%s: ldw=%lld must be a multiple of 128 covering 9*Cin", who, (long long)ldw);
    if (int e = check_epilogue(who, Cout, ldc, residual, ldr, act)) return e;
    const int ups = upsample2x ? 1 : 0;
    const int Hin = H << ups, Win = W << ups;
    const int pads = pad_mode == 0 ? 2 : 1;
    const int OH = (Hin + pads - 3) / stride + 1, OW = (Win + pads - 3) / stride + 1;
    FIE_REQUIRE((int64_t)B * OH * OW < (1ll << 31), "%s: too many output pixels", who);
    GemmArgs a = {};
    a.A1 = (const half_t*)X8; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.stride = stride;
    a.pt = a.pl = pad_mode == 0 ? 1 : 0; a.ups = ups;
    a.Wt = (const half_t*)W8packed; a.ldw = ldw; a.w_scale = w_scale; a.a_scale = a_scale; a.C = (half_t*)Y; a.ldc = ldc;
    a.M = B * OH * OW; a.N = Cout; a.K = K; a.K1 = K;
    a.bias = (const half_t*)bias; a.rowbias = (const half_t*)rowbias; a.ld_rowbias = ld_rowbias;
    a.rows_per_batch = OH * OW; a.res = (const half_t*)residual; a.ldr = ldr; a.scale = scale; a.act = act;
    a.a1_bytes = (int64_t)B * H * W * Cin;
    a.w_bytes = fie_roundup(Cout, 128) * ldw;
    if (int rc = take_gn_target(who, gn, a)) return rc;
    return launch<1>(ctx, a);
}

// GroupNorm (+ SiLU) -> conv3x3 in ONE launch: the conv reads the UN-normalised tensor and applies y = silu(x * sc[c] + sh[c]) to every halo chunk while it
// is resident in LDS (csrc/conv_halo.hip, GNA), with the per-(image, channel) coefficients of fie_groupnorm_coef_f16.  Exists for what
// fie_conv3x3_gn_ok says: one image, H, W % 16 == 0, 128 <= Cin <= 1024 in whole 64-channel chunks, Cout % 128 == 0, an armed fie_gn_stats_target
// for the output with 4 / 8 / 16 channels per group, optional bias and residual.
int fie_conv3x3_gn_ok(fie_ctx* ctx, int B, int H, int W, int Cin, int Cout, int out_groups) {
    (void)ctx;                                              // a pure function of the shape (the C++ walks ask it in their planning pass, without a context)
    if (B != 1 || H <= 0 || W <= 0 || H % 16 || W % 16 || Cin % BK || Cin < 2 * BK || Cin > 1024 || Cout % 128 || Cout / 128 > 64 || out_groups <= 0 || Cout % out_groups) return 0;
    const int cg = Cout / out_groups;
    if (cg != 4 && cg != 8 && cg != 16) return 0;
    return (int64_t)H * W * Cin * 2 < (1ll << 31) && (int64_t)H * W * Cout * 2 <= (1ll << 30) ? 1 : 0;
}

int fie_conv3x3_gn_nhwc_f16(fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, const float* coef, int silu, const void* Wpacked, int64_t ldw, void* Y,
                            int64_t ldc, int Cout, const void* bias, const void* residual, int64_t ldr) {
    FIE_REQUIRE(coef != nullptr, "fie_conv3x3_gn_nhwc_f16: NULL coefficient table");
    return conv_impl("fie_conv3x3_gn_nhwc_f16", ctx, X, B, H, W, Cin, 0, 1, 0, Wpacked, ldw, nullptr, Y, ldc, Cout, bias, nullptr, 0, residual, ldr, 1.0f,
                     FIE_ACT_NONE, nullptr, 0, 0, nullptr, 0, 0, coef, silu ? 1 : 0);
}

int fie_conv3x3_nhwc_f16(fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, int upsample2x, int stride,
                         int pad_mode, const void* Wpacked, int64_t ldw, void* Y, int64_t ldc, int Cout,
                         const void* bias, const void* rowbias, int64_t ld_rowbias, const void* residual,
                         int64_t ldr, float scale, int act) {
    return conv_impl("fie_conv3x3_nhwc_f16", ctx, X, B, H, W, Cin, upsample2x, stride, pad_mode, Wpacked, ldw, nullptr, Y, ldc, Cout, bias,
                     rowbias, ld_rowbias, residual, ldr, scale, act);
}

// conv3x3(X) + [X2 | X3] W1x1^T in one GEMM: a resnet's second conv together with its 1x1 shortcut (upstream models/resnet.py:
// conv2(h) + conv_shortcut(input)).  Wpacked rows are [9 * Cin taps | C2 | C3] wide; the separate shortcut GEMM, its [M, Cout] output and
// the residual read of it disappear, and the sum stays in fp32 until the one rounding.
int fie_conv3x3_plus_nhwc_f16(fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, const void* Wpacked, int64_t ldw, void* Y, int64_t ldc, int Cout,
                              const void* bias, const void* rowbias, int64_t ld_rowbias, float scale, int act, const void* X2, int64_t ld2, int C2,
                              const void* X3, int64_t ld3, int C3) {
    return conv_impl("fie_conv3x3_plus_nhwc_f16", ctx, X, B, H, W, Cin, 0, 1, 0, Wpacked, ldw, nullptr, Y, ldc, Cout, bias, rowbias, ld_rowbias, nullptr, 0,
                     scale, act, X2, ld2, C2, X3, ld3, C3);
}

// conv3x3(nearest-2x(X)) as four 2x2 convs on X, one per output parity: the three taps of a 3x3 window on the upsampled image fall on TWO
// input pixels per axis (parity 0: {ky 0} | {ky 1, 2}; parity 1: {ky 0, 1} | {ky 2}), so with the weights of coinciding taps summed
// beforehand (W4: [4 = py * 2 + px][Npad][ldw], K index = (a * 2 + b) * Cin + ci) 4 instead of 9 multiply-adds per output give the same
// sums -- zero padding included: an upsampled row is outside the image exactly when its input row is.  2.25x fewer FLOPs on the
// decoder's / UNet's up-sampling convs.  A parity is the ring / phased kernels' conv view with 2x2 taps (pt = 1 - py, pl = 1 - px) whose
// epilogue scatters row (b, y, x) to pixel (2y + py, 2x + px); the four parities share one launch.
int fie_conv_up2x_nhwc_f16(fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, const void* W4, int64_t ldw, int Npad, void* Y, int64_t ldc,
                           int Cout, const void* bias, const void* rowbias, int64_t ld_rowbias, float scale, int act) {
    const char* who = "fie_conv_up2x_nhwc_f16";
    const GnTarget gnt = grab_gn_target(ctx);               // GroupNorm sums of the [B, 2H, 2W, Cout] output: a quarter of the granules per parity
    FIE_REQUIRE(ctx && X && W4 && Y, "%s: NULL ctx/X/W/Y", who);
    FIE_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % BK == 0 && Cout > 0 && Npad >= Cout, "%s: bad shape (Cin %% 64 == 0)", who);
    FIE_REQUIRE(act != FIE_ACT_GEGLU, "%s: GEGLU not supported", who);
    const int K = 4 * Cin;
    FIE_REQUIRE(ldw % BK == 0 && ldw >= K, "%s: ldw=%lld must be a multiple of 64 covering 4*Cin", who, (long long)ldw);
    if (int e = check_epilogue(who, Cout, ldc, nullptr, 0, act)) return e;
    FIE_REQUIRE((int64_t)B * H * W * 4 < (1ll << 31), "%s: too many output pixels", who);
    float* gn = gnt.partial;
    const int64_t gn_rows = gnt.rows;
    const int gn_groups = gnt.groups;
    if (gn) {
        FIE_REQUIRE(gn_rows == 4ll * H * W && (H * W) % 32 == 0 && Cout % gn_groups == 0, "%s: GroupNorm statistics: %lld rows per image for a %dx%d input", who, (long long)gn_rows, H, W);
        const int cg = Cout / gn_groups;
        FIE_REQUIRE(cg == 4 || cg == 8 || cg == 16, "%s: GroupNorm statistics need 4, 8 or 16 channels per group (got %d)", who, cg);
    }
    // ONE launch for the four parities: tile id = 4 * tile + parity (gemm_common.h: take_parity), so the four parity tiles of an output tile
    // run together and share their input rows in L2, and the small up-sampler convs of the UNet fill the CUs (160 tiles per parity otherwise)
    GemmArgs a = {};
    a.A1 = (const half_t*)X; a.H = H; a.W = W; a.Cin = Cin; a.OH = H; a.OW = W; a.stride = 1;
    a.taps2 = 1; a.oscat = 2; a.pt = a.pl = 1; a.ups = 0;
    a.Wt = (const half_t*)W4; a.w_par_stride = (int64_t)Npad * ldw; a.ldw = ldw; a.C = (half_t*)Y; a.ldc = ldc;
    a.M = B * H * W; a.N = Cout; a.K = K; a.K1 = K;
    a.bias = (const half_t*)bias; a.rowbias = (const half_t*)rowbias; a.ld_rowbias = ld_rowbias;
    a.rows_per_batch = H * W; a.scale = scale; a.act = act;
    a.a1_bytes = (int64_t)B * H * W * Cin * 2;
    a.w_bytes = (int64_t)Npad * ldw * 2;                    // per parity: the kernel's descriptor starts at its own matrix
    if (gn) {
        a.gn_partial = gn; a.gn_rows = H * W; a.gn_G = gn_groups; a.gn_cg = Cout / gn_groups;
        a.gn_nch = (int)(gn_rows / 32);
    }
    const bool dma_ok = a.a1_bytes < (1ll << 31) && a.w_bytes < (1ll << 31) && ((int64_t)(4ll * a.M - 1) * ldc + Cout) * 2 <= (1ll << 30);
    FIE_REQUIRE(dma_ok, "%s: tensors too large for the LDS-DMA kernels (use fie_conv3x3_nhwc_f16 with upsample2x)", who);
    if (int rc = launch<1>(ctx, a)) return rc;
    return FIE_OK;
}

int fie_conv3x3_w8_nhwc_f16(fie_ctx* ctx, const void* X, int B, int H, int W, int Cin, int upsample2x, int stride, int pad_mode,
                            const void* W8packed, int64_t ldw, const float* w_scale, void* Y, int64_t ldc, int Cout, const void* bias,
                            const void* rowbias, int64_t ld_rowbias, const void* residual, int64_t ldr, float scale, int act) {
    FIE_REQUIRE(w_scale != nullptr, "fie_conv3x3_w8_nhwc_f16: w_scale is NULL");
    return conv_impl("fie_conv3x3_w8_nhwc_f16", ctx, X, B, H, W, Cin, upsample2x, stride, pad_mode, W8packed, ldw, w_scale, Y, ldc, Cout,
                     bias, rowbias, ld_rowbias, residual, ldr, scale, act);
}

int fie_pack_rows_f16(fie_ctx* ctx, const void* src, int64_t ld_src, int N, int K, void* dst, int64_t ldw,
                      int Npad, int interleave2) {
    FIE_REQUIRE(ctx && src && dst, "fie_pack_rows_f16: NULL argument");
    FIE_REQUIRE(N > 0 && K > 0 && Npad >= N && ldw >= K, "fie_pack_rows_f16: bad shape");
    FIE_REQUIRE(!interleave2 || N % 2 == 0, "fie_pack_rows_f16: interleave2 needs even N");
    fie_launch(ctx, pack_rows_kernel, dim3(1024), dim3(256), 0, (const half_t*)src, ld_src, N, K, (half_t*)dst, ldw, Npad, interleave2);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

int fie_pack_conv3x3_f16(fie_ctx* ctx, const void* src_oihw, int Cout, int Cin, int cin_pad, void* dst, int64_t ldw,
                         int Npad) {
    FIE_REQUIRE(ctx && src_oihw && dst, "fie_pack_conv3x3_f16: NULL argument");
    FIE_REQUIRE(cin_pad >= Cin && cin_pad % 8 == 0 && ldw >= 9 * cin_pad && Npad >= Cout, "fie_pack_conv3x3_f16: bad shape");
    fie_launch(ctx, pack_conv_kernel, dim3(1024), dim3(256), 0, (const half_t*)src_oihw, Cout, Cin, cin_pad, (half_t*)dst, ldw, Npad);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

}  // extern "C"
