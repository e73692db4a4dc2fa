// Thin 3x3 convs (tile code 77): at most 16 output channels -- the VAE decoder's conv_out (128 -> 3 written as 4 at 1024^2), the encoder's conv_out
// (512 -> 8), the UNet's conv_out (320 -> 4) and the first convs of the ControlNet's conditioning embedding (3 -> 16 -> 16 at 1024^2): upstream
// models/autoencoders/vae.py (Decoder.conv_out / Encoder.conv_out), models/unets/unet_2d_condition.py (conv_out), models/controlnets/controlnet.py
// (ControlNetConditioningEmbedding), reached from /root/reference/src/pipeline.py:261-272.
//
// These are HBM-bound by their one large tensor (conv_out of the decoder: 268 MB of input for 8 MB of output), but on the im2col tiles they ran
// as 128x64 GEMM tiles with 4 of 64 columns in use: every input pixel staged nine times through the LDS-DMA path, 238 us where the input read
// alone is ~55 us.  Here a wave owns a 16-pixel-wide strip of ROWS output rows and ALL output channels: the MFMA is issued transposed
// (A = 16 weight rows, B = 16 pixels) so that a lane ends up with 4 consecutive channels of one pixel (one 8-byte store, 128 contiguous bytes per
// 16 pixels at ldc = 4), operands go global -> VGPR directly (no LDS, no barrier: nothing is shared between waves but cache lines), and in the
// REUSE form (stride 1, pad 1, Cin % 32 == 0) an input fragment loaded once feeds all nine taps: the three vertical ones of three output rows, the
// horizontal ones as one-lane DPP shifts of it, so a wave requests every input byte of its (ROWS + 2) x 18 pixel halo exactly once.  The general form (any Cin % 8 == 0, stride 1 / 2, either padding) walks
// the im2col K axis in steps of 32: a lane's 8 k-values never straddle a tap because Cin % 8 == 0.
// Sums: fp32 accumulators, K order differs from the im2col tiles (last-bit differences in f16 against them; deterministic).
#include "fie_internal.h"
#include "gemm_common.h"

using namespace fie_gemm;

namespace {

__device__ __forceinline__ f16x8 ld8_or_zero(const half_t* p, bool ok) {
    f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return ok ? *reinterpret_cast<const f16x8*>(p) : z;
}

// block = 4 waves stacked in y: a 16 x (4 ROWS) output patch of one image.  grid = B * ceil(OH / (4 ROWS)) * ceil(OW / 16), remapped so that
// consecutive patches (which share halo columns / rows) run on one XCD.
template <int ROWS, bool REUSE>
__global__ __launch_bounds__(256) void conv_thin_kernel(const GemmArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, kq = lane >> 4;
    const int tiles_x = (p.OW + 15) >> 4, tiles_y = (p.OH + 4 * ROWS - 1) / (4 * ROWS);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / (tiles_x * tiles_y);
    bid -= b * tiles_x * tiles_y;
    const int ty = bid / tiles_x, tx = bid - ty * tiles_x;
    const int ox = tx * 16 + px;
    const int oy0 = ty * 4 * ROWS + wave * ROWS;
    const half_t* X = p.A1 + (int64_t)b * p.H * p.W * p.Cin;
    const half_t* wrow = p.Wt + (int64_t)px * p.ldw + kq * 8;         // weight row = output channel px (rows past N are zero in the packed matrix)
    f32x4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (REUSE) {
        // stride 1, pad 1: input row iy = oy + ky - 1, input column ix = ox + kx - 1.  Per 32-channel chunk a lane loads ITS pixel of the ROWS + 2 input
        // rows once (xc) and the strip's two halo pixels ride in lanes px == 0 (column x0 - 1) and px == 15 (column x0 + 16) of a second fragment (xe);
        // the kx = 0 / 2 operands are the centre fragment shifted by one lane within its DPP row (px = lane & 15 IS the row position; the lane with
        // no in-row source keeps `old` = the halo pixel).  Every input byte is requested once per wave: with kx as a loop around the loads the same
        // lines came back three times, a working set apart that no cache level holds (first form: 204 us on the decoder's conv_out).
        const int nch = p.Cin >> 5;
        const int x0 = tx * 16, ex = px == 0 ? x0 - 1 : x0 + 16;
        const bool vc = ox < p.W, ve = (px == 0 || px == 15) && ex >= 0 && ex < p.W;
        const half_t* xcen = X + (int64_t)(vc ? ox : 0) * p.Cin + kq * 8;
        const half_t* xedg = X + (int64_t)(ve ? ex : 0) * p.Cin + kq * 8;
        const int64_t row_ld = (int64_t)p.W * p.Cin;
        const u32x4 zero4 = {0u, 0u, 0u, 0u};
        for (int c = 0; c < nch; ++c) {
            f16x8 w[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t] = ld8_or_zero(wrow + t * p.Cin + c * 32, px < p.N);
            u32x4 xa[ROWS + 2], xb[ROWS + 2];
#pragma unroll
            for (int ri = 0; ri < ROWS + 2; ++ri) {
                const int iy = oy0 + ri - 1;
                const int64_t ro = (int64_t)(iy >= 0 && iy < p.H ? iy : 0) * row_ld + c * 32;       // clamped: always in range, zeroed below
                xa[ri] = *reinterpret_cast<const u32x4*>(xcen + ro);
                xb[ri] = *reinterpret_cast<const u32x4*>(xedg + ro);
            }
            __builtin_amdgcn_sched_barrier(0);      // every load of the chunk in flight before the first MFMA (left alone the scheduler sinks them between the MFMAs, six at a time behind vmcnt(0) waits)
#pragma unroll
            for (int ri = 0; ri < ROWS + 2; ++ri) {
                const int iy = oy0 + ri - 1;
                const bool rv = iy >= 0 && iy < p.H;                        // wave-uniform
                const u32x4 xc = rv && vc ? xa[ri] : zero4, xe = rv && ve ? xb[ri] : zero4;
                u32x4 xl, xr;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    xl[d] = (unsigned)__builtin_amdgcn_update_dpp((int)xe[d], (int)xc[d], 0x111, 0xF, 0xF, false);    // row_shr:1: lane px <- px - 1
                    xr[d] = (unsigned)__builtin_amdgcn_update_dpp((int)xe[d], (int)xc[d], 0x101, 0xF, 0xF, false);    // row_shl:1: lane px <- px + 1
                }
                const f16x8 fl = __builtin_bit_cast(f16x8, xl), fc = __builtin_bit_cast(f16x8, xc), fr = __builtin_bit_cast(f16x8, xr);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int r = ri - ky;                                  // output row fed by input row ri through tap row ky
                    if (r >= 0 && r < ROWS) {
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ky * 3 + 0], fl, acc[r], 0, 0, 0);
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ky * 3 + 1], fc, acc[r], 0, 0, 0);
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ky * 3 + 2], fr, acc[r], 0, 0, 0);
                    }
                }
            }
        }
    } else {
        const int nk = (p.K + 31) >> 5;
        int tap = (kq * 8) / p.Cin, ci = kq * 8 - tap * p.Cin;
        for (int s = 0; s < nk; ++s) {
            const f16x8 w = *reinterpret_cast<const f16x8*>(wrow + s * 32);      // k < ldw always (ldw % 64 == 0 covers K); zero past K
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int ix = ox * p.stride + kx - p.pl;
            const bool vx = tap < 9 && ix >= 0 && ix < p.W;
            f16x8 x[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int iy = (oy0 + r) * p.stride + ky - p.pt;
                x[r] = ld8_or_zero(X + ((int64_t)iy * p.W + ix) * p.Cin + ci, vx && iy >= 0 && iy < p.H);
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x[r], acc[r], 0, 0, 0);
            ci += 32;
            while (ci >= p.Cin) { ci -= p.Cin; ++tap; }
        }
    }

    // lane: channels kq * 4 .. + 3 of pixel (b, oy0 + r, ox).  Epilogue order as gemm_common.h: bias, activation, scale.
    const int n0 = kq * 4;
    if (n0 >= p.N || ox >= p.OW) return;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        const f16x4 b4 = *reinterpret_cast<const f16x4*>(p.bias + n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = (float)b4[i];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int oy = oy0 + r;
        if (oy >= p.OH) break;
        f16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = acc[r][i] + bv[i];
            if (p.act == FIE_ACT_SILU) v = fie_silu(v);
            v *= p.scale;
            o[i] = (half_t)v;
        }
        *reinterpret_cast<f16x4*>(p.C + ((int64_t)(b * p.OH + oy) * p.OW + ox) * p.ldc + n0) = o;
    }
}

}  // namespace

bool fie_conv_thin_ok(const GemmArgs& a) {
    return a.N <= 16 && a.N % 4 == 0 && a.ldc % 4 == 0 && a.Cin % 8 == 0 && !a.ups && !a.taps2 && !a.oscat && !a.A2 && !a.w_scale && !a.rowbias && !a.res &&
           !a.gn_partial && !a.gna_tab && !a.ln_tab && !a.out_f8 && (a.act == FIE_ACT_NONE || a.act == FIE_ACT_SILU) && (a.stride == 1 || a.stride == 2) &&
           a.ldw >= fie_roundup(a.K, 32) && a.w_bytes >= 16 * a.ldw * 2;
}

int fie_launch_conv_thin(fie_ctx* ctx, GemmArgs& a) {
    FIE_REQUIRE(fie_conv_thin_ok(a), "thin conv (tile code 77): Cout <= 16, plain bias / SiLU epilogue, f16 weights, no side inputs only");
    const int B = a.M / (a.OH * a.OW);
    const bool reuse = a.stride == 1 && a.Cin % 32 == 0 && a.pt == 1 && a.pl == 1;
    const int tiles_x = (a.OW + 15) / 16;
    // 8 rows per wave while that still gives every CU a few blocks, else 2 (small maps).  4 rows per wave (3 waves per SIMD instead of 2) measured
    // slower on the decoder's conv_out: 105 against 92 us (profiles/r04_conv_thin.md)
    const int64_t blocks8 = (int64_t)B * ((a.OH + 31) / 32) * tiles_x;
    const int rows = blocks8 >= 4 * (int64_t)ctx->num_cus ? 8 : 2;
    const int64_t grid = (int64_t)B * ((a.OH + 4 * rows - 1) / (4 * rows)) * tiles_x;
    FIE_REQUIRE(grid < (1ll << 31), "thin conv: grid too large");
    const dim3 g((unsigned)grid), blk(256);
    if (reuse) {
        if (rows == 8) fie_launch(ctx, (conv_thin_kernel<8, true>), g, blk, 0, a);
        else fie_launch(ctx, (conv_thin_kernel<2, true>), g, blk, 0, a);
    } else {
        if (rows == 8) fie_launch(ctx, (conv_thin_kernel<8, false>), g, blk, 0, a);
        else fie_launch(ctx, (conv_thin_kernel<2, false>), g, blk, 0, a);
    }
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}
