// fp8 (OCP e4m3) ACTIVATIONS x fp8 WEIGHTS on the CDNA4 block-scaled MFMA -- BASELINE.json config 5, second half: with e4m3 weights alone
// (gemm_w8.hip) the activations stay fp16 and are converted per fragment on the MFMA waves, so fp8 bought half the weight bytes and nothing
// else (round 2: 2 % SLOWER than fp16).  Here the producer of an activation (LayerNorm, attention, the GEGLU epilogue) writes e4m3 bytes,
// and the GEMM runs v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (E8M0 127 = 2^0): K = 128 per instruction at twice the fp16
// MFMA rate (MI355X_MICROARCH.md, Matrix cores), and a 128-byte LDS row now holds 128 k-values instead of 64 -- half the bytes per FLOP
// through the per-CU global->LDS path that paces the M = 2048 GEMMs.  Scales: one fp32 per output channel for the weights (fie_pack_rows_f8)
// times ONE per-tensor activation scale (a_scale), both applied to the fp32 accumulator first in the epilogue.
//
// Kernel = the LDS-DMA ring kernel of gemm_conv.hip (GEMM view only) with byte elements: K-step = 128 elements = the same 128-byte rows, the
// same 16-byte-chunk XOR swizzle by (row & 7), the same DMA pieces and counted vmcnt; a lane's fragment is 32 contiguous k-values of one row
// (lane = (row 0-15, k-block 0-3): two ds_read_b128), operands swapped (weights first) so a lane owns 4 consecutive output channels.
// Entry: fie_gemm_x8_f16 (include/fie.h).
#include "gemm_common.h"

using namespace fie_gemm;

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int KE = 128;                   // k-values per K-step (one 128-byte row)
constexpr int kUnitScale = 0x7F7F7F7F;    // E8M0 127 in every byte: block scale 2^0

template <int BM, int BN, int ST, int NW, int MODE = 0, bool LEAN = false>      // MODE 0 = GEMM view, 2 = 3x3 conv view (im2col of an NHWC e4m3 tensor, Cin % 128 == 0); LEAN: gemm_common.h epilogue_lean_scaled
__global__ __launch_bounds__(NW * 64) void gemm3x8_kernel(GemmArgs p) {
    constexpr int WGN = NW / 2;
    constexpr int WM = BM / 2, WN = BN / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int RA = BM / (8 * NW), RW = BN / (8 * NW);
    constexpr int NP = RA + RW;
    constexpr int STAGE = (BM + BN) * 128;                           // bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char smx[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    const int bid_all = xcd_remap(blockIdx.x, gridDim.x);
    const int nsplit = p.splitk > 1 ? p.splitk : 1;
    const int tile_all = bid_all / nsplit, slice = bid_all - tile_all * nsplit;
    const int m0 = (p.order ? tile_all % p.nbm : tile_all / p.nbn) * BM;
    const int n0 = (p.order ? tile_all / p.nbm : tile_all % p.nbn) * BN;
    const int lr = lane >> 3;
    const int c8 = (lane & 7) ^ lr;

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)p.a1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, (int)p.w_bytes, 0x00020000);
    // one per-lane offset per operand + a uniform stride per 8-row piece group (rows >= M: past the descriptor's extent, read as zeros)
    const unsigned a_base = (unsigned)(m0 + wave * 8 + lr) * (unsigned)p.lda1 + c8 * 16u;
    const unsigned a_step = (unsigned)(NW * 8) * (unsigned)p.lda1;
    const unsigned w_base = (unsigned)(n0 + wave * 8 + lr) * (unsigned)p.ldw + c8 * 16u;
    const unsigned w_step = (unsigned)(NW * 8) * (unsigned)p.ldw;

    const int nk_all = (p.K + KE - 1) / KE;
    const int kbeg = (int)((int64_t)nk_all * slice / nsplit);
    const int nk = (int)((int64_t)nk_all * (slice + 1) / nsplit) - kbeg;
    const bool ktail = (p.K % KE) != 0;

    // conv view: per-lane pixel coordinates of this lane's row in each of its RA pieces; a K-step = 128 channels of one tap
    int a_ih[RA], a_iw[RA];
    unsigned a_img[RA], a_off[RA];
    bool a_ok[RA];
    const int csteps = MODE == 2 ? p.Cin / KE : 1;
    int ftap = kbeg / csteps, cs = kbeg - ftap * csteps;
    bool tap_fresh = true;
    if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0 + (wave + NW * i) * 8 + lr;
            a_ok[i] = m < p.M;
            const int hw = p.OH * p.OW;
            const int b = m / hw, rem = m - b * hw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_ih[i] = oh * p.stride - p.pt;
            a_iw[i] = ow * p.stride - p.pl;
            a_img[i] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.Cin;
            a_off[i] = kOob;
        }
    }

    auto issue = [&](int kt, int stage) {
        unsigned char* sa = smx + stage * STAGE;
        unsigned char* sw = sa + BM * 128;
        const unsigned so = (unsigned)kt * KE;
        if constexpr (MODE == 2) {
            if (cs == 0 || tap_fresh) {
                tap_fresh = false;
                const int ky = (ftap * 11) >> 5, kx = ftap - 3 * ky;
                const int hlim = p.H << p.ups, wlim = p.W << p.ups;
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const int ih = a_ih[i] + ky, iw = a_iw[i] + kx;
                    const bool ok = a_ok[i] && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
                    a_off[i] = ok ? a_img[i] + (unsigned)((ih >> p.ups) * p.W + (iw >> p.ups)) * (unsigned)p.Cin + c8 * 16u : kOob;
                }
            }
            const unsigned soa = (unsigned)cs * KE;
#pragma unroll
            for (int i = 0; i < RA; ++i) bload16(rs_a, reinterpret_cast<half_t*>(sa + (wave + NW * i) * 1024), a_off[i], soa);
            if (++cs == csteps) { cs = 0; ++ftap; }
        } else if (ktail && kt == nk_all - 1) {                            // last, partial K-step: k >= K reads as zero (K % 16 == 0)
            const bool in_k = kt * KE + c8 * 16 < p.K;
#pragma unroll
            for (int i = 0; i < RA; ++i) bload16(rs_a, reinterpret_cast<half_t*>(sa + (wave + NW * i) * 1024), in_k ? a_base + (unsigned)i * a_step : kOob, so);
        } else {
#pragma unroll
            for (int i = 0; i < RA; ++i) bload16(rs_a, reinterpret_cast<half_t*>(sa + (wave + NW * i) * 1024), a_base + (unsigned)i * a_step, so);
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) bload16(rs_w, reinterpret_cast<half_t*>(sw + (wave + NW * i) * 1024), w_base + (unsigned)i * w_step, so);
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // bias row + residual tile ahead of the K loop, as in the f16 ring kernels (gemm_common.h: EpiPre)
    EpiPre<FM, FN> pre;
    pre.on = false;
    if constexpr (FM * FN <= 16) {
        if (LEAN || (nsplit == 1 && p.epi_prefetch)) epilogue_prefetch<FM, FN, WM, WN>(p, pre, m0, n0, wm, wn, lane);
    }
    f32x4 ws[LEAN ? FN : 1];                                     // LEAN: the dequantisation scales of the lane's columns ride along with the bias row
    if constexpr (LEAN) {
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w_scale), 0, p.N * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < FN; ++i) ws[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (unsigned)(n0 + wn * WN + i * 16 + (lane >> 4) * 4) * 4u, 0, 0));
    }

#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s < nk) issue(kbeg + s, s);

    const int fr = lane & 15, fq = lane >> 4;
    // fragment = k-block fq of row r: 16-byte chunks 2 fq and 2 fq + 1, each at its swizzled place (row & 7 == fr & 7 for every fragment row)
    const int ch0 = ((2 * fq) ^ (fr & 7)) << 4, ch1 = ((2 * fq + 1) ^ (fr & 7)) << 4;
    auto frag = [&](const unsigned char* base, int row) {
        const v4i lo = *reinterpret_cast<const v4i*>(base + row * 128 + ch0), hi = *reinterpret_cast<const v4i*>(base + row * 128 + ch1);
        return (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    int stage = 0, fill = ST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const int later = min(kt + ST - 2, nk - 1) - kt;
        if (ST >= 3 && later >= 1) wait_vm_barrier<NP>(); else wait_vm_barrier<0>();
        if (kt + ST - 1 < nk) issue(kbeg + kt + ST - 1, fill);
        const unsigned char* sa = smx + stage * STAGE;
        const unsigned char* sw = sa + BM * 128;
        v8i fw[FN];
#pragma unroll
        for (int i = 0; i < FN; ++i) fw[i] = frag(sw, wn * WN + i * 16 + fr);
        constexpr int HM = FM * FN > 16 ? FM / 2 : FM;               // big tiles: the activation fragments in two halves (register budget)
#pragma unroll
        for (int jh = 0; jh < FM / HM; ++jh) {
            v8i fa[HM];
#pragma unroll
            for (int j = 0; j < HM; ++j) fa[j] = frag(sa, wm * WM + (jh * HM + j) * 16 + fr);
            if (NW == 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < HM; ++j)
                    acc[i][jh * HM + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[i], fa[j], acc[i][jh * HM + j], 0, 0, 0, kUnitScale, 0, kUnitScale);
            if (NW == 8) __builtin_amdgcn_s_setprio(0);
        }
        stage = stage + 1 == ST ? 0 : stage + 1;
        fill = fill + 1 == ST ? 0 : fill + 1;
    }
    if constexpr (FM * FN <= 16 && !LEAN) {
        if (nsplit > 1 && !splitk_reduce<FM, FN, NW>(p, acc, tile_all, slice, tid, smx)) return;
    }
    if constexpr (FM * FN > 16) {
        // column chunks of two fragments, the chunk index a COMPILE-TIME constant: as a `#pragma unroll` loop the two huge inlined bodies exceed the
        // pragma-unroll threshold as soon as the epilogue grows by a few instructions, the loop stays rolled, acc[2 * c] becomes a runtime index and the
        // whole accumulator array moves to scratch -- K loop included (round 4: FF1 on tile 64 went from 55 to 87 us that way)
        static_for([&](auto cc) {
            constexpr int C = decltype(cc)::value;
            epilogue<FM, 2, WM, WN, true>(p, reinterpret_cast<f32x4(&)[2][FM]>(acc[2 * C]), m0, n0 + 32 * C, wm, wn, lane);
        }, std::make_integer_sequence<int, FN / 2>{});
        if constexpr (FN & 1)
            epilogue<FM, 1, WM, WN, true>(p, reinterpret_cast<f32x4(&)[1][FM]>(acc[FN - 1]), m0, n0 + 16 * (FN - 1), wm, wn, lane);
    } else {
        if constexpr (LEAN) epilogue_lean_scaled<FM, FN, WM, WN>(p, acc, m0, n0, wm, wn, lane, pre, ws);
        else epilogue<FM, FN, WM, WN, true>(p, acc, m0, n0, wm, wn, lane, &pre);
    }
}

template <int BM, int BN, int ST>
constexpr int x8_lds() { return ST * (BM + BN) * 128; }

template <int BM, int BN, int NW>
constexpr bool x8_has_lean() { return BM * BN / (NW * 64) <= 64; }

template <int BM, int BN, int ST, int NW>
hipError_t x8_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3x8_kernel<BM, BN, ST, NW, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, x8_lds<BM, BN, ST>());
    if constexpr (x8_has_lean<BM, BN, NW>()) {
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3x8_kernel<BM, BN, ST, NW, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, x8_lds<BM, BN, ST>());
    }
    if (e == hipSuccess && BN != 320)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3x8_kernel<BM, BN, ST, NW, BN == 320 ? 0 : 2>), hipFuncAttributeMaxDynamicSharedMemorySize, x8_lds<BM, BN, ST>());
    return e;
}

template <int BM, int BN, int ST, int NW>
void launch_x8(fie_ctx* ctx, const GemmArgs& a, dim3 grid, int conv) {
    if (conv) fie_launch(ctx, (gemm3x8_kernel<BM, BN, ST, NW, BN == 320 ? 0 : 2>), grid, dim3(NW * 64), x8_lds<BM, BN, ST>(), a);
    else {
        if constexpr (x8_has_lean<BM, BN, NW>()) {
            // a plain Linear on e4m3 activations (optional bias / residual, f16 out): the kernel with the lean epilogue
            if (a.epi_prefetch && a.splitk <= 1 && !a.rowbias && a.act == FIE_ACT_NONE && a.scale == 1.f && !a.gn_partial && !a.out_f8 && !a.oscat && a.probe == 0) {
                fie_launch(ctx, (gemm3x8_kernel<BM, BN, ST, NW, 0, true>), grid, dim3(NW * 64), x8_lds<BM, BN, ST>(), a);
                return;
            }
        }
        fie_launch(ctx, (gemm3x8_kernel<BM, BN, ST, NW, 0>), grid, dim3(NW * 64), x8_lds<BM, BN, ST>(), a);
    }
}

// ---- e4m3 conversion of an fp16 activation tensor (tests, and producers that have no fused form): y = sat(x * inv_scale)
__global__ void quant_f8_kernel(const half_t* x, int64_t ldx, unsigned char* y, int64_t ldy, int64_t rows, int C, float inv) {
    const int nch = C >> 3;
    const int64_t total = rows * nch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nch;
        const int c = (int)(i - r * nch) * 8;
        float v[8];
        fie_load8(x + r * ldx + c, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fie_sat448(v[j] * inv);
        int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
        int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
        *reinterpret_cast<int2*>(y + r * ldy + c) = make_int2(lo, hi);
    }
}

}  // namespace

int fie_gemm_x8_init(void) {
    hipError_t e = x8_attr<128, 64, 3, 4>();
    if (e == hipSuccess) e = x8_attr<64, 64, 3, 4>();
    if (e == hipSuccess) e = x8_attr<128, 96, 3, 4>();
    if (e == hipSuccess) e = x8_attr<128, 128, 3, 8>();
    if (e == hipSuccess) e = x8_attr<128, 128, 2, 8>();
    if (e == hipSuccess) e = x8_attr<192, 128, 2, 8>();
    if (e == hipSuccess) e = x8_attr<256, 128, 3, 8>();
    if (e == hipSuccess) e = x8_attr<256, 320, 2, 8>();
    if (e != hipSuccess) {
        fie_set_error("gemm_x8: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return FIE_EHIP;
    }
    return FIE_OK;
}

// code: 42 / 43 / 47 (128x64 / 64x64 / 128x96, 4 waves, 3 stages), 51 / 62 (128x128 / 256x128, 8 waves, 3 stages), 52 / 54 (128x128 / 192x128, 2 stages),
// 63 (256x320, 2 stages); the grid carries the split-K factor (a.splitk)
int fie_launch_gemm_x8(fie_ctx* ctx, const GemmArgs& a, int code, int conv) {
    const dim3 grid((unsigned)(a.nbm * a.nbn * (a.splitk > 1 ? a.splitk : 1)));
    if (conv && code == 63) { fie_set_error("fie_launch_gemm_x8: tile code 63 is built for the GEMM view only"); return FIE_EINVAL; }
    switch (code) {
        case 42: launch_x8<128, 64, 3, 4>(ctx, a, grid, conv); break;
        case 43: launch_x8<64, 64, 3, 4>(ctx, a, grid, conv); break;
        case 47: launch_x8<128, 96, 3, 4>(ctx, a, grid, conv); break;
        case 51: launch_x8<128, 128, 3, 8>(ctx, a, grid, conv); break;
        case 52: launch_x8<128, 128, 2, 8>(ctx, a, grid, conv); break;
        case 54: launch_x8<192, 128, 2, 8>(ctx, a, grid, conv); break;
        case 62: launch_x8<256, 128, 3, 8>(ctx, a, grid, conv); break;
        case 63: launch_x8<256, 320, 2, 8>(ctx, a, grid, 0); break;
        default: fie_set_error("fie_launch_gemm_x8: tile code %d not built", code); return FIE_EINVAL;
    }
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

extern "C" int fie_quantize_f8(fie_ctx* ctx, const void* X, int64_t ldx, void* Y8, int64_t ldy, int64_t rows, int C, float inv_scale) {
    FIE_REQUIRE(ctx && X && Y8 && rows > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, "fie_quantize_f8: bad argument");
    const int64_t total = rows * (C >> 3);
    fie_launch(ctx, quant_f8_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, (const half_t*)X, ldx,
               (unsigned char*)Y8, ldy, rows, C, inv_scale);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}
