// fp8 (OCP e4m3) WEIGHTS on the CDNA4 fp8 MFMA -- BASELINE.json config 5 ("SDXL fp8 weights on CDNA4 fp8 MFMA", the
// low-precision stretch; /root/reference/src/pipeline.py:67-71 is where the reference picks its precision).  Entries:
// fie_gemm_w8_f16 / fie_conv3x3_w8_nhwc_f16 / fie_pack_*_f8 (include/fie.h).
//
//   * weights: e4m3 with ONE fp32 scale per output channel (scale[n] = max_k |W[n,k]| / 448), packed [Npad][Kpad] bytes, K contiguous;
//     the scale is applied to the fp32 accumulator in the epilogue, in front of bias / activation / residual;
//   * activations stay fp16 in HBM and in LDS; the wave converts each 16 x 32 activation fragment to e4m3 in registers
//     (clamp to +-448, then v_cvt_scalef32_pk_fp8_f16 at scale 1: round-to-nearest-even; the instruction itself does not saturate) right before
//     v_mfma_f32_16x16x32_fp8_fp8 (fp32 accumulate).  Non-scaled fp8 MFMA runs at the fp16 rate on gfx950
//     (MI355X_MICROARCH.md, Matrix cores): what fp8 weights buy is half the weight bytes in HBM and half the weight pieces through the
//     per-CU global->LDS path that paces these kernels.
//
// Kernel = the LDS-DMA ring kernel of gemm_conv.hip (ST stages, counted vmcnt, one raw barrier per K-step, swapped operands) with a
// 64-byte-per-row weight image: a DMA piece is 16 rows x 64 B, 16-byte chunks XOR-swizzled by (row >> 2) & 3 (rows r and r + 4 would
// otherwise share banks for the ds_read_b64 fragment reads), applied on the SOURCE chunk and on the read.
#include "gemm_common.h"

using namespace fie_gemm;

namespace {

typedef short s16x2 __attribute__((ext_vector_type(2)));

// v_cvt_scalef32_pk_fp8_f16 does NOT saturate: a magnitude beyond 448 converts to the e4m3 NaN (measured, tests/test_fp8_gpu.py
// ::test_fp8_activations_beyond_e4m3_range_saturate; round 2 shipped it unclamped).  Two packed f16 min / max per pair clamp first.
__device__ __forceinline__ f16x2 clamp448(f16x2 x) {
    const f16x2 hi = {(half_t)448.f, (half_t)448.f}, lo = {(half_t)-448.f, (half_t)-448.f};
    return __builtin_elementwise_max(__builtin_elementwise_min(x, hi), lo);
}

__device__ __forceinline__ long f16x8_to_fp8x8(const f16x8& v) {
    s16x2 lo = {0, 0}, hi = {0, 0};
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, clamp448((f16x2){v[0], v[1]}), 1.0f, false);
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, clamp448((f16x2){v[2], v[3]}), 1.0f, true);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, clamp448((f16x2){v[4], v[5]}), 1.0f, false);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, clamp448((f16x2){v[6], v[7]}), 1.0f, true);
    return (long)(unsigned)__builtin_bit_cast(int, lo) | ((long)__builtin_bit_cast(int, hi) << 32);
}

template <int BM, int BN, int ST, int MODE, int NW>    // MODE 0 = GEMM, 2 = conv with Cin % 64 == 0
__global__ __launch_bounds__(NW * 64) void gemm3w8_kernel(GemmArgs p) {
    constexpr int WGN = NW / 2;
    constexpr int WM = BM / 2, WN = BN / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int RA = BM / (8 * NW), RW = BN / (16 * NW);
    static_assert(RW >= 1, "a weight piece is 16 rows");
    constexpr int NP = RA + RW;
    constexpr int STAGE_B = BM * 128 + BN * 64;                     // bytes per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (p.order ? bid % p.nbm : bid / p.nbn) * BM;
    const int n0 = (p.order ? bid / p.nbm : bid % p.nbn) * BN;
    const int lr = lane >> 3;
    const int c8 = (lane & 7) ^ lr;
    const int lr16 = lane >> 2;                                     // weight piece: 16 rows x 4 chunks of 16 B
    const int c4 = (lane & 3) ^ ((lr16 >> 2) & 3);

    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)p.a1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, (int)p.a2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, (int)p.w_bytes, 0x00020000);

    unsigned a_off1[RA], a_off2[RA];
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
    unsigned w_off[RW];                                             // ldw counts BYTES here (one per element)
#pragma unroll
    for (int i = 0; i < RW; ++i) w_off[i] = (unsigned)(n0 + (wave + NW * i) * 16 + lr16) * (unsigned)p.ldw + c4 * 16u;

    int cs = 0, ftap = 0;
    const int csteps = MODE == 2 ? p.Cin / BK : 1;
    const int k1_steps = p.K1 / BK;
    const int nk = (p.K + BK - 1) / BK;
    const bool ktail = (p.K % BK) != 0;

    auto issue = [&](int kt, int stage) {
        half_t* sa = reinterpret_cast<half_t*>(smem8 + stage * STAGE_B);
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
        } else if (ktail && kt == nk - 1) {
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
        unsigned char* sw = smem8 + stage * STAGE_B + BM * 128;
#pragma unroll
        for (int i = 0; i < RW; ++i)                                 // weights past Kpad never occur (Kpad % 64 == 0); rows >= Npad read zeros
            bload16(rs_w, reinterpret_cast<half_t*>(sw + (wave + NW * i) * 1024), w_off[i], (unsigned)kt * BK);
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s < nk) issue(s, s);

    const int fr = lane & 15, fq = lane >> 4;
    const int wsw = (fr >> 2) & 3;                                  // (row >> 2) & 3 of every weight fragment row (other terms are multiples of 16)
    int stage = 0, fill = ST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const int later = min(kt + ST - 2, nk - 1) - kt;
        if (ST >= 3 && later >= 1) wait_vm_barrier<NP>(); else wait_vm_barrier<0>();
        if (kt + ST - 1 < nk) issue(kt + ST - 1, fill);
        const half_t* sa = reinterpret_cast<const half_t*>(smem8 + stage * STAGE_B);
        const unsigned char* sw = smem8 + stage * STAGE_B + BM * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            long fw[FN], fb[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i)
                fw[i] = *reinterpret_cast<const long*>(sw + (wn * WN + i * 16 + fr) * 64 + (((kk * 2 + (fq >> 1)) ^ wsw) << 4) + (fq & 1) * 8);
#pragma unroll
            for (int j = 0; j < FM; ++j)
                fb[j] = f16x8_to_fp8x8(*reinterpret_cast<const f16x8*>(sa + lds_off(wm * WM + j * 16 + fr, kk * 4 + fq)));
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fw[i], fb[j], acc[i][j], 0, 0, 0);
        }
        stage = stage + 1 == ST ? 0 : stage + 1;
        fill = fill + 1 == ST ? 0 : fill + 1;
    }
    epilogue<FM, FN, WM, WN, true>(p, acc, m0, n0, wm, wn, lane);       // p.w_scale: acc *= scale[n] first
}

template <int BM, int BN, int ST>
constexpr int w8_lds() { return ST * (BM * 128 + BN * 64); }

// ---- weight quantisation (one-time, on device): one 256-thread block per packed row
template <bool CONV>
__global__ __launch_bounds__(256) void pack_f8_kernel(const half_t* src, int64_t ld_src, int N, int K, int Cin, int cin_pad, unsigned char* dst,
                                                       int64_t ldw, float* scales, int interleave2) {
    const int n = blockIdx.x;
    __shared__ float red[4];
    auto at = [&](int k) -> float {                                   // packed column k of packed row n, as fp32
        if (n >= N) return 0.f;
        if (CONV) {
            const int tap = k / cin_pad, ci = k - tap * cin_pad;
            return (tap < 9 && ci < Cin) ? (float)src[((int64_t)n * Cin + ci) * 9 + tap] : 0.f;
        }
        const int sn = interleave2 ? ((n & 1) ? (N / 2 + (n >> 1)) : (n >> 1)) : n;
        return k < K ? (float)src[(int64_t)sn * ld_src + k] : 0.f;
    };
    const int kcols = CONV ? 9 * cin_pad : K;
    float amax = 0.f;
    for (int k = threadIdx.x; k < kcols; k += 256) amax = fmaxf(amax, fabsf(at(k)));
#pragma unroll
    for (int o = 32; o; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;            // 448 = largest finite e4m3
    const float inv = 1.0f / scale;
    if (threadIdx.x == 0) scales[n] = scale;
    for (int64_t k2 = threadIdx.x; k2 < ldw / 2; k2 += 256) {
        const int k = (int)k2 * 2;
        const float a = k < kcols ? at(k) * inv : 0.f, b = k + 1 < kcols ? at(k + 1) * inv : 0.f;
        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
        *reinterpret_cast<unsigned short*>(dst + (int64_t)n * ldw + k) = (unsigned short)(pk & 0xffff);
    }
}

}  // namespace

int fie_gemm_w8_init(void) {
    hipError_t e = hipSuccess;
    auto set = [&](const void* f, int lds) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds); };
    set(reinterpret_cast<const void*>(&gemm3w8_kernel<256, 128, 3, 0, 8>), w8_lds<256, 128, 3>());
    set(reinterpret_cast<const void*>(&gemm3w8_kernel<256, 128, 3, 2, 8>), w8_lds<256, 128, 3>());
    set(reinterpret_cast<const void*>(&gemm3w8_kernel<128, 64, 3, 0, 4>), w8_lds<128, 64, 3>());
    set(reinterpret_cast<const void*>(&gemm3w8_kernel<128, 64, 3, 2, 4>), w8_lds<128, 64, 3>());
    set(reinterpret_cast<const void*>(&gemm3w8_kernel<64, 64, 3, 0, 4>), w8_lds<64, 64, 3>());
    set(reinterpret_cast<const void*>(&gemm3w8_kernel<64, 64, 3, 2, 4>), w8_lds<64, 64, 3>());
    set(reinterpret_cast<const void*>(&gemm3w8_kernel<128, 128, 2, 0, 8>), w8_lds<128, 128, 2>());
    set(reinterpret_cast<const void*>(&gemm3w8_kernel<128, 128, 2, 2, 8>), w8_lds<128, 128, 2>());
    set(reinterpret_cast<const void*>(&gemm3w8_kernel<192, 128, 2, 0, 8>), w8_lds<192, 128, 2>());
    set(reinterpret_cast<const void*>(&gemm3w8_kernel<192, 128, 2, 2, 8>), w8_lds<192, 128, 2>());
    if (e != hipSuccess) {
        fie_set_error("gemm_w8: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return FIE_EHIP;
    }
    return FIE_OK;
}

template <int BM, int BN, int NW, int ST = 3>
static void launch_w8(fie_ctx* ctx, const GemmArgs& a, int conv, dim3 grid) {
    constexpr int lds = w8_lds<BM, BN, ST>();
    if (conv) fie_launch(ctx, (gemm3w8_kernel<BM, BN, ST, 2, NW>), grid, dim3(NW * 64), lds, a);
    else fie_launch(ctx, (gemm3w8_kernel<BM, BN, ST, 0, NW>), grid, dim3(NW * 64), lds, a);
}

// code: 62 (256x128 x 3 stages, 8 waves), 42 (128x64), 43 (64x64), 52 / 54 (128x128 / 192x128 x 2 stages, 8 waves: two and more blocks per CU)
int fie_launch_gemm_w8(fie_ctx* ctx, const GemmArgs& a, int conv, int code) {
    const dim3 grid((unsigned)(a.nbm * a.nbn));
    if (code == 62) launch_w8<256, 128, 8>(ctx, a, conv, grid);
    else if (code == 54) launch_w8<192, 128, 8, 2>(ctx, a, conv, grid);
    else if (code == 52) launch_w8<128, 128, 8, 2>(ctx, a, conv, grid);
    else if (code == 42) launch_w8<128, 64, 4>(ctx, a, conv, grid);
    else launch_w8<64, 64, 4>(ctx, a, conv, grid);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

extern "C" {

int fie_pack_rows_f8(fie_ctx* ctx, const void* src, int64_t ld_src, int N, int K, void* dst, int64_t ldw, int Npad, float* scales,
                     int interleave2) {
    FIE_REQUIRE(ctx && src && dst && scales, "fie_pack_rows_f8: NULL argument");
    FIE_REQUIRE(N > 0 && K > 0 && Npad >= N && ldw >= K && ldw % 64 == 0, "fie_pack_rows_f8: bad shape");
    FIE_REQUIRE(!interleave2 || N % 2 == 0, "fie_pack_rows_f8: interleave2 needs even N");
    fie_launch(ctx, (pack_f8_kernel<false>), dim3(Npad), dim3(256), 0, (const half_t*)src, ld_src, N, K, 0, 0, (unsigned char*)dst, ldw, scales, interleave2);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

int fie_pack_conv3x3_f8(fie_ctx* ctx, const void* src_oihw, int Cout, int Cin, int cin_pad, void* dst, int64_t ldw, int Npad,
                        float* scales) {
    FIE_REQUIRE(ctx && src_oihw && dst && scales, "fie_pack_conv3x3_f8: NULL argument");
    FIE_REQUIRE(cin_pad >= Cin && cin_pad % 8 == 0 && ldw >= 9 * cin_pad && ldw % 64 == 0 && Npad >= Cout, "fie_pack_conv3x3_f8: bad shape");
    fie_launch(ctx, (pack_f8_kernel<true>), dim3(Npad), dim3(256), 0, (const half_t*)src_oihw, 0, Cout, 9 * cin_pad, Cin, cin_pad, (unsigned char*)dst, ldw, scales, 0);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

}  // extern "C"
