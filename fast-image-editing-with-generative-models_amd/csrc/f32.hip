// fp32 ("--full_precision" / "--quality_mode") path of the C ABI: the same graphs with fp32 storage and exact fp32
// arithmetic on v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32 FMA chain; 1/16 of the fp16 MFMA rate -- this path buys
// reference-grade numerics, not speed, and doubles as an on-GPU fp32 check of the fp16 path at full size).
// (include/fie.h: fie_gemm_f32, fie_conv3x3_nhwc_f32, fie_softmax_rows_f32)
//
// One contraction kernel serves Linear, 3x3 conv (same im2col gather as the fp16 kernels) and both attention products
// (batched over (image, head) through blockIdx.z; the P V product reads V as a [K][N] matrix).  64x64x16 tile, 4 waves,
// register-staged single LDS buffer (row stride 17 floats: conflict-free single-float fragment reads), swapped operands
// as in the fp16 kernels (a lane owns 4 consecutive output columns).  Attention = Q K^T GEMM -> row softmax -> P V GEMM:
// with 288 GB of HBM the fp32 score matrices (<= 3 GB) are simply materialised.
#include "fie_internal.h"

namespace {

struct G32 {
    const float* A1; int64_t lda1; int K1;
    const float* A2; int64_t lda2;
    int H, W, Cin, OH, OW, stride, pt, pl, ups;      // conv view of A1
    const float* Wt; int64_t ldw; int w_kn;           // w_kn: weights given as [K][N]
    float* C; int64_t ldc;
    int M, N, K;
    const float* bias;
    const float* rowbias; int64_t ld_rowbias; int rows_per_batch;
    const float* res; int64_t ldr;
    float scale; int act;
    int nb2;                                          // batch z = z1 * nb2 + z2
    int64_t sA1, sA2, sW1, sW2, sC1, sC2;
};

constexpr int T = 64, KB = 16, LDS_LD = 17;

template <int MODE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(G32 p) {
    __shared__ float sa[T * LDS_LD], sw[T * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
    const int z1 = blockIdx.z / p.nb2, z2 = blockIdx.z - z1 * p.nb2;
    const float* A1 = p.A1 + z1 * p.sA1 + z2 * p.sA2;
    const float* Wt = p.Wt + z1 * p.sW1 + z2 * p.sW2;
    float* C = p.C + z1 * p.sC1 + z2 * p.sC2;

    const int kc = tid & 15, r0 = tid >> 4;           // A / W[N][K] staging: column kc, rows r0 + 16 i
    const int nc = tid & 63, kr = tid >> 6;           // W[K][N] staging: column nc, k rows kr + 4 i
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + KB - 1) / KB;
    for (int kt = 0; kt < nk; ++kt) {
        float ra[4], rw[4];
        const int k = kt * KB + kc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + r0 + 16 * i;
            float v = 0.f;
            if (m < p.M && k < p.K) {
                if (MODE == 1) {
                    const int tap = k / p.Cin, ci = k - tap * p.Cin;
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    const int hw = p.OH * p.OW;
                    const int b = m / hw, rem = m - b * hw;
                    const int oh = rem / p.OW, ow = rem - oh * p.OW;
                    const int ih = oh * p.stride - p.pt + ky, iw = ow * p.stride - p.pl + kx;
                    if (ih >= 0 && ih < (p.H << p.ups) && iw >= 0 && iw < (p.W << p.ups))
                        v = A1[((int64_t)(b * p.H + (ih >> p.ups)) * p.W + (iw >> p.ups)) * p.Cin + ci];
                } else {
                    v = k < p.K1 ? A1[(int64_t)m * p.lda1 + k] : p.A2[(int64_t)m * p.lda2 + (k - p.K1)];
                }
            }
            ra[i] = v;
        }
        if (!p.w_kn) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + r0 + 16 * i;
                rw[i] = (n < p.N && k < p.K) ? Wt[(int64_t)n * p.ldw + k] : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = kt * KB + kr + 4 * i, n = n0 + nc;
                rw[i] = (n < p.N && kk < p.K) ? Wt[(int64_t)kk * p.ldw + n] : 0.f;
            }
        }
        __syncthreads();                               // previous tile consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) sa[(r0 + 16 * i) * LDS_LD + kc] = ra[i];
        if (!p.w_kn) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sw[(r0 + 16 * i) * LDS_LD + kc] = rw[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) sw[nc * LDS_LD + kr + 4 * i] = rw[i];
        }
        __syncthreads();
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float fw[2], fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[i] = sw[(wn * 32 + i * 16 + fr) * LDS_LD + ks * 4 + fq];
#pragma unroll
            for (int j = 0; j < 2; ++j) fa[j] = sa[(wm * 32 + j * 16 + fr) * LDS_LD + ks * 4 + fq];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
    }
    // epilogue: lane holds C[m = .. + fr][n = .. + fq*4 + r]
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wm * 32 + j * 16 + fr;
        if (m >= p.M) continue;
        const float* rb = p.rowbias ? p.rowbias + (int64_t)(m / p.rows_per_batch) * p.ld_rowbias : nullptr;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = n0 + wn * 32 + i * 16 + fq * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r];
                if (n + r < p.N) {
                    if (p.bias) v[r] += p.bias[n + r];
                    if (rb) v[r] += rb[n + r];
                }
            }
            if (p.act == FIE_ACT_GEGLU) {              // interleaved (value, gate) rows -> N/2 outputs
#pragma unroll
                for (int r = 0; r < 4; r += 2)
                    if (n + r + 1 < p.N) C[(int64_t)m * p.ldc + ((n + r) >> 1)] = v[r] * fie_gelu_exact(v[r + 1]) * p.scale;
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r >= p.N) continue;
                float x = v[r];
                if (p.act == FIE_ACT_SILU) x = x / (1.0f + expf(-x));
                else if (p.act == FIE_ACT_GELU) x = fie_gelu_exact(x);
                else if (p.act == FIE_ACT_QUICK_GELU) x = x / (1.0f + expf(-1.702f * x));
                x *= p.scale;
                if (p.res) x += p.res[(int64_t)m * p.ldr + n + r];
                C[(int64_t)m * p.ldc + n + r] = x;
            }
        }
    }
}

// in-place row softmax of S[rows][cols] (row stride ld): softmax(scale * s), keys > q masked when causal (q = row % tq)
__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(float* S, int64_t rows, int cols, int64_t ld, float scale, int causal, int tq) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* s = S + row * ld;
    const int lim = causal ? min(cols, (int)(row % tq) + 1) : cols;
    float mx = -INFINITY;
    for (int c = lane; c < lim; c += 64) mx = fmaxf(mx, s[c] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int c = lane; c < lim; c += 64) sum += expf(s[c] * scale - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int c = lane; c < cols; c += 64) s[c] = c < lim ? expf(s[c] * scale - mx) * inv : 0.f;
}

int launch32(fie_ctx* ctx, G32& a, int mode, int batch) {
    const dim3 grid((unsigned)((a.N + T - 1) / T), (unsigned)((a.M + T - 1) / T), (unsigned)batch);
    if (mode == 1) fie_launch(ctx, gemm_f32_kernel<1>, grid, dim3(256), 0, a);
    else fie_launch(ctx, gemm_f32_kernel<0>, grid, dim3(256), 0, a);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

}  // namespace

extern "C" {

int fie_gemm_f32(fie_ctx* ctx, const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, const float* W,
                 int64_t ldw, int w_is_kn, float* C, int64_t ldc, int M, int N, int K, const float* bias,
                 const float* rowbias, int64_t ld_rowbias, int rows_per_batch, const float* residual, int64_t ldr,
                 float scale, int act, int nb1, int nb2, int64_t sA1, int64_t sA2, int64_t sW1, int64_t sW2, int64_t sC1,
                 int64_t sC2) {
    FIE_REQUIRE(ctx && A1 && W && C, "fie_gemm_f32: NULL ctx/A1/W/C");
    FIE_REQUIRE(M > 0 && N > 0 && K > 0 && K1 > 0 && K1 <= K && (K1 == K || A2), "fie_gemm_f32: bad shape M=%d N=%d K=%d K1=%d", M, N, K, K1);
    FIE_REQUIRE(act >= FIE_ACT_NONE && act <= FIE_ACT_GEGLU && !(act == FIE_ACT_GEGLU && (residual || N % 2)), "fie_gemm_f32: bad epilogue");
    FIE_REQUIRE(nb1 >= 1 && nb2 >= 1 && (int64_t)nb1 * nb2 <= 65535, "fie_gemm_f32: bad batch %d x %d", nb1, nb2);
    FIE_REQUIRE(!rowbias || rows_per_batch > 0, "fie_gemm_f32: rowbias needs rows_per_batch");
    G32 a = {};
    a.A1 = A1; a.lda1 = lda1; a.K1 = K1; a.A2 = A2; a.lda2 = lda2; a.Wt = W; a.ldw = ldw; a.w_kn = w_is_kn; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.bias = bias; a.rowbias = rowbias; a.ld_rowbias = ld_rowbias;
    a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1; a.res = residual; a.ldr = ldr; a.scale = scale; a.act = act;
    a.nb2 = nb2; a.sA1 = sA1; a.sA2 = sA2; a.sW1 = sW1; a.sW2 = sW2; a.sC1 = sC1; a.sC2 = sC2;
    return launch32(ctx, a, 0, nb1 * nb2);
}

int fie_conv3x3_nhwc_f32(fie_ctx* ctx, const float* X, int B, int H, int W, int Cin, int upsample2x, int stride, int pad_mode,
                         const float* Wkc, int64_t ldw, float* Y, int64_t ldc, int Cout, const float* bias,
                         const float* rowbias, int64_t ld_rowbias, const float* residual, int64_t ldr, float scale, int act) {
    FIE_REQUIRE(ctx && X && Wkc && Y, "fie_conv3x3_nhwc_f32: NULL ctx/X/W/Y");
    FIE_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2) && (pad_mode == 0 || pad_mode == 1) &&
                    act != FIE_ACT_GEGLU && ldw >= 9 * Cin, "fie_conv3x3_nhwc_f32: bad argument");
    const int ups = upsample2x ? 1 : 0, pads = pad_mode == 0 ? 2 : 1;
    const int OH = ((H << ups) + pads - 3) / stride + 1, OW = ((W << ups) + pads - 3) / stride + 1;
    G32 a = {};
    a.A1 = X; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.stride = stride; a.pt = a.pl = pad_mode == 0 ? 1 : 0; a.ups = ups;
    a.Wt = Wkc; a.ldw = ldw; a.C = Y; a.ldc = ldc; a.M = B * OH * OW; a.N = Cout; a.K = 9 * Cin; a.K1 = a.K;
    a.bias = bias; a.rowbias = rowbias; a.ld_rowbias = ld_rowbias; a.rows_per_batch = OH * OW; a.res = residual; a.ldr = ldr;
    a.scale = scale; a.act = act; a.nb2 = 1;
    return launch32(ctx, a, 1, 1);
}

int fie_softmax_rows_f32(fie_ctx* ctx, float* S, int64_t rows, int cols, int64_t ld, float scale, int causal, int tq) {
    FIE_REQUIRE(ctx && S && rows > 0 && cols > 0 && ld >= cols && tq > 0, "fie_softmax_rows_f32: bad argument");
    fie_launch(ctx, softmax_rows_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S, rows, cols, ld, scale, causal, tq);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

}  // extern "C"
