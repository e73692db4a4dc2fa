// K3 / K3v / K4 / CLIP attention: flash-style softmax(scale QK^T [+causal]) V on fp16 MFMA (gfx950).
// (include/fie.h: fie_attention_f16)
//
// Structure (wave64, v_mfma_f32_16x16x32_f16):
//   * block = 4 waves; each wave owns QF x 16 query rows and keeps their Q fragments in registers;
//     K and V tiles of KT keys are staged once per block in LDS (128-B-chunk XOR swizzle) and shared by the waves.
//   * scores are computed TRANSPOSED, S^T = K Q^T (A operand = K rows, B operand = Q), so a lane holds one query
//     column: the softmax row statistics are lane-local (+2 xor-shuffles over the 4 lane groups) and the fp16 P^T
//     fragment is directly the B operand of O^T = V^T P^T.  V^T fragments come from the row-major V tile with
//     ds_read_b64_tr_b16 (hardware transpose read), key order permuted consistently on both operands.
//   * O^T accumulators: lane holds 4 consecutive d of one query -> the online-softmax rescale is lane-local and the
//     output store is 8 bytes per lane.
#include "fie_internal.h"

namespace {

struct AttnArgs {
    const half_t* Q; int64_t ldq;
    const half_t* K; int64_t ldk;
    const half_t* V; int64_t ldv;
    half_t* O; int64_t ldo;
    int H, Tq, Tk;
    float scale_log2;
    int causal;
    float o8_inv;      // > 0 (attn2_kernel only): O is e4m3 bytes, value * o8_inv saturated to +-448, ldo in bytes (fie_attention_f16_o8)
};

template <int D>
__device__ __forceinline__ int kv_off(int row, int chunk) {
    // row-major [KT][D] fp16, 16-byte chunks XOR-swizzled in their low 3 bits
    return row * D + ((chunk ^ (row & 7)) << 3);
}

template <int D, int QF, int KT>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs p) {
    constexpr int KF = KT / 16;        // key fragments per tile
    constexpr int DK = D / 32;         // k-steps of the QK^T product
    constexpr int DF = D / 16;         // output d fragments
    constexpr int PS = KT / 32;        // k-steps of the PV product
    constexpr int CH = D / 8;          // 16-byte chunks per row
    extern __shared__ __attribute__((aligned(16))) half_t smem[];
    half_t* sk = smem;
    half_t* sv = smem + KT * D;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (64 * QF) + wave * (16 * QF);

    const half_t* Qb = p.Q + (int64_t)b * p.Tq * p.ldq + h * D;
    const half_t* Kb = p.K + (int64_t)b * p.Tk * p.ldk + h * D;
    const half_t* Vb = p.V + (int64_t)b * p.Tk * p.ldv + h * D;

    // Q fragments (B operand of S^T): lane holds Q[q = q0 + qf*16 + fr][d = kk*32 + fq*8 .. +7]
    f16x8 qf_[QF][DK];
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < QF; ++a) {
        const int q = q0 + a * 16 + fr;
#pragma unroll
        for (int kk = 0; kk < DK; ++kk)
            qf_[a][kk] = q < p.Tq ? *reinterpret_cast<const f16x8*>(Qb + (int64_t)q * p.ldq + kk * 32 + fq * 8) : zero8;
    }

    f32x4 o[QF][DF];
    float mrun[QF], lrun[QF];
#pragma unroll
    for (int a = 0; a < QF; ++a) {
        mrun[a] = -1e30f;
        lrun[a] = 0.f;
#pragma unroll
        for (int d = 0; d < DF; ++d) o[a][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    int kend = p.Tk;
    if (p.causal) {
        const int qmax = min(p.Tq, (int)(blockIdx.x + 1) * 64 * QF);   // keys <= last query of the block
        kend = min(kend, qmax);
    }
    const int ntiles = (kend + KT - 1) / KT;

    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * KT;
        __syncthreads();       // previous tile fully consumed
        // stage K and V tiles: KT*CH chunks each
        for (int i = tid; i < KT * CH; i += 256) {
            const int row = i / CH, ch = i - row * CH;
            const int key = key0 + row;
            f16x8 kv = zero8, vv = zero8;
            if (key < p.Tk) {
                kv = *reinterpret_cast<const f16x8*>(Kb + (int64_t)key * p.ldk + ch * 8);
                vv = *reinterpret_cast<const f16x8*>(Vb + (int64_t)key * p.ldv + ch * 8);
            }
            *reinterpret_cast<f16x8*>(sk + kv_off<D>(row, ch)) = kv;
            *reinterpret_cast<f16x8*>(sv + kv_off<D>(row, ch)) = vv;
        }
        __syncthreads();

        // S^T[key][q] = sum_d K[key][d] Q[q][d]
        f32x4 s[QF][KF];
#pragma unroll
        for (int a = 0; a < QF; ++a)
#pragma unroll
            for (int f = 0; f < KF; ++f) s[a][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) {
#pragma unroll
            for (int f = 0; f < KF; ++f) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(sk + kv_off<D>(f * 16 + fr, kk * 4 + fq));
#pragma unroll
                for (int a = 0; a < QF; ++a)
                    s[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf_[a][kk], s[a][f], 0, 0, 0);
            }
        }

        // online softmax; lane element (a, f, r): key = key0 + f*16 + fq*4 + r, query = q0 + a*16 + fr
        f16x8 pf[QF][PS];
#pragma unroll
        for (int a = 0; a < QF; ++a) {
            const int q = q0 + a * 16 + fr;
            float mx = -1e30f;
#pragma unroll
            for (int f = 0; f < KF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = key0 + f * 16 + fq * 4 + r;
                    float v = s[a][f][r] * p.scale_log2;
                    if (key >= p.Tk || (p.causal && key > q)) v = -1e30f;
                    s[a][f][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(mrun[a], mx);
            const float alpha = exp2f(mrun[a] - mnew);
            mrun[a] = mnew;
            float sum = 0.f;
#pragma unroll
            for (int f = 0; f < KF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(s[a][f][r] - mnew);
                    sum += e;
                    pf[a][f >> 1][(f & 1) * 4 + r] = (half_t)e;
                }
            lrun[a] = lrun[a] * alpha + sum;
#pragma unroll
            for (int d = 0; d < DF; ++d) o[a][d] *= alpha;
        }

        // O^T[d][q] += sum_key V[key][d] P[q][key]; V^T fragments by transposed LDS reads
        const int trow = fq * 4 + (fr >> 2);                 // row inside a 16-key block supplied by this lane
        const int tsub = (lane & 1) * 4;                     // element offset inside the 16-byte chunk
#pragma unroll
        for (int d = 0; d < DF; ++d) {
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) {
                const int ra = ps * 32 + trow, rb = ra + 16;
                const int ch = d * 2 + ((lane & 3) >> 1);
                const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(sv + kv_off<D>(ra, ch) + tsub));
                const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(sv + kv_off<D>(rb, ch) + tsub));
                union { struct { s16x4 lo, hi; } s; f16x8 v; } u;
                u.s.lo = va;
                u.s.hi = vb;
#pragma unroll
                for (int a = 0; a < QF; ++a)
                    o[a][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u.v, pf[a][ps], o[a][d], 0, 0, 0);
            }
        }
    }

    // finalize: total row sum over the 4 lane groups, normalise, store 4 consecutive d per lane
#pragma unroll
    for (int a = 0; a < QF; ++a) {
        float l = lrun[a];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int q = q0 + a * 16 + fr;
        if (q >= p.Tq) continue;
        half_t* orow = p.O + ((int64_t)b * p.Tq + q) * p.ldo + h * D;
#pragma unroll
        for (int d = 0; d < DF; ++d) {
            f16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[a][d][r] * inv);
            *reinterpret_cast<f16x4*>(orow + d * 16 + fq * 4) = v;
        }
    }
}

template <int D, int QF, int KT>
int launch_attn(fie_ctx* ctx, const AttnArgs& a, int B) {
    const size_t lds = (size_t)2 * KT * D * sizeof(half_t);      // function attribute (dynamic LDS limit): fie_attn_init
    const dim3 grid((unsigned)((a.Tq + 64 * QF - 1) / (64 * QF)), (unsigned)a.H, (unsigned)B), block(256);
    fie_launch(ctx, (attn_kernel<D, QF, KT>), grid, block, lds, a);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}


// =====================================================================================================================
// v2: same data flow, restructured around the two costs the profile showed (VALU-bound softmax, unpipelined staging):
//   * K/V tiles arrive by LDS-DMA into a 2-stage ring (prefetch of tile t+1 overlaps tile t; one barrier per tile)
//   * Q is pre-scaled by scale*log2(e); the score accumulators start at -m_ref so exp2 needs no subtraction; the
//     running reference only moves when a row max grows by more than 2^6 (deferred rescale) or on the first tile
//   * masks are applied only on boundary tiles; max via v_max3 + permlane swaps; P packed with v_cvt_pkrtz and its row
//     sum taken from the packed fp16 values with v_dot2 (numerator and denominator see the same rounding)

__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max2f(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// xor-16 / xor-32 exchanges by v_permlane16_swap / v_permlane32_swap (VALU, no LDS round trip): after swap(x, y = x)
// every lane holds {own, partner} in {r[0], r[1]}.  Two traps met on hardware / in the ISA: (1) LLVM folds swap(x, x) with
// one SSA value on both operands as if the outputs were equal (it emitted max(r0, r0)), so the second operand is laundered
// through an empty asm; (2) the value must come from a compiler-visible VALU op so the hazard recogniser can pad
// "VALU write -> v_permlane read" (it cannot see into inline asm).
__device__ __forceinline__ void swap_pair(float x, bool half32, float& a, float& b) {
    unsigned u = __builtin_bit_cast(unsigned, x), v = u;
    asm volatile("" : "+v"(v));
    const auto r = half32 ? __builtin_amdgcn_permlane32_swap(u, v, false, false) : __builtin_amdgcn_permlane16_swap(u, v, false, false);
    unsigned ra = r[0], rb = r[1];
    asm volatile("" : "+v"(ra), "+v"(rb));         // ... and so are the two results
    a = __builtin_bit_cast(float, ra);
    b = __builtin_bit_cast(float, rb);
}
__device__ __forceinline__ float xor16_32_max(float x) {
    float a, b;
    swap_pair(__builtin_canonicalizef(x), false, a, b);
    swap_pair(__builtin_fmaxf(a, b), true, a, b);
    return __builtin_fmaxf(a, b);
}
__device__ __forceinline__ float xor16_32_sum(float x) {
    float a, b;
    swap_pair(x + 0.0f, false, a, b);
    swap_pair(a + b, true, a, b);
    return a + b;
}

__device__ __forceinline__ void attn_bload16(__amdgpu_buffer_rsrc_t rsrc, half_t* lds_dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

// NWV = waves per block (4; 2 = the "narrow" form: 64 queries per block as 2 waves x 32 queries, see fie_attention_f16): a wave re-reads the whole K / V
// tile from LDS for its 16 * QF queries, so QF = 1 needs 1 KiB of fragment reads per MFMA -- the full LDS bandwidth of a CU at the MFMA rate.
// ST = stages of the K / V ring: 2 = tile t+1 in flight during tile t behind a full vmcnt(0); 3 = two tiles ahead behind a counted wait.
template <int D, int QF, int KT, int NWV = 4, int ST = 2>
__global__ __launch_bounds__(NWV * 64) void attn2_kernel(AttnArgs p) {
    constexpr int KF = KT / 16, DK = D / 32, DF = D / 16, PS = KT / 32;
    constexpr int CH = D / 8;                  // 16-byte chunks per row
    constexpr int RP = 64 / CH > 0 ? 64 / CH : 1;   // rows per 1-KiB LDS-DMA piece (8 for D=64, 1 for D=512)
    constexpr int PPR = CH / 64 > 0 ? CH / 64 : 1;  // pieces per row (1)
    static_assert(PPR == 1, "row longer than one LDS-DMA piece");
    constexpr int NPT = KT / RP;               // pieces per K (or V) tile
    constexpr int PW = 2 * NPT / NWV;          // pieces per wave per tile (K and V)
    constexpr int TILE = KT * D;               // halfs
    constexpr float THR = 6.0f;
    extern __shared__ __attribute__((aligned(16))) half_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (NWV * 16 * QF) + wave * (16 * QF);

    const half_t* Qb = p.Q + (int64_t)b * p.Tq * p.ldq + h * D;
    const half_t* Kb = p.K + (int64_t)b * p.Tk * p.ldk + h * D;
    const half_t* Vb = p.V + (int64_t)b * p.Tk * p.ldv + h * D;

    // LDS-DMA (buffer_load ... lds): per-lane byte offsets inside a tile are computed ONCE; the tile's first key enters as
    // the scalar offset; keys >= Tk fall outside the descriptor and read as zero.  Pieces wave + 4 i: first NPT = K, rest = V.
    const int prow = lane / CH, pphys = lane % CH;
    const int64_t kb64 = ((int64_t)(p.Tk - 1) * p.ldk + D) * 2, vb64 = ((int64_t)(p.Tk - 1) * p.ldv + D) * 2;
    const unsigned kbytes = kb64 > 0x7fffffff ? 0x7fffffffu : (unsigned)kb64;
    const unsigned vbytes = vb64 > 0x7fffffff ? 0x7fffffffu : (unsigned)vb64;
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)kbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)vbytes, 0x00020000);
    unsigned pvoff[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int pc = wave + NWV * i;
        const bool isv = pc >= NPT;
        const int row = (isv ? pc - NPT : pc) * RP + prow;
        pvoff[i] = (unsigned)row * (unsigned)(isv ? p.ldv : p.ldk) * 2u + (unsigned)(pphys ^ (row & 7)) * 16u;
    }

    auto issue = [&](int t, int stage) {
        half_t* sk = smem + stage * 2 * TILE;
        half_t* sv = sk + TILE;
        const unsigned sok = (unsigned)(t * KT) * (unsigned)p.ldk * 2u, sov = (unsigned)(t * KT) * (unsigned)p.ldv * 2u;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int pc = wave + NWV * i;
            const bool isv = pc >= NPT;
            const int pr = isv ? pc - NPT : pc;
            // (the builtin must sit in a __device__ helper: used directly in the __global__ body, the host pass silently
            // drops the kernel's stub and the library fails to load)
            if (isv) attn_bload16(rs_v, sv + pr * 512, pvoff[i], sov);
            else attn_bload16(rs_k, sk + pr * 512, pvoff[i], sok);
        }
    };

    int kend = p.Tk;
    if (p.causal) kend = min(kend, min(p.Tq, (int)(blockIdx.x + 1) * NWV * 16 * QF));
    const int ntiles = (kend + KT - 1) / KT;
    issue(0, 0);
    if (ST == 3 && ntiles > 1) issue(1, 1);

    // Q fragments, pre-scaled so that scores are already in the log2 domain
    f16x8 qf_[QF][DK];
#pragma unroll
    for (int a = 0; a < QF; ++a) {
        const int q = q0 + a * 16 + fr;
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) {
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (q < p.Tq) v = *reinterpret_cast<const f16x8*>(Qb + (int64_t)q * p.ldq + kk * 32 + fq * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)v[j] * p.scale_log2);
            qf_[a][kk] = v;
        }
    }

    f32x4 o[QF][DF];
    f32x4 negm[QF];
    float mref[QF], lrun[QF];
#pragma unroll
    for (int a = 0; a < QF; ++a) {
        mref[a] = 0.f;
        negm[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        lrun[a] = 0.f;
#pragma unroll
        for (int d = 0; d < DF; ++d) o[a][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f16x2 ones = {(half_t)1.f, (half_t)1.f};
    // per-lane LDS bases (halfs).  K fragment (row f*16 + fr, chunk kk*4 + fq): the swizzle only touches the low 3 chunk
    // bits, so base[kk & 1] + (kk >> 1) * 64 + f * 16 * D.  V^T transposed read (row ps*32 [+16] + trow, chunk d*2 + c1):
    // base[d & 3] + (d >> 2) * 64 + row constants.
    const int trow = fq * 4 + (fr >> 2), tsub = (lane & 1) * 4, c1 = (lane & 3) >> 1;
    int kbase[2], vbase[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) kbase[j] = fr * D + ((((j << 2) | fq) ^ (fr & 7)) << 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) vbase[j] = trow * D + ((((j << 1) | c1) ^ (trow & 7)) << 3) + tsub;

    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * KT;
        if (ST == 3 && t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(ST == 3 ? PW : 0) : "memory");   // tile t landed (t+1 may still fly)
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // tile t landed; tile t-1 fully consumed
        if (ST == 3) { if (t + 2 < ntiles) issue(t + 2, (t + 2) % 3); }
        else if (t + 1 < ntiles) issue(t + 1, (t + 1) & 1);
        const half_t* sk = smem + (ST == 3 ? t % 3 : t & 1) * 2 * TILE;
        const half_t* sv = sk + TILE;

        // the score chains start from the persistent -m_ref registers (C operand of the first MFMA): no per-tile init
        f32x4 s[QF][KF];
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) {
#pragma unroll
            for (int f = 0; f < KF; ++f) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(sk + kbase[kk & 1] + (kk >> 1) * 64 + f * 16 * D);
#pragma unroll
                for (int a = 0; a < QF; ++a)
                    s[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf_[a][kk], kk == 0 ? negm[a] : s[a][f], 0, 0, 0);
            }
        }

        const bool edge = (key0 + KT > p.Tk) || (p.causal && key0 + KT > q0);     // wave-uniform
        u32x4 pf[QF][PS];                 // packed fp16 P^T fragments: 4 dwords = 8 halfs, no half-extract/insert traffic
#pragma unroll
        for (int a = 0; a < QF; ++a) {
            if (edge) {
                const int q = q0 + a * 16 + fr;
#pragma unroll
                for (int f = 0; f < KF; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = key0 + f * 16 + fq * 4 + r;
                        if (key >= p.Tk || (p.causal && key > q)) s[a][f][r] = -1e30f;
                    }
            }
            float mx = max3f(s[a][0][0], s[a][0][1], s[a][0][2]);
            mx = max2f(mx, s[a][0][3]);
#pragma unroll
            for (int f = 1; f < KF; ++f) {
                mx = max3f(mx, s[a][f][0], s[a][f][1]);
                mx = max3f(mx, s[a][f][2], s[a][f][3]);
            }
            mx = xor16_32_max(mx);
            if (t == 0 || __any(mx > THR)) {
                // move the reference: rows whose max grew (or every row on the first tile) are re-based
                const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                mref[a] += delta;
                negm[a] = (f32x4){-mref[a], -mref[a], -mref[a], -mref[a]};
                lrun[a] *= alpha;
#pragma unroll
                for (int d = 0; d < DF; ++d) o[a][d] *= alpha;
#pragma unroll
                for (int f = 0; f < KF; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[a][f][r] -= delta;
            }
            float sum = 0.f;
#pragma unroll
            for (int f = 0; f < KF; ++f) {
                const auto lo = __builtin_amdgcn_cvt_pkrtz(__builtin_amdgcn_exp2f(s[a][f][0]), __builtin_amdgcn_exp2f(s[a][f][1]));
                const auto hi = __builtin_amdgcn_cvt_pkrtz(__builtin_amdgcn_exp2f(s[a][f][2]), __builtin_amdgcn_exp2f(s[a][f][3]));
                sum = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, lo), ones, sum, false);
                sum = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, hi), ones, sum, false);
                pf[a][f >> 1][(f & 1) * 2 + 0] = __builtin_bit_cast(unsigned, lo);
                pf[a][f >> 1][(f & 1) * 2 + 1] = __builtin_bit_cast(unsigned, hi);
            }
            lrun[a] += sum;
        }

#pragma unroll
        for (int d = 0; d < DF; ++d) {
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) {
                const half_t* vp = sv + vbase[d & 3] + (d >> 2) * 64 + ps * 32 * D;
                const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp));
                const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 16 * D));
                union { struct { s16x4 lo, hi; } s; f16x8 v; } u;
                u.s.lo = va;
                u.s.hi = vb;
#pragma unroll
                for (int a = 0; a < QF; ++a)
                    o[a][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u.v, __builtin_bit_cast(f16x8, pf[a][ps]), o[a][d], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int a = 0; a < QF; ++a) {
        const float l = xor16_32_sum(lrun[a]);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int q = q0 + a * 16 + fr;
        if (q >= p.Tq) continue;
        if (p.o8_inv > 0.f) {                               // e4m3 output for an fp8-activation out-projection: 4 bytes per lane and fragment
            unsigned char* orow8 = reinterpret_cast<unsigned char*>(p.O) + ((int64_t)b * p.Tq + q) * p.ldo + h * D;
            const float s8 = inv * p.o8_inv;
#pragma unroll
            for (int d = 0; d < DF; ++d) {
                float x[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = fie_sat448(o[a][d][r] * s8);
                int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], pk, true);
                *reinterpret_cast<int*>(orow8 + d * 16 + fq * 4) = pk;
            }
            continue;
        }
        half_t* orow = p.O + ((int64_t)b * p.Tq + q) * p.ldo + h * D;
#pragma unroll
        for (int d = 0; d < DF; ++d) {
            f16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[a][d][r] * inv);
            *reinterpret_cast<f16x4*>(orow + d * 16 + fq * 4) = v;
        }
    }
}

template <int D, int QF, int KT, int NWV = 4, int ST = 2>
int launch_attn2(fie_ctx* ctx, const AttnArgs& a, int B) {
    const size_t lds = (size_t)2 * ST * KT * D * sizeof(half_t);
    constexpr int QB = NWV * 16 * QF;                     // queries per block
    const dim3 grid((unsigned)((a.Tq + QB - 1) / QB), (unsigned)a.H, (unsigned)B), block(NWV * 64);
    fie_launch(ctx, (attn2_kernel<D, QF, KT, NWV, ST>), grid, block, lds, a);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

template <int D, int QF, int KT>
hipError_t attn_attrs() {      // dynamic-LDS limits of both kernel generations, once per device (fie_ctx_create), never on the launch path
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<D, QF, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KT * D * (int)sizeof(half_t));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_kernel<D, QF, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * KT * D * (int)sizeof(half_t));
    return e;
}

}  // namespace

int fie_attn_init(void) {
    hipError_t e = attn_attrs<512, 1, 32>();
    if (e == hipSuccess) e = attn_attrs<64, 2, 64>();
    if (e == hipSuccess) e = attn_attrs<64, 1, 64>();
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_kernel<64, 2, 64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 64 * (int)sizeof(half_t));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_kernel<64, 1, 96>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 96 * 64 * (int)sizeof(half_t));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_kernel<64, 1, 64, 4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 64 * 64 * (int)sizeof(half_t));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_kernel<64, 1, 64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 64 * (int)sizeof(half_t));
    if (e != hipSuccess) {
        fie_set_error("attention: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return FIE_EHIP;
    }
    return FIE_OK;
}

static int attention_impl(fie_ctx* ctx, const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O, int64_t ldo, int B, int H,
                          int Tq, int Tk, int D, float scale, int causal, float o8_inv) {
    FIE_REQUIRE(ctx && Q && K && V && O, "fie_attention_f16: NULL argument");
    FIE_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "fie_attention_f16: bad shape");
    FIE_REQUIRE(D == 64 || D == 512, "fie_attention_f16: head dim %d not built (64, 512)", D);
    FIE_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "fie_attention_f16: strides must be 16-byte aligned");
    FIE_REQUIRE(ldq >= (int64_t)H * D && ldk >= (int64_t)H * D && ldv >= (int64_t)H * D && ldo >= (int64_t)H * D,
                "fie_attention_f16: row stride smaller than H*D");
    AttnArgs a;
    a.Q = (const half_t*)Q; a.ldq = ldq; a.K = (const half_t*)K; a.ldk = ldk; a.V = (const half_t*)V; a.ldv = ldv;
    a.O = (half_t*)O; a.ldo = ldo; a.H = H; a.Tq = Tq; a.Tk = Tk; a.causal = causal;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.o8_inv = o8_inv;
    FIE_REQUIRE(o8_inv == 0.f || (ctx->attn_variant != 1 && D == 64), "fie_attention_f16_o8: e4m3 output is built into the d = 64 kernel only");
    FIE_DESC(ctx, "attn B=%d H=%d Tq=%d Tk=%d D=%d flop=%.0f", B, H, Tq, Tk, D, 4.0 * B * H * Tq * Tk * D);
    const int64_t blocks128 = (int64_t)((Tq + 127) / 128) * H * B;
    if (ctx->attn_variant == 1) {
        if (D == 512) return launch_attn<512, 1, 32>(ctx, a, B);
        if (blocks128 >= ctx->num_cus * 2) return launch_attn<64, 2, 64>(ctx, a, B);
        return launch_attn<64, 1, 64>(ctx, a, B);
    }
    if (D == 512) return launch_attn2<512, 1, 32>(ctx, a, B);
    if (ctx->attn_variant == 2) return launch_attn2<64, 2, 64>(ctx, a, B);      // A/B: 128 queries per block everywhere
    if (ctx->attn_variant == 3) return launch_attn2<64, 1, 64>(ctx, a, B);      // A/B: 64 queries per block as 4 waves x 16 everywhere
    if (ctx->attn_variant == 4) return launch_attn2<64, 2, 64, 2>(ctx, a, B);   // A/B: 64 queries per block as TWO waves x 32 (half the LDS fragment reads per MFMA)
    if (ctx->attn_variant == 5) return launch_attn2<64, 1, 64, 4, 3>(ctx, a, B);   // A/B: three-stage K / V ring (two tiles ahead, counted vmcnt)
    if (ctx->attn_variant == 6) return launch_attn2<64, 1, 64, 8>(ctx, a, B);      // A/B: 128 queries per block as EIGHT waves x 16 (one K / V tile fill serves twice the queries)
    // Round 3, whole UNet forward (profiles/r03_attention_block_forms_in_unet.log): 16 queries per wave everywhere 15.24 ms; 32 per wave for the
    // 4096-token maps (the round-2 rule) 15.40-15.44 ms, whether the small maps run 4 x 16 or 2 x 32 -- so 128-query blocks only for grids of
    // four and more waves of them (batched jobs), where the round-2 measurements were taken
    if (blocks128 >= ctx->num_cus * 4) return launch_attn2<64, 2, 64>(ctx, a, B);
    // cross-attention over the 77 text tokens: ONE 96-key tile instead of two 64-key tiles (the second one 13 keys wide): no second pass through the
    // wait / barrier / softmax-rescale machinery of the key loop (round 3: 7.9 -> see profiles/README.md)
    if (Tk <= 96 && !causal && ctx->attn_variant == 0) return launch_attn2<64, 1, 96>(ctx, a, B);
    return launch_attn2<64, 1, 64>(ctx, a, B);
}

extern "C" int fie_attention_f16(fie_ctx* ctx, const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V,
                                 int64_t ldv, void* O, int64_t ldo, int B, int H, int Tq, int Tk, int D, float scale,
                                 int causal) {
    return attention_impl(ctx, Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Tq, Tk, D, scale, causal, 0.f);
}

// The same attention with an e4m3 output (BASELINE config 5: the attention out-projections read fp8 activations): O8 [B * Tq, H * D] bytes,
// row stride ldo8 BYTES, value * inv_scale saturated to the e4m3 range.  d = 64 only.
extern "C" int fie_attention_f16_o8(fie_ctx* ctx, const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv, void* O8,
                                    int64_t ldo8, int B, int H, int Tq, int Tk, int D, float scale, int causal, float inv_scale) {
    FIE_REQUIRE(inv_scale > 0.f, "fie_attention_f16_o8: inv_scale must be positive");
    return attention_impl(ctx, Q, ldq, K, ldk, V, ldv, O8, ldo8, B, H, Tq, Tk, D, scale, causal, inv_scale);
}

extern "C" int fie_debug_attn_variant(fie_ctx* ctx, int v) {
    FIE_REQUIRE(ctx != nullptr && v >= 0 && v <= 6, "fie_debug_attn_variant: bad argument");
    ctx->attn_variant = v;
    return FIE_OK;
}
