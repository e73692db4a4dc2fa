// 256x256x64 fp16 MFMA GEMM / 3x3 implicit-GEMM convolution with a PHASED K-loop (the "256^2 8-phase" schedule of
// /opt/skills/guides/cdna_hip_programming.md section 5, restated for this library's data path: buffer_load ... lds DMA through
// wave-uniform descriptors, swapped MFMA operands, the (row & 7) chunk swizzle).  Entries: fie_gemm_f16 / fie_conv3x3_nhwc_f16
// pick it for shapes that give every CU a 256x256 tile (gemm_conv.hip, launch table).
//
// Geometry.  512 threads = 8 waves as 2 (activation rows) x 4 (output channels); a wave owns 128 rows x 64 channels = 8 x 4
// MFMA tiles (128 accumulator VGPRs).  LDS = two K-tile buffers of (256 activation rows + 256 weight rows) x 128 B = 128 KiB.
//
// A K-tile is FOUR PHASES of 16 MFMAs, one quadrant (64 rows x 32 channels x K 64) of the wave's tile each, in the order
// (a0,w0) (a0,w1) (a1,w1) (a1,w0); a phase is
//       ds_read the fragments the quadrant still lacks | issue ONE half-tile of DMA | lgkmcnt(0) | s_barrier | 16 MFMA | s_barrier
// and the two wave groups (waves 0-3: rows 0-127, waves 4-7: rows 128-255; one wave of each group per SIMD) run ONE BARRIER
// APART, so that on every SIMD one wave issues MFMAs while its partner reads LDS and issues DMA: the matrix pipe never waits for
// a load segment.
//
// DMA half-tiles = the 128 rows that all waves read in the same phase: W0 / W1 = the first / second 32 weight rows of every
// wave column, A0 / A1 = the first / second 64 activation rows of both wave rows.  Last reads inside a K-tile: W0, A0 in phase 1
// (w0 stays in registers for phase 4), W1 in phase 2, A1 in phase 3 -- so the buffer is recycled at half-tile granularity:
// global phase q issues half-tile q + 7 of the stream (W0, A0, W1, A1 of K-tile 0, W0, ... of K-tile 1, ...):
//       phase 1 of K-tile t: A1 of t+1 (other buffer)     phase 2: W0 of t+2     phase 3: A0 of t+2     phase 4: W1 of t+2
// Every half-tile is issued >= 6 phases before its first read; ONE counted wait per K-tile (phase 4, vmcnt(6): the three
// youngest half-tiles, 2 pieces per wave each, stay in flight) retires K-tile t+1.  vmcnt never reaches 0 inside the loop.
//
// Hazards (guide: "Read a staged buffer one phase AFTER the wait that retires it"; restage rules).
//   RAW  a wave's counted wait sits before its phase-4 barrier; K-tile t+1 is first read in the NEXT phase, i.e. after a barrier
//        that every wave of BOTH groups reaches only after its own wait (group 1 lags by one barrier: its wait precedes the
//        barrier that opens group 0's next load segment).
//   WAR  every wave drains its LDS reads (lgkmcnt(0)) BEFORE the first barrier of the reading phase, so a slot may be refilled
//        one phase after its last read even though the other group is one barrier behind.
#include "gemm_common.h"

using namespace fie_gemm;

namespace {

constexpr int BUFH = 512 * BK;      // halfs per K-tile buffer: activation rows 0-255, weight rows 256-511

template <int MODE, int SPLIT>    // MODE 0 = GEMM (A = [A1 | A2]), 2 = conv with Cin % 64 == 0; SPLIT: second DMA piece inside the MFMA cluster
__global__ __launch_bounds__(512) void gemm8_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) half_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int bid = take_parity(p, xcd_remap(blockIdx.x, gridDim.x));
    const int m0 = (p.order ? bid % p.nbm : bid / p.nbn) * 256;
    const int n0 = (p.order ? bid / p.nbm : bid % p.nbn) * 256;
    const int lr = lane >> 3;
    const int c8 = (lane & 7) ^ lr;

    const int live = p.probe == 1 ? 0 : 1;                           // probe 1: zero-record descriptors drop every load
    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)p.a1_bytes * live, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, (int)p.a2_bytes * live, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, (int)p.w_bytes * live, 0x00020000);
    const int lm0 = p.probe == 2 ? 0 : m0, ln0 = p.probe == 2 ? 0 : n0;   // probe 2: all tiles load tile (0,0)

    // ---- DMA pieces (8 rows x 128 B per wave-instruction).  A half-tile has 16 pieces; this wave issues pieces `wave` and
    // `wave + 8`.  Tile row of piece i:  weights (i >> 2) * 64 + (i & 3) * 8 (+ 32 for W1);  activations i * 8 (+ 64 for A1).
    const int wrow0 = (wave >> 2) * 64 + (wave & 3) * 8;            // piece `wave`;  piece wave + 8 is 128 rows further
    const int arow0 = wave * 8;                                     // likewise
    unsigned w_off[2][2];                                           // [set][piece]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            w_off[s][j] = (unsigned)(ln0 + wrow0 + 128 * j + 32 * s + lr) * (unsigned)p.ldw * 2u + c8 * 16u;
    unsigned a_off[2][2], a_off2[2][2];                             // GEMM: offsets into A1 / A2; conv: offsets of the current tap
    int a_ih[2][2], a_iw[2][2];
    unsigned a_img[2][2];
    bool a_ok[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = lm0 + arow0 + 128 * j + 64 * s + lr;
            a_ok[s][j] = m < p.M;
            if (MODE == 2) {
                const int hw = p.OH * p.OW;
                const int b = m / hw, rem = m - b * hw;
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                a_ih[s][j] = oh * p.stride - p.pt;
                a_iw[s][j] = ow * p.stride - p.pl;
                a_img[s][j] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.Cin * 2u;
                a_off[s][j] = kOob;
                a_off2[s][j] = 0;
            } else {
                a_off[s][j] = a_ok[s][j] ? (unsigned)m * (unsigned)p.lda1 * 2u + c8 * 16u : kOob;
                a_off2[s][j] = a_ok[s][j] ? (unsigned)m * (unsigned)p.lda2 * 2u + c8 * 16u : kOob;
                a_ih[s][j] = a_iw[s][j] = 0;
                a_img[s][j] = 0;
            }
        }
    int cs[2] = {0, 0}, ftap[2] = {0, 0};                            // conv: channel step inside the tap / tap of the next issue, per set
    const int csteps = MODE == 2 ? p.Cin / BK : 1;
    const int k1_steps = p.K1 / BK;                                  // GEMM: K-tiles served by A1 (K1 % 64 == 0 unless K1 == K)
    const int nk = (p.K + BK - 1) / BK;
    const bool ktail = (p.K % BK) != 0;

    // One half-tile = two DMA pieces per wave (J = 0, 1).  Every index below is a compile-time constant: a runtime-indexed
    // offset array would live in scratch, and its reload's vmcnt(0) would drain the DMA pipeline in every phase.
    // SPLIT: piece 0 goes out in the phase's load segment, piece 1 from inside the MFMA cluster.
    unsigned so_a[2] = {0, 0};                                       // conv: scalar K offset of the half-tile being issued, per set
    auto fire_w = [&](auto setc, auto piecec, int kt, int buf) {
        constexpr int S = decltype(setc)::value, J = decltype(piecec)::value;
        if (kt >= nk) return;
        bload16(rs_w, smem + buf * BUFH + (256 + wrow0 + 32 * S + 128 * J) * BK, w_off[S][J], (unsigned)kt * (BK * 2));
    };
    auto prep_a = [&](auto setc, int kt) {                           // conv: (re)compute the tap's offsets, advance (channel step, tap)
        constexpr int S = decltype(setc)::value;
        if (MODE != 2 || kt >= nk) return;
        if (cs[S] == 0) {
            const int ky = p.taps2 ? ftap[S] >> 1 : (ftap[S] * 11) >> 5, kx = p.taps2 ? ftap[S] & 1 : ftap[S] - 3 * ky;
            const int hlim = p.H << p.ups, wlim = p.W << p.ups;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ih = a_ih[S][j] + ky, iw = a_iw[S][j] + kx;
                const bool ok = a_ok[S][j] && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
                a_off[S][j] = ok ? a_img[S][j] + (unsigned)((ih >> p.ups) * p.W + (iw >> p.ups)) * (unsigned)p.Cin * 2u + c8 * 16u : kOob;
            }
        }
        so_a[S] = (unsigned)cs[S] * (BK * 2);
        if (++cs[S] == csteps) { cs[S] = 0; ++ftap[S]; }
    };
    auto fire_a = [&](auto setc, auto piecec, int kt, int buf) {
        constexpr int S = decltype(setc)::value, J = decltype(piecec)::value;
        if (kt >= nk) return;
        half_t* dst = smem + buf * BUFH + (arow0 + 64 * S + 128 * J) * BK;
        if (MODE == 2) {
            bload16(rs_a1, dst, a_off[S][J], so_a[S]);
        } else if (ktail && kt == nk - 1) {                           // last, partial K-tile: columns >= K read as zero
            bload16(rs_a1, dst, kt * BK + c8 * 8 < p.K ? a_off[S][J] : kOob, (unsigned)kt * (BK * 2));
        } else if (kt < k1_steps || k1_steps == 0) {
            bload16(rs_a1, dst, a_off[S][J], (unsigned)kt * (BK * 2));
        } else {
            bload16(rs_a2, dst, a_off2[S][J], (unsigned)(kt - k1_steps) * (BK * 2));
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto issue_w = [&](auto setc, int kt, int buf) { fire_w(setc, P0{}, kt, buf); fire_w(setc, P1{}, kt, buf); };
    auto issue_a = [&](auto setc, int kt, int buf) { prep_a(setc, kt); fire_a(setc, P0{}, kt, buf); fire_a(setc, P1{}, kt, buf); };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    f32x4 acc[4][8];                                                 // [channel tile][row tile] of the wave's 64 x 128 outputs
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: K-tile 0 complete + the first three half-tiles of K-tile 1
    issue_w(S0{}, 0, 0); issue_a(S0{}, 0, 0); issue_w(S1{}, 0, 0); issue_a(S1{}, 0, 0);
    issue_w(S0{}, 1, 1); issue_a(S0{}, 1, 1); issue_w(S1{}, 1, 1);
    if (nk > 1) wait_vm_barrier<6>(); else wait_vm_barrier<0>();
    if (wm == 1) __builtin_amdgcn_s_barrier();                      // group 1 runs one barrier behind group 0 from here on

    const int fr = lane & 15, fq = lane >> 4;
    // fragment addresses: row & 7 == fr & 7 for every fragment row (all other terms are multiples of 8)
    const int frag0 = fr * BK + ((fq ^ (fr & 7)) << 3);              // chunk fq      (k 0-31)
    const int frag1 = fr * BK + (((4 + fq) ^ (fr & 7)) << 3);        // chunk 4 + fq  (k 32-63)
    const int a_base = wm * 128 * BK, w_base = (256 + wn * 64) * BK;

    f16x8 fa[4][2], fw0[2][2], fw1[2][2];

    // 16 MFMAs of one quadrant: channel tiles ci0, ci0 + 1 (fragments fw) x row tiles rj0 .. rj0 + 3 (fragments fa); with SPLIT
    // the half-tile's second DMA piece is issued behind the first four MFMAs
#define FIE_QUADRANT(FW, CI0, RJ0, LATE)                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
            acc[CI0 + i][RJ0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FW[i][0], fa[j][0], acc[CI0 + i][RJ0 + j], 0, 0, 0); \
            acc[CI0 + i][RJ0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FW[i][1], fa[j][1], acc[CI0 + i][RJ0 + j], 0, 0, 0); \
            if (SPLIT && i == 0 && j == 1) {                                                                               \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
                LATE;                                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
            }                                                                                                              \
        }                                                                                                                  \
    }                                                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                                         \
    __builtin_amdgcn_sched_barrier(0);

    auto ktile = [&](auto bufc, int t) {
        constexpr int BUF = decltype(bufc)::value;
        const half_t* sa = smem + BUF * BUFH + a_base;
        const half_t* sw = smem + BUF * BUFH + w_base;
        // ------------------------------------------------------------------ phase 1: quadrant (a0, w0)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fw0[i][0] = *reinterpret_cast<const f16x8*>(sw + i * 16 * BK + frag0);
            fw0[i][1] = *reinterpret_cast<const f16x8*>(sw + i * 16 * BK + frag1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fa[j][0] = *reinterpret_cast<const f16x8*>(sa + j * 16 * BK + frag0);
            fa[j][1] = *reinterpret_cast<const f16x8*>(sa + j * 16 * BK + frag1);
        }
        prep_a(S1{}, t + 1);
        fire_a(S1{}, P0{}, t + 1, BUF ^ 1);
        if (!SPLIT) fire_a(S1{}, P1{}, t + 1, BUF ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FIE_QUADRANT(fw0, 0, 0, fire_a(S1{}, P1{}, t + 1, BUF ^ 1))
        __builtin_amdgcn_s_barrier();
        // ------------------------------------------------------------------ phase 2: quadrant (a0, w1)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fw1[i][0] = *reinterpret_cast<const f16x8*>(sw + (32 + i * 16) * BK + frag0);
            fw1[i][1] = *reinterpret_cast<const f16x8*>(sw + (32 + i * 16) * BK + frag1);
        }
        fire_w(S0{}, P0{}, t + 2, BUF);
        if (!SPLIT) fire_w(S0{}, P1{}, t + 2, BUF);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FIE_QUADRANT(fw1, 2, 0, fire_w(S0{}, P1{}, t + 2, BUF))
        __builtin_amdgcn_s_barrier();
        // ------------------------------------------------------------------ phase 3: quadrant (a1, w1)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fa[j][0] = *reinterpret_cast<const f16x8*>(sa + (64 + j * 16) * BK + frag0);
            fa[j][1] = *reinterpret_cast<const f16x8*>(sa + (64 + j * 16) * BK + frag1);
        }
        prep_a(S0{}, t + 2);
        fire_a(S0{}, P0{}, t + 2, BUF);
        if (!SPLIT) fire_a(S0{}, P1{}, t + 2, BUF);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FIE_QUADRANT(fw1, 2, 4, fire_a(S0{}, P1{}, t + 2, BUF))
        __builtin_amdgcn_s_barrier();
        // ------------------------------------------------------------------ phase 4: quadrant (a1, w0), w0 still in registers
        fire_w(S1{}, P0{}, t + 2, BUF);
        if (!SPLIT) fire_w(S1{}, P1{}, t + 2, BUF);
        // K-tile t+1 (read from the next phase on) must have landed: only W0, A0 and the pieces of W1 of K-tile t+2 issued so far
        // may still be in flight (3 half-tiles x 2 pieces; with SPLIT this phase's second piece is not out yet: 5)
        if (t + 2 < nk) wait_vm<SPLIT ? 5 : 6>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        FIE_QUADRANT(fw0, 0, 4, fire_w(S1{}, P1{}, t + 2, BUF))
        __builtin_amdgcn_s_barrier();
    };
#undef FIE_QUADRANT

    for (int t = 0; t < nk; t += 2) {
        ktile(S0{}, t);
        if (t + 1 < nk) ktile(S1{}, t + 1);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                      // pairs with group 1's last barrier

    // two column halves: with all 32 fragments in one pass the epilogue's temporaries do not fit beside 128 accumulator registers
    epilogue<8, 2, 128, 64, true>(p, reinterpret_cast<f32x4(&)[2][8]>(acc[0]), m0, n0, wm, wn, lane);
    epilogue<8, 2, 128, 64, true>(p, reinterpret_cast<f32x4(&)[2][8]>(acc[2]), m0, n0 + 32, wm, wn, lane);
}

constexpr int kLds8 = 2 * BUFH * (int)sizeof(half_t);               // 128 KiB

}  // namespace

int fie_gemm8_init(void) {
    hipError_t e = hipSuccess;
    for (const void* f : {reinterpret_cast<const void*>(&gemm8_kernel<0, 1>), reinterpret_cast<const void*>(&gemm8_kernel<2, 1>),
                          reinterpret_cast<const void*>(&gemm8_kernel<0, 0>), reinterpret_cast<const void*>(&gemm8_kernel<2, 0>)})
        if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kLds8);
    if (e != hipSuccess) {
        fie_set_error("gemm8: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return FIE_EHIP;
    }
    return FIE_OK;
}

int fie_launch_gemm8(fie_ctx* ctx, const GemmArgs& a, int conv, int split) {
    const dim3 grid((unsigned)(a.nbm * a.nbn * (a.oscat == 2 ? 4 : 1)));
    if (conv && split) fie_launch(ctx, (gemm8_kernel<2, 1>), grid, dim3(512), kLds8, a);
    else if (conv) fie_launch(ctx, (gemm8_kernel<2, 0>), grid, dim3(512), kLds8, a);
    else if (split) fie_launch(ctx, (gemm8_kernel<0, 1>), grid, dim3(512), kLds8, a);
    else fie_launch(ctx, (gemm8_kernel<0, 0>), grid, dim3(512), kLds8, a);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}
