// K1, halo-resident form ("LDS-staged halo tiles"): the stride-1 3x3 convolution of the VAE
// (upstream diffusers models/resnet.py ResnetBlock2D.conv1 / conv2, models/autoencoders/vae.py) for NHWC fp16 maps whose height and width are
// multiples of 16 and Cin % 64 == 0.  Same entry (fie_conv3x3_nhwc_f16), same weights (fie_pack_conv3x3_f16), same epilogue as the im2col kernels
// of gemm_conv.hip / gemm8.hip; what changes is how the ACTIVATIONS reach the matrix cores.
//
// Why.  The im2col kernels stage every input pixel NINE times (once per tap) through the per-CU L2 -> LDS path, which sustains ~33-36 B/clk per CU
// (MI355X_MICROARCH.md, "Indexed rows: gather into LDS": 66-73 GB/s per CU).  A 256x128 im2col tile stages (256 + 128) x 128 B per 64-deep K-step
// for 1024 MFMA cycles = 48 B/clk: the load path, not the MFMA, paces them (0.26-0.33 of peak on the 1024^2 x 128 convs, profiles/r03_per_shape_roofline.md).
// Here a block owns a 16 x 16 PATCH of output pixels of one image (the GEMM's 256 rows) and 128 output channels.  For each 64-channel chunk of
// the input it loads the patch's 18 x 18 halo ONCE (41.5 KB, double buffered: chunk c + 1 streams in under chunk c's nine taps) and runs the nine
// taps as nine K-steps whose activation fragments are read straight from the halo at a tap offset; only the weights (16 KB per K-step) go through
// a ring.  Staged bytes per K-step: 16 KB + 41.5 / 9 KB = 20.6 KB = 20 B/clk at the MFMA rate: the kernel is MFMA-paced.
//
// K order.  chunk-major (chunk, tap) instead of the im2col kernels' (tap, chunk): the fp32 sums are added in another order, so outputs agree with
// the other conv kernels to rounding, not bit for bit (the tile CHOICE among the im2col kernels never changes bits; this one does, like split-K).
//
// Geometry.  512 threads = 8 waves as 4 (patch-row groups) x 2 (channel halves); a wave owns 4 patch rows x 64 channels = 4 x 4 MFMA tiles of
// v_mfma_f32_16x16x32_f16 with swapped operands (weights = A operand, activations = B operand: a lane holds 4 consecutive output channels of
// one pixel, the layout gemm_common.h's epilogue stores).  Fragment j of a wave = the 16 pixels of patch row wm * 4 + j, so its rows are OW apart
// in the [B * OH * OW, N] output: GemmArgs::frag_ld.
//
// LDS (133,120 B): two halo buffers of 41 pieces x 1 KiB (324 pixels x 128 B, piece = 8 pixels), a 1 KiB dump for the seven piece slots that
// have no piece (8 waves x 6 slots = 48 >= 41), three weight stages of 128 rows x 128 B.  Halo image: pixel p = hy * 18 + hx at byte p * 128, its
// eight 16-byte channel chunks XOR-swizzled by (hx & 7) -- applied on the SOURCE chunk of the LDS-DMA (the destination of a wave's DMA is linear)
// and on the fragment reads; the 16 lanes of a ds_read_b128 group then cover all 16 slots of a 256-byte bank row for every tap (checked by
// enumeration: tools/halo_bank_check.py).  Weight stages: the ring kernels' image (row r at r * 128, chunks swizzled by r & 7).
//
// Schedule.  The two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER APART, as in gemm8.hip: while one group issues its
// 32 MFMAs of a K-step the other one reads the next K-step's 16 fragments and issues its share of the DMA.  Per wave and K-step (tap t of chunk c):
//       [load segment]  DMA: 2 weight pieces of K-step kt + 2, one halo piece of chunk c + 1 (t < 6) | 16 ds_read_b128 | lgkmcnt(0) | vmcnt(N) | s_barrier
//       [MFMA segment]  32 MFMA | s_barrier
// Hazards (cdna_hip_programming.md: "Read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  the counted wait before the load segment's barrier retires this wave's pieces of K-step kt + 1 (issued a whole K-step earlier: two in
//        flight) and -- loads complete in issue order -- every halo piece issued before them; K-step kt + 1 is first read after a barrier that
//        every wave of both groups reaches only after its own wait.  N = the DMA instructions this wave issued behind those pieces: the nine taps
//        are unrolled so that N is a compile-time constant per tap (the last chunk, which prefetches no halo and runs out of weights, is its own
//        instantiation).
//   WAR  weight stage (kt + 2) % 3 was last read in K-step kt - 1, halo buffer (c + 1) & 1 in chunk c - 1; every wave drains its LDS reads
//        (lgkmcnt(0)) before the load segment's barrier, so the first refill, issued behind that barrier, cannot overtake a read.
#include "gemm_common.h"

using namespace fie_gemm;

namespace {

constexpr int HP = 18;                          // halo pixels per side of a 16 x 16 patch (3x3 taps, stride 1)
constexpr int HPIECES = (HP * HP + 7) / 8;      // 41 DMA pieces of 8 pixels x 128 B
constexpr int HALO_B = HPIECES * 1024;          // 41,984 bytes per halo buffer
constexpr int HSLOTS = 6;                       // halo piece slots per wave and chunk: taps 0..5 issue one each
constexpr int BNH = 128;                        // output channels per block
constexpr int STH = 3;                          // weight stages
constexpr int WSTAGE_B = BNH * 128;             // 16,384 bytes
constexpr int DUMP_OFF = 2 * HALO_B;
constexpr int WRING_OFF = DUMP_OFF + 1024;
constexpr int kLdsHalo = WRING_OFF + STH * WSTAGE_B;      // 133,120 bytes
constexpr int RWH = BNH / 64;                   // weight pieces per wave and K-step (16 pieces of 8 rows over 8 waves)

// DMA instructions a wave issues in the load segment of tap T: weights of K-step kt + 2 (none in the last chunk's taps 7, 8: past the end),
// one halo piece of the next chunk in taps 0..5 (none in the last chunk)
constexpr int n_w(bool last, int t) { return last && t + 2 > 8 ? 0 : RWH; }
constexpr int n_h(bool last, int t) { return !last && t < HSLOTS ? 1 : 0; }
// vmcnt that retires the weights of K-step kt + 1 (issued in tap T - 1's load segment, before that segment's halo piece): everything this wave
// issued behind them may stay in flight.  Tap -1 = tap 8 of the previous chunk (never the last chunk): no halo piece.
constexpr int n_wait(bool last, int t) { return (t > 0 ? n_h(last, t - 1) : 0) + n_w(last, t) + n_h(last, t); }

template <bool STAMP>
__global__ __launch_bounds__(512) void conv_halo_kernel(GemmArgs p) {
    constexpr int FM = 4, FN = 4, WM = 64, WN = 64;
    extern __shared__ __attribute__((aligned(16))) half_t smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                   // 0: leads, 1: one barrier behind
    const int wm = wave & 3, wn = wave >> 2;     // patch rows wm * 4 .. + 3, channels wn * 64 .. + 63

    // ---- tile: column tiles fastest (they share the patch's halo in L2), then patches row-major within an image
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % p.nbn, patch = bid / p.nbn;
    const int ppr = p.OW >> 4, ppi = (p.OH >> 4) * ppr;
    const int b = patch / ppi, pr = patch - b * ppi;
    const int py = pr / ppr, px = pr - py * ppr;
    const int y0 = py << 4, x0 = px << 4;
    const int m0 = (b * p.OH + y0) * p.OW + x0, n0 = nt * BNH;

    const int live = p.probe == 1 ? 0 : 1;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)p.a1_bytes * live, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, (int)p.w_bytes * live, 0x00020000);

    // ---- halo DMA: slot i of this wave = piece q = wave + 8 i (8 pixels x 128 B); lane = (pixel lane >> 3, chunk position lane & 7)
    unsigned h_off[HSLOTS];
#pragma unroll
    for (int i = 0; i < HSLOTS; ++i) {
        const int q = wave + 8 * i;
        const int pl = q * 8 + (lane >> 3);
        const int hy = pl / HP, hx = pl - hy * HP;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool ok = q < HPIECES && pl < HP * HP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const unsigned chunk = (unsigned)((lane & 7) ^ (hx & 7));
        h_off[i] = ok ? ((unsigned)(b * p.H + iy) * (unsigned)p.W + (unsigned)ix) * (unsigned)p.Cin * 2u + chunk * 16u : kOob;
    }
    auto halo_dst = [&](int i, int buf) -> half_t* {          // slots past the last piece write zeros into the dump
        const int q = wave + 8 * i;
        return reinterpret_cast<half_t*>(lds + (q < HPIECES ? buf * HALO_B + q * 1024 : DUMP_OFF));
    };
    // ---- weight DMA: piece wave + 8 i = rows (wave + 8 i) * 8 .. + 7 of the block's 128 weight rows
    const int lr = lane >> 3;
    const unsigned w_base = (unsigned)(n0 + wave * 8 + lr) * (unsigned)p.ldw * 2u + (unsigned)((lane & 7) ^ lr) * 16u;
    const unsigned w_step = 64u * (unsigned)p.ldw * 2u;
    auto issue_w = [&](int stage, unsigned soff) {
#pragma unroll
        for (int i = 0; i < RWH; ++i)
            bload16(rs_w, reinterpret_cast<half_t*>(lds + WRING_OFF + stage * WSTAGE_B + (wave + 8 * i) * 1024), w_base + (unsigned)i * w_step, soff);
    };
    const unsigned cin2 = (unsigned)p.Cin * 2u;                // bytes between two taps' columns of a packed weight row

    // ---- fragment read offsets
    const int fr = lane & 15, fq = lane >> 4;
    unsigned a_rd[3][2];                                       // [kx][k half]: byte offset inside a halo buffer of pixel (wm * 4, fr + kx), chunk kh * 4 + fq
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
            a_rd[kx][kh] = (unsigned)((wm * FM * HP + fr + kx) * 128 + (((kh * 4 + fq) ^ ((fr + kx) & 7)) << 4));
    unsigned w_rd[2];                                          // [k half]: byte offset inside a weight stage of row wn * 64 + fr
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) w_rd[kh] = (unsigned)(2 * lds_off(wn * WN + fr, kh * 4 + fq));

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    unsigned seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;     // STAMP: [0] DMA issue, [1] reads issued + landed, [2] wait + barrier, [3] MFMA issue, [4] barrier, [5] prologue, [6] epilogue
    auto stamp = [&](int i) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
            seg[i] += t - tprev;
            tprev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (STAMP) tprev = (unsigned)__builtin_amdgcn_s_memtime();

    const int nchunks = p.Cin / BK;
    // ---- prologue: the whole halo of chunk 0, the weights of K-steps 0 and 1
#pragma unroll
    for (int i = 0; i < HSLOTS; ++i) bload16(rs_a, halo_dst(i, 0), h_off[i], 0u);
    issue_w(0, 0u);
    issue_w(1, nchunks * 9 > 1 ? cin2 : 0u);                   // tap 1 of chunk 0 (Cin >= 64: there are always nine K-steps)
    wait_vm_barrier<RWH>();                                    // halo 0 and K-step 0 landed (K-step 1 may be in flight)
    if (grp == 1) __builtin_amdgcn_s_barrier();                // group 1 runs one barrier behind from here on
    stamp(5);

    f16x8 fw[2][FN], fa[2][FM];
    // one K-step = tap T of chunk c (the caller passes c; T and LAST are compile-time)
    auto kstep = [&](auto lastc, auto tapc, int c, unsigned hb) {
        constexpr bool LAST = decltype(lastc)::value;
        constexpr int T = decltype(tapc)::value;
        constexpr int KY = T / 3, KX = T % 3;
        constexpr int STAGE = T % STH, FILL = (T + 2) % STH;    // 9 % 3 == 0: the stage of a K-step depends on its tap only
        // ---- load segment
        if constexpr (n_w(LAST, T) > 0) {
            constexpr int T2 = (T + 2) % 9;
            const int c2 = T + 2 > 8 ? c + 1 : c;
            issue_w(FILL, (unsigned)T2 * cin2 + (unsigned)c2 * (BK * 2));
        }
        if constexpr (n_h(LAST, T) > 0) bload16(rs_a, halo_dst(T, (c + 1) & 1), h_off[T], (unsigned)(c + 1) * (BK * 2));
        stamp(0);
        const char* const hal = lds + hb;
        const char* const wst = lds + WRING_OFF + STAGE * WSTAGE_B;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int i = 0; i < FN; ++i) fw[kh][i] = *reinterpret_cast<const f16x8*>(wst + w_rd[kh] + i * 16 * 128);
#pragma unroll
            for (int j = 0; j < FM; ++j) fa[kh][j] = *reinterpret_cast<const f16x8*>(hal + a_rd[KX][kh] + (j + KY) * HP * 128);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(1);
        wait_vm_barrier<n_wait(LAST, T)>();
        stamp(2);
        // ---- MFMA segment
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[kh][i], fa[kh][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        stamp(3);
        __builtin_amdgcn_s_barrier();
        stamp(4);
    };
    auto chunk = [&](auto lastc, int c) {
        const unsigned hb = (c & 1) ? (unsigned)HALO_B : 0u;
        kstep(lastc, std::integral_constant<int, 0>{}, c, hb);
        kstep(lastc, std::integral_constant<int, 1>{}, c, hb);
        kstep(lastc, std::integral_constant<int, 2>{}, c, hb);
        kstep(lastc, std::integral_constant<int, 3>{}, c, hb);
        kstep(lastc, std::integral_constant<int, 4>{}, c, hb);
        kstep(lastc, std::integral_constant<int, 5>{}, c, hb);
        kstep(lastc, std::integral_constant<int, 6>{}, c, hb);
        kstep(lastc, std::integral_constant<int, 7>{}, c, hb);
        kstep(lastc, std::integral_constant<int, 8>{}, c, hb);
    };
    for (int c = 0; c + 1 < nchunks; ++c) chunk(std::false_type{}, c);
    chunk(std::true_type{}, nchunks - 1);
    if (grp == 0) __builtin_amdgcn_s_barrier();                // pairs with group 1's last barrier

    epilogue<FM, FN, WM, WN, true>(p, acc, m0, n0, wm, wn, lane);
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(6);
        if (p.stamps && lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p.stamps[((size_t)bid * 8 + wave) * 8 + i] = seg[i];
        }
    }
}

}  // namespace

int fie_conv_halo_init(void) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalo);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalo);
    if (e != hipSuccess) {
        fie_set_error("conv_halo: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return FIE_EHIP;
    }
    return FIE_OK;
}

// Shapes the halo-resident kernel takes: the plain same-size conv (stride 1, zero padding 1, no up-sampling gather, no 1x1 side inputs, fp16
// weights) on maps whose height and width are multiples of 16, Cin % 64 == 0, operands the 32-bit buffer offsets reach (checked by the caller: dma_ok)
bool fie_conv_halo_ok(const GemmArgs& a) {
    return a.stride == 1 && !a.ups && !a.taps2 && !a.oscat && !a.A2 && !a.A3 && a.pt == 1 && a.pl == 1 && a.H == a.OH && a.W == a.OW && a.OH % 16 == 0 &&
           a.OW % 16 == 0 && a.Cin % BK == 0 && a.Cin >= BK && !a.w_scale && a.K == 9 * a.Cin && !a.out_f8 && a.splitk <= 1;
}

int fie_launch_conv_halo(fie_ctx* ctx, GemmArgs& a, int stamped) {
    FIE_REQUIRE(fie_conv_halo_ok(a), "tile code 71 (halo-resident conv): stride-1 same-size 3x3 conv with H, W %% 16 == 0 and Cin %% 64 == 0 only");
    a.frag_ld = a.OW;
    a.nbn = (a.N + BNH - 1) / BNH;
    a.nbm = (a.M / (a.OH * a.OW)) * (a.OH >> 4) * (a.OW >> 4);
    const dim3 grid((unsigned)(a.nbm * a.nbn));
    if (stamped) fie_launch(ctx, (conv_halo_kernel<true>), grid, dim3(512), kLdsHalo, a);
    else fie_launch(ctx, (conv_halo_kernel<false>), grid, dim3(512), kLdsHalo, a);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}
