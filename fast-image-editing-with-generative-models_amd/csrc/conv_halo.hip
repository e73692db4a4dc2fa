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
// LDS (133,120 B): three weight stages of 128 rows x 128 B, two halo buffers of 41 pieces x 1 KiB (324 pixels x 128 B, piece = 8 pixels), a 1 KiB
// dump for the seven piece slots that have no piece (8 waves x 6 slots = 48 >= 41).  Halo image: pixel p = hy * 18 + hx at byte p * 128, its
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
// LDS: [three weight stages][halo 0][halo 1][dump].  The ring comes FIRST so that stage * 16 KiB + fragment * 2 KiB fits the 16-bit offset field of
// ds_read_b128: two per-lane base registers serve every weight fragment read (behind the halos the six stage bases lived in six registers)
constexpr int WRING_OFF = 0;
constexpr int HALO_OFF = STH * WSTAGE_B;                  // 49,152
constexpr int DUMP_OFF = HALO_OFF + 2 * HALO_B;
constexpr int kLdsHalo = DUMP_OFF + 1024;                 // 133,120 bytes
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
        return reinterpret_cast<half_t*>(lds + (q < HPIECES ? HALO_OFF + buf * HALO_B + q * 1024 : DUMP_OFF));
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
        const char* const hal = lds + HALO_OFF + hb;
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

    epilogue<FM, FN, WM, WN, true, true>(p, acc, m0, n0, wm, wn, lane);
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(6);
        if (p.stamps && lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p.stamps[((size_t)bid * 8 + wave) * 8 + i] = seg[i];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// v2 (tile code 72): the same K loop in a PERSISTENT block with DEFERRED STORES.  Stamps on v1 (profiles/r04_halo_conv_v1_first_run.log): of a
// 1024^2 x 128 -> 128 tile's ~37 k cycles the K loop takes 24 k; the prologue (a cold 41 KB halo + 32 KB of weights: latency) 4.4-5.7 k and the
// epilogue 7.4-9.6 k -- sixteen 8-byte-per-lane store instructions per wave that every CU issues at the same moment (blocks run in step), store-issue
// bound (MI355X_MICROARCH.md, cycle constants: "attention epilogue store tail").  Here a block walks tiles bid, bid + G, bid + 2 G, ... and
//   * the LAST chunk of a tile prefetches the NEXT tile's first halo (its six piece slots) and first two weight stages (9 K-steps per chunk and
//     three stages: the ring position carries over), so the next K loop starts without a prologue;
//   * at the end of a tile the accumulators get bias / row bias / activation / residual, the GroupNorm sums are written, and the results are
//     packed to f16 into 32 registers; the sixteen stores are then issued TWO PER K-STEP in the load segments of the next tile's first chunk
//     (taps 0..7), where they ride on the counted waits like any DMA piece: no wave ever waits for a store, and the chip's write traffic is
//     spread over the K loop instead of coming in bursts;
//   * RES: the residual tile (the resnet's identity shortcut, upstream resnet.py: output = conv2(h) + input) is loaded sixteen 8-byte pieces per
//     lane inside the MFMA segment of the tile's LAST K-step, into the registers the deferred outputs left in the first chunk (Cin >= 128: the
//     first chunk is never the last).
// Every vector-memory instruction between two counted waits is issued unconditionally (out-of-range lanes and absent operands go through the
// descriptors' range check), so the vmcnt immediates stay compile-time constants per (chunk kind, tap); after the one irregular event (the row
// bias of another image) the wave drains completely, which is always safe.
// The column tile of a block is fixed (the grid is a multiple of the column tiles): one weight stream and one bias vector per block.
constexpr int K_FIRST = 0, K_MID = 1, K_LAST_F = 2, K_LAST_N = 3, K_LASTREG_S = 4;      // chunk kinds: first (stores), middle, last with / without a next tile to prefetch, last nine-tap chunk in front of side steps
constexpr int n2_w(int kind, int t) { return kind == K_LAST_N && t + 2 > 8 ? 0 : RWH; }
constexpr int n2_h(int kind, int t) { return kind != K_LAST_N && t < HSLOTS ? 1 : 0; }
// VAR (switches of v2): 2 = ST16: sixteen-byte stores (two fragments' halves exchanged by v_permlane16_swap: 8 stores per wave and tile, one per tap);
// 4 = STM: the stores are issued from inside the MFMA segment; 8 = IM: EVERY vector-memory instruction of a K-step is issued from inside the MFMA segment
constexpr int n2_st(int var, int kind, int t) { return kind == K_FIRST && t < 8 ? ((var & 2) ? 1 : 2) : 0; }
// LOADS (LDS-DMA pieces) issued BEHIND the weights in tap t.  STORES are deliberately not counted: measured in round 4
// (tools/halo_race.py; profiles/r04_halo_conv_stores_do_not_keep_vmcnt_order.md), a store does not reliably keep its place in the vmcnt order
// relative to LDS-DMA loads -- with the deferred stores counted, a wait let a weight piece through unretired about once in 500 launches (8-byte
// stores of whole lanes; range-dropped dummies and the mostly-dropped GroupNorm stores far more often).  Counting loads only is safe either way:
// stores that retire early are not in the count, stores still pending only make the wait stricter.
constexpr int n2_tail(bool res, int kind, int t) { return n2_h(kind, t); }
// vmcnt that retires the weights of K-step kt + 1 (issued first among tap t - 1's instructions; tap 8 of every kind issues no load behind its
// weights).  IM: tap t issues nothing before its wait (its instructions go out in the MFMA segment behind it)
constexpr int n2_wait(int var, bool res, int kind, int t) {
    return (t > 0 ? n2_tail(res, kind, t - 1) : 0) + ((var & 8) ? 0 : n2_w(kind, t) + n2_tail(res, kind, t));
}
static_assert(n2_tail(true, K_LAST_F, 8) == 0 && n2_tail(true, K_FIRST, 8) == 0 && n2_st(0, K_FIRST, 8) == 0, "tap 8 carries nothing behind its weights");

template <bool RES, bool GN, bool RB, int VAR, bool STAMP, bool SIDE = false, bool GNA = false>
__global__ __launch_bounds__(512) void conv_halo2_kernel(GemmArgs p) {
    constexpr int FM = 4, FN = 4, WN = 64;
    constexpr bool ST16 = (VAR & 2) != 0, STM = (VAR & 4) != 0, IM = (VAR & 8) != 0;
    // DEFER: the GroupNorm sums of a tile are taken from its packed f16 outputs while the NEXT tile's first chunk runs, a few VALU instructions behind
    // every MFMA of K-steps 0..7 (the wave's issue slot is idle 12 of every 16 cycles there), instead of ~1 400 cycles of tile-end math that both wave
    // groups wait for; the 16-byte stores' lane swap moves to the store slot for the same reason (the sums want the unswapped layout)
    constexpr bool DEFER = ST16 && STM && !IM;
    static_assert(!(SIDE && (RES || RB)), "1x1 side inputs belong to a resnet's conv2 + shortcut: no residual, no row bias");
    extern __shared__ __attribute__((aligned(16))) half_t smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave & 3, wn = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;

    const int G = (int)gridDim.x;                               // a multiple of p.nbn: tile % nbn is the same for all of a block's tiles
    const int ntiles = p.nbm * p.nbn;
    const int bid = xcd_remap(blockIdx.x, G);
    const int n0 = (bid % p.nbn) * BNH;
    const int ppr = p.OW >> 4, ppi = (p.OH >> 4) * ppr;
    const int ncol = n0 + wn * WN + fq * 4;                     // this lane's first output channel

    const int live = p.probe == 1 ? 0 : 1;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)p.a1_bytes * live, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, (int)p.w_bytes * live, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((int64_t)(p.M - 1) * p.ldc + p.N) * 2), 0x00020000);

    // ---- tile geometry: first output row, image, the patch's corner
    auto geom = [&](int tile, int& m0, int& img, int& y0, int& x0) {
        const int patch = tile / p.nbn;
        const int b = patch / ppi, pr = patch - b * ppi;
        const int py = pr / ppr, px = pr - py * ppr;
        y0 = py << 4; x0 = px << 4;
        m0 = (b * p.OH + y0) * p.OW + x0;
        img = b;
    };
    // halo piece slot i of this wave = piece q = wave + 8 i (8 pixels x 128 B); lane = (pixel lane >> 3, chunk position lane & 7).  Per slot ONE
    // packed register (hy << 8 | hx; hy = 255: no pixel); the offsets are recomputed from it at every use behind an opaque asm -- left to
    // itself the compiler hoists two dozen loop-invariant partial results per lane out of the tile loop and spills the deferred outputs instead
    unsigned hq[HSLOTS];
#pragma unroll
    for (int i = 0; i < HSLOTS; ++i) {
        const int q = wave + 8 * i;
        const int pl = q * 8 + (lane >> 3);
        const int hy = pl / HP, hx = pl - hy * HP;
        hq[i] = q < HPIECES && pl < HP * HP ? (unsigned)((hy << 8) | hx) : 0xFF00u;
    }
    auto halo_off_px = [&](int i, int img, int y0, int x0, unsigned pix_bytes) -> unsigned {      // the same pixel of an image whose pixels are pix_bytes apart (a 1x1 side input)
        unsigned t = hq[i];
        asm volatile("" : "+v"(t));
        const int hy = (int)(t >> 8), hx = (int)(t & 255u);
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool ok = hy < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned chunk = (unsigned)((lane & 7) ^ (hx & 7));
        return ok ? ((unsigned)(img * p.H + iy) * (unsigned)p.W + (unsigned)ix) * pix_bytes + chunk * 16u : kOob;
    };
    auto halo_off = [&](int i, int img, int y0, int x0) -> unsigned {
        unsigned t = hq[i];
        asm volatile("" : "+v"(t));
        const int hy = (int)(t >> 8), hx = (int)(t & 255u);
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool ok = hy < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned chunk = (unsigned)((lane & 7) ^ (hx & 7));
        return ok ? ((unsigned)(img * p.H + iy) * (unsigned)p.W + (unsigned)ix) * (unsigned)p.Cin * 2u + chunk * 16u : kOob;
    };
    auto halo_dst = [&](int i, int buf) -> half_t* {
        const int q = wave + 8 * i;
        return reinterpret_cast<half_t*>(lds + (q < HPIECES ? HALO_OFF + buf * HALO_B + q * 1024 : DUMP_OFF));
    };
    const int lr = lane >> 3;
    const unsigned w_base = (unsigned)(n0 + wave * 8 + lr) * (unsigned)p.ldw * 2u + (unsigned)((lane & 7) ^ lr) * 16u;
    const unsigned w_step = 64u * (unsigned)p.ldw * 2u;
    auto issue_w = [&](int stage, unsigned soff) {
#pragma unroll
        for (int i = 0; i < RWH; ++i)
            bload16(rs_w, reinterpret_cast<half_t*>(lds + WRING_OFF + stage * WSTAGE_B + (wave + 8 * i) * 1024), w_base + (unsigned)i * w_step, soff);
    };
    const unsigned cin2 = (unsigned)p.Cin * 2u;

    unsigned a_rd[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
            a_rd[kx][kh] = (unsigned)((wm * FM * HP + fr + kx) * 128 + (((kh * 4 + fq) ^ ((fr + kx) & 7)) << 4));
    unsigned w_rd[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) w_rd[kh] = (unsigned)(2 * lds_off(wn * WN + fr, kh * 4 + fq));

    // bias of this lane's channels (zeros when absent): loaded once, a block keeps its column tile
    u32x2 bv[FN], rbv[FN];
    {
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.bias), 0, p.bias ? p.N * 2 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            bv[i] = __builtin_amdgcn_raw_buffer_load_b64(rs_b, (unsigned)(ncol + i * 16) * 2u, 0, 0);
            rbv[i] = (u32x2){0u, 0u};
        }
    }
    // byte offset of (first fragment row, this lane's first channel) in a [M, ld] matrix; fragment row j is j * OW * ld * 2 further
    auto row_off16 = [&](int m0, int64_t ld) -> unsigned {      // ST16: this lane stores pixel row (fq & 1) of a fragment pair, channels (fq >> 1) * 8 .. + 7 of the fragment
        return ((unsigned)(m0 + (wm * FM + (fq & 1)) * p.OW + fr) * (unsigned)ld + (unsigned)(n0 + wn * WN + (fq >> 1) * 8)) * 2u;
    };
    auto row_off = [&](int m0, int64_t ld) -> unsigned {
        return ((unsigned)(m0 + wm * FM * p.OW + fr) * (unsigned)ld + (unsigned)ncol) * 2u;        // N % 128 == 0 (launcher): every wave's channels are inside the matrix
    };
    const unsigned c_jstep = (unsigned)p.OW * (unsigned)p.ldc * 2u, r_jstep = (unsigned)p.OW * (unsigned)p.ldr * 2u;

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // ONE register array, two tenants with disjoint lifetimes: the previous tile's results (packed f16) from its tile end until their store slots in
    // this tile's FIRST chunk; RES: this tile's residual values from their load slots in the LAST chunk (never the first: Cin >= 128) until the tile end
    u32x2 outp[FN][FM];
    unsigned ro_prev = 0, rr_cur = 0;
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) outp[i][j] = (u32x2){0u, 0u};
    // DEFER && GN: where the previous tile's sums go (slot index of its patch-row pair 0 for this wave, in float2 units) and the running pair's sums
    unsigned gslot_prev = 0;
    float gsum = 0.f, gsq = 0.f;
    const int gn_sh = p.gn_cg == 16 ? 4 : p.gn_cg == 8 ? 3 : 2;      // log2(channels per group)

    unsigned seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;     // STAMP: [0] DMA / store issue, [1] reads, [2] wait + barrier, [3] MFMA, [4] barrier, [5] prologue, [6] flush, [7] tile-end math
    unsigned kseg[5] = {0, 0, 0, 0, 0};                           // STAMP: whole K-steps by chunk kind (first = stores, middle, last with / without prefetch)
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

    const int nchunks = p.Cin / BK;                            // >= 2 (checked by the launcher)
    // SIDE (fie_conv3x3_plus_nhwc_f16: a resnet's conv2 + its 1x1 shortcut as one GEMM): behind the nine-tap chunks, one K-step per 64 channels of the
    // side inputs X2 | X3 (row m of [B*H*W, C] matrices = pixel m of NHWC images): the step reads the CENTRE tap of a halo loaded from the side image,
    // its weights sit behind the taps in the packed rows.  A tile then has 9 n + m K-steps, so the ring position of its first K-step (sb) moves by
    // m % 3 per tile and the stage indices become runtime values; a side step prefetches the WHOLE next halo (six slots) and drains (vmcnt(0)): the
    // m <= 8 side steps of a tile run at the load path's pace, the 9 n regular ones as before
    const int n2side = SIDE ? p.C2x / BK : 0, nside = SIDE ? (p.C2x + p.C3x) / BK : 0;
    const int steps = 9 * nchunks + nside;
    int sb = 0;
    auto rot3 = [](int x) { return x % 3; };                 // (a scalar: tap + ring position)
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, SIDE ? (int)p.a2_bytes * live : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A3, 0, SIDE ? (int)p.a3_bytes * live : 0, 0x00020000);
    auto w_soff = [&](int s) -> unsigned {                   // byte offset of tile-local K-step s in a packed weight row
        if (s < 9 * nchunks) { const int c = s / 9, t = s - 9 * c; return (unsigned)t * cin2 + (unsigned)c * (BK * 2); }
        return 9u * cin2 + (unsigned)(s - 9 * nchunks) * (BK * 2);
    };
    int tile = bid, m0 = 0, img = 0, y0 = 0, x0 = 0, img_rb = -1;
    geom(tile, m0, img, y0, x0);
    // the bias has landed before anything else is in flight (its registers are read at every tile end)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < FN; ++i) asm volatile("" : "+v"(bv[i]));

    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(p.gn_partial, 0, GN && p.gn_partial ? (p.M / (p.OH * p.OW)) * (p.gn_nch ? p.gn_nch : p.gn_rows >> 5) * p.gn_G * 8 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.res), 0, RES && p.res ? (int)(((int64_t)(p.M - 1) * p.ldr + p.N) * 2) : 0, 0x00020000);

    // ---- GNA: GroupNorm (+ SiLU) of the conv's INPUT applied on the resident halo (upstream resnet.py: norm -> nonlinearity -> conv; the separate
    // apply kernel -- one read and one write of the whole tensor -- disappears).  Every lane normalises, in place, the 16 bytes it DMA'd itself (its own
    // vmcnt says when they have landed), two taps after issuing them, 8 bytes behind each of two MFMA groups: y = silu(x * sc[c] + sh[c]) with the
    // per-(image, channel) coefficients of fie_groupnorm_coef_f16 (the same fp32 expression as gn_apply_kernel: same bits), from a table in LDS
    // behind the dump; pixels outside the image stay the zeros the range check loaded (the conv pads the NORMALISED tensor with zeros).
    auto gna_half = [&](auto slc, auto halfc, int buf, int chunk_c, int yy, int xx) {
        constexpr int SL = decltype(slc)::value, HALF = decltype(halfc)::value;
        unsigned t = hq[SL];
        asm volatile("" : "+v"(t));
        const int hy = (int)(t >> 8), hx = (int)(t & 255u);
        const int iy = yy + hy - 1, ix = xx + hx - 1;
        const bool okp = hy < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned chunk8 = (unsigned)((lane & 7) ^ (hx & 7));
        char* const dst = reinterpret_cast<char*>(halo_dst(SL, buf)) + lane * 16 + HALF * 8;
        const float4* tab = reinterpret_cast<const float4*>(lds + kLdsHalo + ((unsigned)chunk_c * 64u + chunk8 * 8u + HALF * 4u) * 8u);
        const u32x2 xv = *reinterpret_cast<const u32x2*>(dst);
        const float4 c0 = tab[0], c1 = tab[1];
        f16x4 xh;
        __builtin_memcpy(&xh, &xv, 8);
        float y0f = (float)xh[0] * c0.x + c0.y, y1f = (float)xh[1] * c0.z + c0.w, y2f = (float)xh[2] * c1.x + c1.y, y3f = (float)xh[3] * c1.z + c1.w;
        if (p.gna_silu) { y0f = fie_silu(y0f); y1f = fie_silu(y1f); y2f = fie_silu(y2f); y3f = fie_silu(y3f); }
        const f16x4 yh = {(half_t)y0f, (half_t)y1f, (half_t)y2f, (half_t)y3f};
        u32x2 yv;
        __builtin_memcpy(&yv, &yh, 8);
        if (!okp) yv = (u32x2){0u, 0u};
        *reinterpret_cast<u32x2*>(dst) = yv;
    };
    if constexpr (GNA) {                                       // the coefficient table of this block's image (launcher: one image per launch): Cin x (sc, sh) floats
        float* const tab = reinterpret_cast<float*>(lds + kLdsHalo);
        for (int i = tid; i < 2 * p.Cin; i += 512) tab[i] = p.gna_tab[i];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // before any LDS-DMA is in flight: the counted waits below count DMA pieces only
    }

    // ---- prologue: the first tile's first halo, the weights of its K-steps 0 and 1
#pragma unroll
    for (int i = 0; i < HSLOTS; ++i) bload16(rs_a, halo_dst(i, 0), halo_off(i, img, y0, x0), 0u);
    issue_w(0, 0u);
    issue_w(1, cin2);
    if constexpr (GNA) {                                       // the first halo is normalised here, in the open (every later one under MFMAs)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RWH) : "memory");
        __builtin_amdgcn_s_barrier();                          // the table: every thread's part written (lgkmcnt(0) above) before anybody reads it
        static_for([&](auto ic) {
            gna_half(ic, std::integral_constant<int, 0>{}, 0, 0, y0, x0);
            gna_half(ic, std::integral_constant<int, 1>{}, 0, 0, y0, x0);
        }, std::make_integer_sequence<int, HSLOTS>{});
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        wait_vm_barrier<RWH>();
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();
    stamp(5);

    f16x8 fw[2][FN], fa[2][FM];
    // NEXT = the tile whose first halo the last chunk prefetches (its geometry in imgn / y0n / x0n); other chunks prefetch this tile's chunk c + 1 from h_off
    int imgn = 0, y0n = 0, x0n = 0;
    bool have_next_ = false;
    auto kstep = [&](auto kindc, auto tapc, int c, unsigned hb, int hpf_buf) {
        constexpr int KIND = decltype(kindc)::value;
        constexpr int T = decltype(tapc)::value;
        constexpr int KY = T / 3, KX = T % 3;
        const int STAGE = SIDE ? rot3(T + sb) : T % STH, FILL = SIDE ? rot3(T + 2 + sb) : (T + 2) % STH;
        const unsigned tk0 = tprev;
        // ---- what this K-step issues besides its fragment reads: weights of K-step kt + 2 (two pieces), a halo piece, deferred stores, residual loads.
        // In the load segment (default), or -- IM -- from inside the MFMA segment, one instruction behind every fourth MFMA: a vector-memory
        // instruction costs ~60 cycles of issue among MFMAs against 100-200 in a load segment that also reads 16 fragments (MI355X_MICROARCH.md,
        // cycle constants), and the load segment is what the MFMA segment of the other wave group waits for
        auto issue_wp = [&](int i) {
            if constexpr (SIDE && KIND == K_LASTREG_S) {         // K-step kt + 2 may be a side step, the next tile's first step, or nothing: then a real load into the dump (the counts stay exact)
                int s2 = c * 9 + T + 2;
                bool ok = true;
                if (s2 >= steps) { s2 -= steps; ok = have_next_; }
                bload16(rs_w, reinterpret_cast<half_t*>(lds + (ok ? WRING_OFF + FILL * WSTAGE_B + (wave + 8 * i) * 1024 : DUMP_OFF)), w_base + (unsigned)i * w_step, ok ? w_soff(s2) : 0u);
            } else {
                constexpr int T2 = (T + 2) % 9;
                const int c2 = T + 2 > 8 ? ((KIND == K_LAST_F || KIND == K_LAST_N) ? 0 : c + 1) : c;
                bload16(rs_w, reinterpret_cast<half_t*>(lds + WRING_OFF + FILL * WSTAGE_B + (wave + 8 * i) * 1024), w_base + (unsigned)i * w_step,
                        (unsigned)T2 * cin2 + (unsigned)c2 * (BK * 2));
            }
        };
        auto issue_h = [&]() {
            if constexpr (SIDE && KIND == K_LASTREG_S) bload16(rs_x2, halo_dst(T, hpf_buf), halo_off_px(T, img, y0, x0, (unsigned)p.lda2 * 2u), 0u);       // the first side chunk's patch
            else if constexpr (KIND == K_LAST_F) bload16(rs_a, halo_dst(T, hpf_buf), halo_off(T, imgn, y0n, x0n), 0u);      // the next tile's first halo: offsets computed at use from the packed slot register
            else bload16(rs_a, halo_dst(T, hpf_buf), halo_off(T, img, y0, x0), (unsigned)(c + 1) * (BK * 2));        // this tile's next chunk
        };
        auto store1 = [&](int s) {
            if constexpr (ST16) {                                // store T = (fragment pair T / 4, channel fragment T % 4): 16 B per lane
                constexpr int JJ = T / FN, I = T % FN;
                u32x4 v = {outp[I][2 * JJ][0], outp[I][2 * JJ][1], outp[I][2 * JJ + 1][0], outp[I][2 * JJ + 1][1]};
                if constexpr (DEFER) {                           // the lane swap of the 16-byte form, here instead of at the tile end
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v[0]), "+v"(v[2]));
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v[1]), "+v"(v[3]));
                }
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_c, ro_prev + (unsigned)(2 * JJ) * c_jstep + (unsigned)(I * 32), 0, 0);
            } else {
                const int k = 2 * T + s, j = k / FN, i = k % FN;
                __builtin_amdgcn_raw_buffer_store_b64(outp[i][j], rs_c, ro_prev + (unsigned)j * c_jstep + (unsigned)(i * 32), 0, 0);
            }
        };
        // RES: the tile's residual values, all sixteen 8-byte loads per lane, go out in the MFMA segment of the tile's LAST K-step (two behind every
        // MFMA group): the registers they land in are free by then (the deferred outputs left them in the first chunk), their latency runs under the
        // MFMAs and the barrier, and no counted wait lies between their issue and their use at the tile end (where every older DMA has landed too)
        auto resload1 = [&](int k) {
            const int j = k / FN, i = k % FN;
            outp[i][j] = __builtin_amdgcn_raw_buffer_load_b64(rs_r, rr_cur + (unsigned)j * r_jstep + (unsigned)(i * 32), 0, 0);
        };
        // DEFER && GN: piece q of the sums of the previous tile's fragment pair (JJ, I) = (T / 4, T % 4), one piece behind every second MFMA.  The lane holds
        // 4 channels (one quad) of pixel fr in patch rows 2 JJ and 2 JJ + 1: v_dot2_f32_f16 sums the f16-rounded values and their squares in fp32, four
        // DPP adds total the 16 pixels of a row, v_permlane16/32_swap add the rows of the same group (8 / 16 channels per group); lane fr == 0 of the
        // group's first row stores the slot (one writer per slot, fixed order: deterministic)
        auto gn_piece = [&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            constexpr int JJ = (T & 7) / FN, I = (T & 7) % FN;
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            auto dot = [&](unsigned w) {
                const h2 x = __builtin_bit_cast(h2, w), one = {(_Float16)1.f, (_Float16)1.f};
                gsum = __builtin_amdgcn_fdot2(x, one, gsum, false);
                gsq = __builtin_amdgcn_fdot2(x, x, gsq, false);
            };
            auto dpp = [&](auto ctl) {
                constexpr int C = decltype(ctl)::value;
                gsum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gsum), C, 0xF, 0xF, true));
                gsq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gsq), C, 0xF, 0xF, true));
            };
            if constexpr (Q == 0) { gsum = 0.f; gsq = 0.f; dot(outp[I][2 * JJ][0]); }
            else if constexpr (Q == 1) dot(outp[I][2 * JJ][1]);
            else if constexpr (Q == 2) dot(outp[I][2 * JJ + 1][0]);
            else if constexpr (Q == 3) dot(outp[I][2 * JJ + 1][1]);
            else if constexpr (Q == 4) dpp(std::integral_constant<int, 0xB1>{});
            else if constexpr (Q == 5) dpp(std::integral_constant<int, 0x4E>{});
            else if constexpr (Q == 6) dpp(std::integral_constant<int, 0x141>{});
            else if constexpr (Q == 7) dpp(std::integral_constant<int, 0x140>{});
            else if constexpr (Q == 8 || Q == 9) {               // rows fq ^ 1 (8, 16 channels per group), then rows fq ^ 2 (16): a = b = v, swap, a + b
                float a0 = gsum, b0 = gsum, a1 = gsq, b1 = gsq;
                if constexpr (Q == 8) {
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(b0));
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a1), "+v"(b1));
                } else {
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(b0));
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a1), "+v"(b1));
                }
                const bool on = gn_sh >= (Q == 8 ? 3 : 4);
                gsum = on ? a0 + b0 : gsum;
                gsq = on ? a1 + b1 : gsq;
            } else if constexpr (Q == 10) {
                const int rows_per_group = 1 << (gn_sh - 2);
                const bool ok = fr == 0 && (fq & (rows_per_group - 1)) == 0;
                const unsigned slot = gslot_prev + (unsigned)(JJ * (p.OW >> 4) * p.gn_G) + (unsigned)(((wn * WN + I * 16) >> gn_sh) + (fq >> (gn_sh - 2)));
                u32x2 bits;
                const float2 v2 = make_float2(gsum, gsq);
                __builtin_memcpy(&bits, &v2, 8);
                __builtin_amdgcn_raw_buffer_store_b64(bits, rs_g, ok ? slot * 8u : kOob, 0, 0);
            }
        };
        constexpr bool GND = GN && DEFER && KIND == K_FIRST && T < 8;
        constexpr int NW_ = n2_w(KIND, T), NH_ = n2_h(KIND, T), NS_ = n2_st(VAR, KIND, T);
        constexpr bool RLT = RES && (KIND == K_LAST_F || KIND == K_LAST_N) && T == 8;
        // slot g (0..7) of the MFMA segment: the instruction issued behind MFMA group g (IM), or everything at once in the load segment
        auto slot = [&](auto gc) {
            constexpr int Gs = decltype(gc)::value;
            if constexpr (Gs < 2) { if constexpr (NW_ > 0) issue_wp(Gs); }
            else if constexpr (Gs == 2) { if constexpr (NH_ > 0) issue_h(); }
            else if constexpr (Gs < 5) { if constexpr (Gs - 3 < NS_) store1(Gs - 3); }
        };
        auto all_slots = [&](bool loads, bool stores_too) {
            if (loads) { slot(std::integral_constant<int, 0>{}); slot(std::integral_constant<int, 1>{}); slot(std::integral_constant<int, 2>{}); }
            if (stores_too) { slot(std::integral_constant<int, 3>{}); slot(std::integral_constant<int, 4>{}); }
        };
        // ---- load segment
        if constexpr (!IM) all_slots(true, !STM);
        stamp(0);
        const char* const hal = lds + HALO_OFF + hb;
        const char* const wst = lds + WRING_OFF + STAGE * WSTAGE_B;
        auto reads = [&](int kh) {
#pragma unroll
            for (int i = 0; i < FN; ++i) fw[kh][i] = *reinterpret_cast<const f16x8*>(wst + w_rd[kh] + i * 16 * 128);
#pragma unroll
            for (int j = 0; j < FM; ++j) fa[kh][j] = *reinterpret_cast<const f16x8*>(hal + a_rd[KX][kh] + (j + KY) * HP * 128);
        };
        reads(0);
        reads(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(1);
        wait_vm_barrier<n2_wait(VAR, RES, KIND, T)>();
        stamp(2);
        // ---- MFMA segment
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        auto group = [&](auto gc) {                              // MFMA group g = (k half g / 4, channel fragment g % 4): four row fragments
            constexpr int Gs = decltype(gc)::value, kh = Gs / FN, i = Gs % FN;
            if constexpr (GND) {                                 // one piece of the deferred GroupNorm sums behind every second MFMA
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[kh][i], fa[kh][0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[kh][i], fa[kh][1], acc[i][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0); gn_piece(std::integral_constant<int, 2 * Gs>{}); __builtin_amdgcn_sched_barrier(0);
                acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[kh][i], fa[kh][2], acc[i][2], 0, 0, 0);
                acc[i][3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[kh][i], fa[kh][3], acc[i][3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0); gn_piece(std::integral_constant<int, 2 * Gs + 1>{}); __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[kh][i], fa[kh][j], acc[i][j], 0, 0, 0);
            }
            if constexpr (RLT) { __builtin_amdgcn_sched_barrier(0); resload1(2 * Gs); resload1(2 * Gs + 1); __builtin_amdgcn_sched_barrier(0); }
            // GNA: half a halo piece (8 bytes of the 16 this lane DMA'd two taps ago) is normalised in place behind MFMA groups 2 and 5
            if constexpr (GNA && T >= 2 && T < 8 && KIND != K_LAST_N && KIND != K_LASTREG_S && (Gs == 2 || Gs == 5)) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (KIND == K_LAST_F) gna_half(std::integral_constant<int, T - 2>{}, std::integral_constant<int, Gs == 2 ? 0 : 1>{}, hpf_buf, 0, y0n, x0n);
                else gna_half(std::integral_constant<int, T - 2>{}, std::integral_constant<int, Gs == 2 ? 0 : 1>{}, hpf_buf, c + 1, y0, x0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (IM) { __builtin_amdgcn_sched_barrier(0); slot(gc); __builtin_amdgcn_sched_barrier(0); }
            else if constexpr (STM && (Gs == 3 || Gs == 4)) { __builtin_amdgcn_sched_barrier(0); slot(gc); __builtin_amdgcn_sched_barrier(0); }
        };
        group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{}); group(std::integral_constant<int, 2>{}); group(std::integral_constant<int, 3>{});
        group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{}); group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{});
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        stamp(3);
        __builtin_amdgcn_s_barrier();
        stamp(4);
        if constexpr (STAMP) kseg[KIND] += tprev - tk0;
    };
    auto chunk = [&](auto kindc, int c, unsigned hb, int hpf_buf) {
        kstep(kindc, std::integral_constant<int, 0>{}, c, hb, hpf_buf);
        kstep(kindc, std::integral_constant<int, 1>{}, c, hb, hpf_buf);
        kstep(kindc, std::integral_constant<int, 2>{}, c, hb, hpf_buf);
        kstep(kindc, std::integral_constant<int, 3>{}, c, hb, hpf_buf);
        kstep(kindc, std::integral_constant<int, 4>{}, c, hb, hpf_buf);
        kstep(kindc, std::integral_constant<int, 5>{}, c, hb, hpf_buf);
        kstep(kindc, std::integral_constant<int, 6>{}, c, hb, hpf_buf);
        kstep(kindc, std::integral_constant<int, 7>{}, c, hb, hpf_buf);
        kstep(kindc, std::integral_constant<int, 8>{}, c, hb, hpf_buf);
    };

    // ---- one side K-step (SIDE): chunk e of [X2 | X3], centre tap; prefetches the whole next halo (side chunk e + 1, or the next tile's first chunk) and
    // the weights of step + 2, then drains: every count behind it starts from zero
    auto side_step = [&](int e, unsigned hb, int hpf_buf) {
        const int s = 9 * nchunks + e;
        const int stage = (e + sb) % 3, fill = (e + 2 + sb) % 3;
        {
            int s2 = s + 2;
            bool ok = true;
            if (s2 >= steps) { s2 -= steps; ok = have_next_; }
            if (ok) {
                const unsigned so = w_soff(s2);
#pragma unroll
                for (int i = 0; i < RWH; ++i)
                    bload16(rs_w, reinterpret_cast<half_t*>(lds + WRING_OFF + fill * WSTAGE_B + (wave + 8 * i) * 1024), w_base + (unsigned)i * w_step, so);
            }
        }
        if (e + 1 < nside) {
            const bool x2 = e + 1 < n2side;
            const unsigned pix = (unsigned)(x2 ? p.lda2 : p.lda3) * 2u, so = (unsigned)(x2 ? e + 1 : e + 1 - n2side) * (BK * 2);
#pragma unroll
            for (int i = 0; i < HSLOTS; ++i) {
                if (x2) bload16(rs_x2, halo_dst(i, hpf_buf), halo_off_px(i, img, y0, x0, pix), so);
                else bload16(rs_x3, halo_dst(i, hpf_buf), halo_off_px(i, img, y0, x0, pix), so);
            }
        } else if (have_next_) {
#pragma unroll
            for (int i = 0; i < HSLOTS; ++i) bload16(rs_a, halo_dst(i, hpf_buf), halo_off(i, imgn, y0n, x0n), 0u);
        }
        stamp(0);
        const char* const hal = lds + HALO_OFF + hb;
        const char* const wst = lds + WRING_OFF + stage * WSTAGE_B;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int i = 0; i < FN; ++i) fw[kh][i] = *reinterpret_cast<const f16x8*>(wst + w_rd[kh] + i * 16 * 128);
#pragma unroll
            for (int j = 0; j < FM; ++j) fa[kh][j] = *reinterpret_cast<const f16x8*>(hal + a_rd[1][kh] + (j + 1) * HP * 128);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(1);
        wait_vm_barrier<0>();
        stamp(2);
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

    // ---- end of a tile: bias, row bias, activation, scale, residual on the accumulators (the order of gemm_common.h's epilogue); GroupNorm sums of
    // the f16-rounded results; pack to f16; clear the accumulators
    auto tile_end = [&](int m0c, int imgc) {
        auto as_h4 = [](u32x2 v) { f16x4 h; __builtin_memcpy(&h, &v, 8); return h; };
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            u32x2 bo = bv[i];
            asm volatile("" : "+v"(bo));                     // opaque: keeps the f16 -> f32 conversions of the bias from being hoisted into 16 loop-invariant registers
            const f16x4 b = as_h4(bo);
#pragma unroll
            for (int j = 0; j < FM; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)b[r];
            if constexpr (RB) {
                u32x2 ro_ = rbv[i];
                asm volatile("" : "+v"(ro_));
                const f16x4 rb = as_h4(ro_);
#pragma unroll
                for (int j = 0; j < FM; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)rb[r];
            }
        }
        // (no activation / scale here: the launcher sends such convs -- the ControlNet's tiny conditioning embedding -- to v1)
        if constexpr (RES) {
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) {
                    const f16x4 r4 = as_h4(outp[i][j]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)r4[r];
                }
        }
        if constexpr (GN && !DEFER) {
            // as gemm_common.h's epilogue (one writer per (granule, group) slot), with the slot stores as unconditional buffer stores
            const int ngrp = 16 / p.gn_cg;
            const int gn_nch = p.gn_nch ? p.gn_nch : p.gn_rows >> 5;
#pragma unroll
            for (int jj = 0; jj < FM / 2; ++jj) {
                const int mg = m0c + (wm * FM + 2 * jj) * p.OW;
                const int rem = mg - imgc * p.gn_rows;
                const int chunk = p.gn_chunk0 + ((rem / p.OW) >> 1) * (p.OW >> 4) + ((rem % p.OW) >> 4);
#pragma unroll
                for (int i = 0; i < FN; ++i) {
                    float sum = 0.f, sq = 0.f;
#pragma unroll
                    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float x = (float)(half_t)acc[i][2 * jj + dj][r];
                            sum += x;
                            sq += x * x;
                        }
                    const float rs = row16_sum(sum), rq = row16_sum(sq);
                    float ts[4], tq[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        ts[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs), 16 * k));
                        tq[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rq), 16 * k));
                    }
                    if (p.gn_cg == 16) { ts[0] = (ts[0] + ts[1]) + (ts[2] + ts[3]); tq[0] = (tq[0] + tq[1]) + (tq[2] + tq[3]); }
                    else if (p.gn_cg == 8) { ts[0] += ts[1]; tq[0] += tq[1]; ts[1] = ts[2] + ts[3]; tq[1] = tq[2] + tq[3]; }
                    const int nfrag = n0 + wn * WN + i * 16;
                    const float vs = lane == 0 ? ts[0] : lane == 1 ? ts[1] : lane == 2 ? ts[2] : ts[3];
                    const float vq = lane == 0 ? tq[0] : lane == 1 ? tq[1] : lane == 2 ? tq[2] : tq[3];
                    const bool ok = lane < ngrp && nfrag + lane * p.gn_cg < p.N;
                    u32x2 bits;
                    const float2 v2 = make_float2(vs, vq);
                    __builtin_memcpy(&bits, &v2, 8);
                    const unsigned off = ok ? (unsigned)(((imgc * gn_nch + chunk) * p.gn_G + nfrag / p.gn_cg + lane) * 8) : kOob;
                    __builtin_amdgcn_raw_buffer_store_b64(bits, rs_g, off, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (half_t)acc[i][j][r];
                __builtin_memcpy(&outp[i][j], &o, 8);
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        if constexpr (ST16 && !DEFER) {
            // fragments (i, 2 jj) and (i, 2 jj + 1): v_permlane16_swap exchanges the odd 16-lane rows of the first with the even rows of the second, dword by
            // dword.  Afterwards a lane of an even row (fq 0 / 2) holds 8 consecutive channels (fq / 2) * 8 .. + 7 of pixel row 2 jj, a lane of an odd row
            // the same of pixel row 2 jj + 1: one 16-byte store instead of two 8-byte ones
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int jj = 0; jj < FM / 2; ++jj)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        unsigned a = outp[i][2 * jj][d], b = outp[i][2 * jj + 1][d];
                        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
                        outp[i][2 * jj][d] = a;
                        outp[i][2 * jj + 1][d] = b;
                    }
        }
    };

    int g = 0;                                                  // chunks run so far: halo buffer g & 1 is being read, (g + 1) & 1 being filled
    bool first_tile = true;
    for (;;) {
        const int next = tile + G;
        const bool have_next = next < ntiles;
        have_next_ = have_next;
        int m0n = 0;
        if (have_next) geom(next, m0n, imgn, y0n, x0n);
        if constexpr (RB) {
            if (img != img_rb) {                                // the row bias of another image: an irregular load, so drain everything (always safe)
                const int nimg = p.M / (p.OH * p.OW);
                const __amdgpu_buffer_rsrc_t rs_rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.rowbias), 0, p.rowbias ? (int)(((int64_t)(nimg - 1) * p.ld_rowbias + p.N) * 2) : 0, 0x00020000);
#pragma unroll
                for (int i = 0; i < FN; ++i) rbv[i] = __builtin_amdgcn_raw_buffer_load_b64(rs_rb, (unsigned)(img * (int)p.ld_rowbias + ncol + i * 16) * 2u, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < FN; ++i) asm volatile("" : "+v"(rbv[i]));
                img_rb = img;
            }
        }
        if constexpr (RES) rr_cur = row_off(m0, p.ldr);
        // the block's first tile has nothing to store yet: its first chunk runs as a middle chunk.  (NOT as a first chunk with out-of-range dummy
        // stores: a store whose lanes are all dropped by the range check does not keep its place in the vmcnt order -- the counted waits behind such
        // dummies let DMA pieces through unretired, rarely and only on some variants: tools/halo_race.py, round 4.)
        if (first_tile) chunk(std::integral_constant<int, K_MID>{}, 0, (g & 1) ? (unsigned)HALO_B : 0u, (g + 1) & 1);
        else chunk(std::integral_constant<int, K_FIRST>{}, 0, (g & 1) ? (unsigned)HALO_B : 0u, (g + 1) & 1);
        first_tile = false;
        ++g;
        for (int c = 1; c + 1 < nchunks; ++c, ++g) chunk(std::integral_constant<int, K_MID>{}, c, (g & 1) ? (unsigned)HALO_B : 0u, (g + 1) & 1);
        if constexpr (SIDE) {
            chunk(std::integral_constant<int, K_LASTREG_S>{}, nchunks - 1, (g & 1) ? (unsigned)HALO_B : 0u, (g + 1) & 1);
            ++g;
            for (int e = 0; e < nside; ++e, ++g) side_step(e, (g & 1) ? (unsigned)HALO_B : 0u, (g + 1) & 1);
            sb = (sb + nside) % 3;                               // the 9 n regular steps leave the ring position alone
        } else {
            if (have_next) chunk(std::integral_constant<int, K_LAST_F>{}, nchunks - 1, (g & 1) ? (unsigned)HALO_B : 0u, (g + 1) & 1);
            else chunk(std::integral_constant<int, K_LAST_N>{}, nchunks - 1, (g & 1) ? (unsigned)HALO_B : 0u, (g + 1) & 1);
            ++g;
        }
        tile_end(m0, img);
        ro_prev = ST16 ? row_off16(m0, p.ldc) : row_off(m0, p.ldc);
        if constexpr (GN && DEFER) {
            // slot of this wave's patch-row pair 0: granule = (row pair of the image, 16-pixel column segment) as in the tile-end form; + n0's first group
            const int gn_nch = p.gn_nch ? p.gn_nch : p.gn_rows >> 5;
            const int chunk = p.gn_chunk0 + ((y0 + wm * FM) >> 1) * (p.OW >> 4) + (x0 >> 4);
            gslot_prev = (unsigned)((img * gn_nch + chunk) * p.gn_G + (n0 >> gn_sh));
        }
        stamp(7);
        if (!have_next) break;
        tile = next; m0 = m0n; img = imgn; y0 = y0n; x0 = x0n;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                // pairs with group 1's last barrier
    // ---- flush: the last tile's sixteen stores (DEFER: and its GroupNorm sums, and the lane swaps of the 16-byte form)
    if constexpr (ST16) {
        if constexpr (GN && DEFER) {
            static_for([&](auto tc) {
                constexpr int TT = decltype(tc)::value;
                constexpr int JJ = TT / FN, I = TT % FN;
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                float su = 0.f, sq = 0.f;
                const unsigned w4[4] = {outp[I][2 * JJ][0], outp[I][2 * JJ][1], outp[I][2 * JJ + 1][0], outp[I][2 * JJ + 1][1]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const h2 x = __builtin_bit_cast(h2, w4[k]), one = {(_Float16)1.f, (_Float16)1.f};
                    su = __builtin_amdgcn_fdot2(x, one, su, false);
                    sq = __builtin_amdgcn_fdot2(x, x, sq, false);
                }
                su = row16_sum(su);
                sq = row16_sum(sq);
                if (gn_sh >= 3) {
                    float a0 = su, b0 = su, a1 = sq, b1 = sq;
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(b0));
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a1), "+v"(b1));
                    su = a0 + b0; sq = a1 + b1;
                }
                if (gn_sh >= 4) {
                    float a0 = su, b0 = su, a1 = sq, b1 = sq;
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(b0));
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a1), "+v"(b1));
                    su = a0 + b0; sq = a1 + b1;
                }
                const int rows_per_group = 1 << (gn_sh - 2);
                const bool ok = fr == 0 && (fq & (rows_per_group - 1)) == 0;
                const unsigned slot = gslot_prev + (unsigned)(JJ * (p.OW >> 4) * p.gn_G) + (unsigned)(((wn * WN + I * 16) >> gn_sh) + (fq >> (gn_sh - 2)));
                u32x2 bits;
                const float2 v2 = make_float2(su, sq);
                __builtin_memcpy(&bits, &v2, 8);
                __builtin_amdgcn_raw_buffer_store_b64(bits, rs_g, ok ? slot * 8u : kOob, 0, 0);
            }, std::make_integer_sequence<int, (FM / 2) * FN>{});
        }
#pragma unroll
        for (int jj = 0; jj < FM / 2; ++jj)
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                u32x4 v = {outp[i][2 * jj][0], outp[i][2 * jj][1], outp[i][2 * jj + 1][0], outp[i][2 * jj + 1][1]};
                if constexpr (DEFER) {
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v[0]), "+v"(v[2]));
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v[1]), "+v"(v[3]));
                }
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_c, ro_prev + (unsigned)(2 * jj) * c_jstep + (unsigned)(i * 32), 0, 0);
            }
    } else {
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int i = 0; i < FN; ++i) __builtin_amdgcn_raw_buffer_store_b64(outp[i][j], rs_c, ro_prev + (unsigned)j * c_jstep + (unsigned)(i * 32), 0, 0);
    }
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(6);
        if (p.stamps && lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p.stamps[((size_t)bid * 8 + wave) * 12 + i] = seg[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) p.stamps[((size_t)bid * 8 + wave) * 12 + 8 + i] = kseg[i];
        }
    }
}

template <bool RES, bool GN, bool RB, int VAR, bool STAMP>
hipError_t halo2_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo2_kernel<RES, GN, RB, VAR, STAMP>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalo);
}

}  // namespace

int fie_conv_halo_init(void) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalo);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalo);
    if (e == hipSuccess) e = halo2_attr<false, false, false, 6, false>();
    if (e == hipSuccess) e = halo2_attr<false, true, false, 6, false>();
    if (e == hipSuccess) e = halo2_attr<true, false, false, 6, false>();
    if (e == hipSuccess) e = halo2_attr<true, true, false, 6, false>();
    if (e == hipSuccess) e = halo2_attr<false, false, true, 6, false>();
    if (e == hipSuccess) e = halo2_attr<false, true, true, 6, false>();
    if (e == hipSuccess) e = halo2_attr<false, false, false, 0, false>();
    if (e == hipSuccess) e = halo2_attr<false, true, false, 0, false>();
    if (e == hipSuccess) e = halo2_attr<true, false, false, 0, false>();
    if (e == hipSuccess) e = halo2_attr<true, true, false, 0, false>();
    if (e == hipSuccess) e = halo2_attr<false, false, true, 0, false>();
    if (e == hipSuccess) e = halo2_attr<false, true, true, 0, false>();
    if (e == hipSuccess) e = halo2_attr<false, false, false, 6, true>();
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo2_kernel<false, false, false, 6, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalo);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo2_kernel<false, true, false, 6, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalo);
    if (e == hipSuccess) e = halo2_attr<true, true, false, 6, true>();
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo2_kernel<false, true, false, 6, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalo + 8192);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo2_kernel<true, true, false, 6, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalo + 8192);
    if (e != hipSuccess) {
        fie_set_error("conv_halo: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return FIE_EHIP;
    }
    return FIE_OK;
}

// Shapes the halo-resident kernel takes: the plain same-size conv (stride 1, zero padding 1, no up-sampling gather, no 1x1 side inputs, fp16
// weights) on maps whose height and width are multiples of 16, Cin % 64 == 0, operands the 32-bit buffer offsets reach (checked by the caller: dma_ok)
bool fie_conv_halo_ok(const GemmArgs& a) {
    const bool base = a.stride == 1 && !a.ups && !a.taps2 && !a.oscat && a.pt == 1 && a.pl == 1 && a.H == a.OH && a.W == a.OW && a.OH % 16 == 0 &&
                      a.OW % 16 == 0 && a.Cin % BK == 0 && a.Cin >= BK && !a.w_scale && !a.out_f8 && a.splitk <= 1;
    if (!base) return false;
    if (!a.A2) return !a.A3 && a.K == 9 * a.Cin;
    // 1x1 side inputs (conv2 + shortcut): the persistent form only, so everything that form asks for
    return a.C2x % BK == 0 && a.C3x % BK == 0 && a.C2x > 0 && (a.A3 != nullptr) == (a.C3x > 0) && a.K == 9 * a.Cin + a.C2x + a.C3x && a.Cin >= 2 * BK && a.N % BNH == 0 &&
           !a.res && !a.rowbias && a.act == FIE_ACT_NONE && a.scale == 1.f && a.lda2 * 2 * (int64_t)a.OW * a.OH < (1ll << 31) && a.lda3 * 2 * (int64_t)a.OW * a.OH < (1ll << 31);
}

// GNA (the input's GroupNorm + SiLU applied on the resident halo): the persistent form, ONE image, GroupNorm sums armed for the output (every resnet conv
// of the VAE has them), no row bias, no side inputs, Cin <= 1024 (an 8 KiB coefficient table behind the dump), as many column tiles as fit the grid
bool fie_conv_halo_gna_ok(const GemmArgs& a) {
    if (!fie_conv_halo_ok(a) || a.A2 || a.rowbias) return false;
    const int nbn = (a.N + BNH - 1) / BNH;
    return a.M == a.OH * a.OW && a.Cin >= 2 * BK && a.Cin <= 1024 && a.N % BNH == 0 && nbn <= 64 && a.act == FIE_ACT_NONE && a.scale == 1.f && a.gn_partial != nullptr &&
           (a.gn_cg == 4 || a.gn_cg == 8 || a.gn_cg == 16);
}

// variant: 0 = v1 (one tile per block, code 71), 1 = v1 with stamps (73), 2 = v2: persistent blocks, deferred 16-byte stores issued inside the MFMA
// segments (72; with at most one tile per block it runs as v1: nothing to defer into), 4 = v2 with stamps (74), 5 = v2 with 8-byte stores issued in
// the load segments (76: the first form, kept for A/B).  Measured and not kept (profiles/r04_halo_conv.md): the stores counted in the waits
// (a race), every vector-memory instruction issued from inside the MFMA segment (no faster), the second k half's fragments read under the MFMAs
// (slower)
int fie_launch_conv_halo(fie_ctx* ctx, GemmArgs& a, int variant) {
    FIE_REQUIRE(fie_conv_halo_ok(a), "halo-resident conv: stride-1 same-size 3x3 conv with H, W %% 16 == 0 and Cin %% 64 == 0 only");
    a.frag_ld = a.OW;
    a.nbn = (a.N + BNH - 1) / BNH;
    a.nbm = (a.M / (a.OH * a.OW)) * (a.OH >> 4) * (a.OW >> 4);
    const int tiles = a.nbm * a.nbn;
    const bool side = a.A2 != nullptr;
    FIE_REQUIRE(!(side && variant != 2), "halo-resident conv: 1x1 side inputs run on tile code 72 only");
    if (a.gna_tab) {                                           // GroupNorm of the input applied on the resident halo: the persistent form with GroupNorm sums out, one image
        FIE_REQUIRE(fie_conv_halo_gna_ok(a), "halo-resident conv with the input's GroupNorm applied in LDS: shape or epilogue not built (fie_conv3x3_gn_ok)");
        int g = tiles < ctx->num_cus ? tiles : ctx->num_cus;
        g -= g % a.nbn;
        const int lds_bytes = kLdsHalo + a.Cin * 8;
        if (a.res) fie_launch(ctx, (conv_halo2_kernel<true, true, false, 6, false, false, true>), dim3((unsigned)g), dim3(512), lds_bytes, a);
        else fie_launch(ctx, (conv_halo2_kernel<false, true, false, 6, false, false, true>), dim3((unsigned)g), dim3(512), lds_bytes, a);
        FIE_LAUNCH_CHECK();
        return FIE_OK;
    }
    if (!side && variant >= 2 && tiles <= ctx->num_cus) variant = variant == 4 ? 1 : 0;      // one tile per block: the one-tile form (its epilogue is shorter than a tile end + a flush)
    // v2 stores during the first chunk and loads residuals in the last: two chunks at least; whole 128-channel column tiles (a wave whose channels lie
    // past N would issue stores / loads that the range check drops whole, and those do not keep the vmcnt order the counted waits rely on); row bias and
    // residual never come together (resnet conv1 / conv2); no activation / scale
    if (!side && variant >= 2 && (a.Cin < 2 * BK || a.N % BNH != 0 || (a.rowbias && a.res) || (variant == 4 && a.rowbias) || a.act != FIE_ACT_NONE || a.scale != 1.f)) variant = variant == 4 ? 1 : 0;
    if (variant < 2) {
        const dim3 grid((unsigned)tiles);
        if (variant == 1) fie_launch(ctx, (conv_halo_kernel<true>), grid, dim3(512), kLdsHalo, a);
        else fie_launch(ctx, (conv_halo_kernel<false>), grid, dim3(512), kLdsHalo, a);
        FIE_LAUNCH_CHECK();
        return FIE_OK;
    }
    // persistent grid: one block per CU (133 KB of LDS each), a multiple of the column tiles so that a block keeps its column tile
    int g = tiles < ctx->num_cus ? tiles : ctx->num_cus;
    g -= g % a.nbn;
    FIE_REQUIRE(g >= a.nbn && a.nbn <= ctx->num_cus, "halo-resident conv: more column tiles than CUs");
    const dim3 grid((unsigned)g);
    const bool res = a.res != nullptr, gn = a.gn_partial != nullptr, rb = a.rowbias != nullptr;
    FIE_REQUIRE(!gn || a.gn_cg == 4 || a.gn_cg == 8 || a.gn_cg == 16, "halo-resident conv: GroupNorm sums need 4, 8 or 16 channels per group");
#define FIE_HALO2(RES, GN, RB, VAR, ST) fie_launch(ctx, (conv_halo2_kernel<RES, GN, RB, VAR, ST>), grid, dim3(512), kLdsHalo, a)
#define FIE_HALO2_V(VAR)                                                                   \
    do {                                                                                   \
        if (rb && gn) FIE_HALO2(false, true, true, VAR, false);                            \
        else if (rb) FIE_HALO2(false, false, true, VAR, false);                            \
        else if (res && gn) FIE_HALO2(true, true, false, VAR, false);                      \
        else if (res) FIE_HALO2(true, false, false, VAR, false);                           \
        else if (gn) FIE_HALO2(false, true, false, VAR, false);                            \
        else FIE_HALO2(false, false, false, VAR, false);                                   \
    } while (0)
    if (side) { if (gn) fie_launch(ctx, (conv_halo2_kernel<false, true, false, 6, false, true>), grid, dim3(512), kLdsHalo, a); else fie_launch(ctx, (conv_halo2_kernel<false, false, false, 6, false, true>), grid, dim3(512), kLdsHalo, a); }
    else if (variant == 4) { if (res || gn) FIE_HALO2(true, true, false, 6, true); else FIE_HALO2(false, false, false, 6, true); }
    else if (variant == 5) FIE_HALO2_V(0);
    else FIE_HALO2_V(6);
#undef FIE_HALO2_V
#undef FIE_HALO2
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}
