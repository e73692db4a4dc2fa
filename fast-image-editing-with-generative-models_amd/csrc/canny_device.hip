// K11 on the device: RGB -> gray -> Canny(low, high, aperture 3, L1 gradient), integer exact, same results as the host entry
// fie_canny_rgb_u8 (canny.cpp) and the numpy oracle.  (include/fie.h: fie_canny_rgb_device_u8)
//
//   canny_nms_kernel    one 32x8 output tile per block; the 36x12 gray halo (replicated image border) and the 34x10 Sobel /
//                       L1-magnitude halo (zero outside the image) live in LDS; writes map = 0 (none) / 1 (weak) / 2 (strong)
//   canny_hyst_kernel   hysteresis as a fixed point: a 32x32 tile + 1-pixel halo in LDS is relaxed until nothing changes
//                       (weak next to strong -> strong), written back, and a device flag records whether any tile changed;
//                       the host re-launches until the flag stays clear (8-connected closure is order independent, so the
//                       result equals the serial flood fill)
//   canny_out_kernel    map == 2 -> 255 on three channels
// The hysteresis loop reads the flag back, so this entry SYNCHRONISES the stream; it belongs to the host-side preparation
// of an edit (`FastEditor.preprocess_image`), not to the captured denoising graph.
#include "fie_internal.h"

namespace {

constexpr int kShift = 15;
constexpr int kTg22 = (int)(0.4142135623730950488016887242097 * (1 << kShift) + 0.5);
constexpr int TX = 32, TY = 8;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void canny_nms_kernel(const uint8_t* rgb, int H, int W, int low, int high, uint8_t* map) {
    __shared__ uint8_t gray[TY + 4][TX + 4];
    __shared__ short sdx[TY + 2][TX + 2], sdy[TY + 2][TX + 2];
    __shared__ int smag[TY + 2][TX + 2];
    const int tid = threadIdx.y * TX + threadIdx.x;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    for (int i = tid; i < (TY + 4) * (TX + 4); i += TX * TY) {
        const int ly = i / (TX + 4), lx = i - ly * (TX + 4);
        const int gx = clampi(x0 + lx - 2, 0, W - 1), gy = clampi(y0 + ly - 2, 0, H - 1);       // BORDER_REPLICATE
        const uint8_t* p = rgb + ((size_t)gy * W + gx) * 3;
        gray[ly][lx] = (uint8_t)((p[0] * 9798 + p[1] * 19235 + p[2] * 3735 + (1 << 14)) >> 15);
    }
    __syncthreads();
    for (int i = tid; i < (TY + 2) * (TX + 2); i += TX * TY) {
        const int ly = i / (TX + 2), lx = i - ly * (TX + 2);
        const int gx = x0 + lx - 1, gy = y0 + ly - 1;
        int dx = 0, dy = 0, mag = 0;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
            // gray[ly + 1][lx + 1] is this pixel; its replicated neighbours are in the halo -- except that replication is
            // relative to the IMAGE border, which the clamped halo load already encodes
            const int a = gray[ly][lx], b = gray[ly][lx + 1], c = gray[ly][lx + 2];
            const int d = gray[ly + 1][lx], f = gray[ly + 1][lx + 2];
            const int g = gray[ly + 2][lx], h = gray[ly + 2][lx + 1], k = gray[ly + 2][lx + 2];
            dx = (c + 2 * f + k) - (a + 2 * d + g);
            dy = (g + 2 * h + k) - (a + 2 * b + c);
            mag = abs(dx) + abs(dy);
        }
        sdx[ly][lx] = (short)dx;
        sdy[ly][lx] = (short)dy;
        smag[ly][lx] = mag;                                  // 0 outside the image
    }
    __syncthreads();
    const int gx = x0 + threadIdx.x, gy = y0 + threadIdx.y;
    if (gx >= W || gy >= H) return;
    const int lx = threadIdx.x + 1, ly = threadIdx.y + 1;
    const int m = smag[ly][lx];
    uint8_t out = 0;
    if (m > low) {
        const int xs = sdx[ly][lx], ys = sdy[ly][lx];
        const long long ax = abs(xs), ay = (long long)abs(ys) << kShift;
        const long long tg22x = ax * kTg22;
        bool keep;
        if (ay < tg22x) {
            keep = m > smag[ly][lx - 1] && m >= smag[ly][lx + 1];
        } else {
            const long long tg67x = tg22x + (ax << (kShift + 1));
            if (ay > tg67x) {
                keep = m > smag[ly - 1][lx] && m >= smag[ly + 1][lx];
            } else {
                const int s = (xs ^ ys) < 0 ? -1 : 1;
                keep = m > smag[ly - 1][lx - s] && m > smag[ly + 1][lx + s];
            }
        }
        if (keep) out = m > high ? 2 : 1;
    }
    map[(size_t)gy * W + gx] = out;
}

constexpr int HT = 32;

__global__ __launch_bounds__(256) void canny_hyst_kernel(uint8_t* map, int H, int W, int* changed) {
    __shared__ uint8_t t[HT + 2][HT + 2 + 2];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * HT, y0 = blockIdx.y * HT;
    for (int i = tid; i < (HT + 2) * (HT + 2); i += 256) {
        const int ly = i / (HT + 2), lx = i - ly * (HT + 2);
        const int gx = x0 + lx - 1, gy = y0 + ly - 1;
        t[ly][lx] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? map[(size_t)gy * W + gx] : 0;
    }
    __syncthreads();
    bool any_local = false;
    for (;;) {
        bool ch = false;
        for (int i = tid; i < HT * HT; i += 256) {
            const int ly = (i >> 5) + 1, lx = (i & 31) + 1;
            if (t[ly][lx] == 1) {
                const bool strong = t[ly - 1][lx - 1] == 2 || t[ly - 1][lx] == 2 || t[ly - 1][lx + 1] == 2 || t[ly][lx - 1] == 2 ||
                                    t[ly][lx + 1] == 2 || t[ly + 1][lx - 1] == 2 || t[ly + 1][lx] == 2 || t[ly + 1][lx + 1] == 2;
                if (strong) {
                    t[ly][lx] = 2;          // monotone 1 -> 2: racing readers see either value, both are valid states
                    ch = true;
                }
            }
        }
        any_local |= ch;
        if (!__syncthreads_or(ch)) break;
    }
    if (any_local) {
        for (int i = tid; i < HT * HT; i += 256) {
            const int ly = (i >> 5) + 1, lx = (i & 31) + 1;
            const int gx = x0 + lx - 1, gy = y0 + ly - 1;
            if (gx < W && gy < H && t[ly][lx] == 2) map[(size_t)gy * W + gx] = 2;
        }
    }
    if (__syncthreads_or(any_local) && tid == 0) atomicOr(changed, 1);
}

__global__ void canny_out_kernel(const uint8_t* map, size_t n, uint8_t* out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint8_t e = map[i] == 2 ? 255 : 0;
        out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = e;
    }
}

}  // namespace

extern "C" {

int64_t fie_canny_workspace_bytes(int H, int W) { return (int64_t)H * W + 256; }     // the label map + the flag words of a round

// One round of hysteresis = kPasses passes, each with its own flag word: the fixed point is reached when the LAST pass of a round changed nothing.
// (Round 3 read ONE flag per round -- "any of the four passes changed something" -- so an image that needs two passes paid for eight and two read-backs.)
constexpr int kPasses = 4;

static int canny_round(fie_ctx* ctx, uint8_t* map, int H, int W, int* flags) {
    if (hipMemsetAsync(flags, 0, kPasses * sizeof(int), ctx->stream) != hipSuccess) { fie_set_error("fie_canny_rgb_device_u8: memset failed"); return FIE_EHIP; }
    const dim3 hgrid((W + HT - 1) / HT, (H + HT - 1) / HT);
    for (int rep = 0; rep < kPasses; ++rep) fie_launch(ctx, canny_hyst_kernel, hgrid, dim3(256), 0, map, H, W, flags + rep);
    FIE_LAUNCH_CHECK();
    return FIE_OK;
}

static void canny_out(fie_ctx* ctx, const uint8_t* map, int H, int W, uint8_t* edges_rgb) {
    const size_t n = (size_t)H * W;
    fie_launch(ctx, canny_out_kernel, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, map, n, edges_rgb);
}

// Asynchronous first half: NMS, `rounds` rounds of hysteresis, the edge map of that state, and the LAST round's flags on their way to `host_flags`
// (kPasses ints of PINNED host memory).  Nothing is waited for: the host goes on with its own work while the device does this.
int fie_canny_rgb_device_begin_u8(fie_ctx* ctx, const uint8_t* rgb, int H, int W, int low, int high, int rounds, void* workspace, uint8_t* edges_rgb,
                                  int* host_flags) {
    FIE_REQUIRE(ctx && rgb && workspace && edges_rgb && host_flags && H > 0 && W > 0 && rounds >= 1 && rounds <= 64, "fie_canny_rgb_device_begin_u8: bad argument");
    if (low > high) { int t = low; low = high; high = t; }
    uint8_t* map = (uint8_t*)workspace;
    int* flags = (int*)(map + (((size_t)H * W + 63) / 64) * 64);
    fie_launch(ctx, canny_nms_kernel, dim3((W + TX - 1) / TX, (H + TY - 1) / TY), dim3(TX, TY), 0, rgb, H, W, low, high, map);
    FIE_LAUNCH_CHECK();
    for (int r = 0; r < rounds; ++r) {
        const int rc = canny_round(ctx, map, H, W, flags);
        if (rc != FIE_OK) return rc;
    }
    canny_out(ctx, map, H, W, edges_rgb);
    FIE_LAUNCH_CHECK();
    if (hipMemcpyAsync(host_flags, flags, kPasses * sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
        fie_set_error("fie_canny_rgb_device_begin_u8: flag copy failed");
        return FIE_EHIP;
    }
    return FIE_OK;
}

// Second half: waits for the first (SYNCHRONISES the ctx stream); when its last pass still changed something, further rounds until one ends unchanged, and
// the edge map again.  iterations (optional): hysteresis passes launched HERE (0: begin's rounds had reached the fixed point, edges_rgb was final).
int fie_canny_rgb_device_finish_u8(fie_ctx* ctx, int H, int W, void* workspace, uint8_t* edges_rgb, int* host_flags, int* iterations) {
    FIE_REQUIRE(ctx && workspace && edges_rgb && host_flags && H > 0 && W > 0, "fie_canny_rgb_device_finish_u8: bad argument");
    uint8_t* map = (uint8_t*)workspace;
    int* flags = (int*)(map + (((size_t)H * W + 63) / 64) * 64);
    const int max_iters = ((W + HT - 1) / HT) * ((H + HT - 1) / HT) + 2 * kPasses;       // a strong seed can cross every tile at most once
    int iters = 0;
    bool more = false;
    for (;;) {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { fie_set_error("fie_canny_rgb_device_finish_u8: synchronisation failed"); return FIE_EHIP; }
        if (!host_flags[kPasses - 1]) break;
        if (iters > max_iters) { fie_set_error("fie_canny_rgb_device_u8: hysteresis did not converge in %d passes", iters); return FIE_EHIP; }
        more = true;
        const int rc = canny_round(ctx, map, H, W, flags);
        if (rc != FIE_OK) return rc;
        if (hipMemcpyAsync(host_flags, flags, kPasses * sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
            fie_set_error("fie_canny_rgb_device_finish_u8: flag copy failed");
            return FIE_EHIP;
        }
        iters += kPasses;
    }
    if (more) {
        canny_out(ctx, map, H, W, edges_rgb);
        FIE_LAUNCH_CHECK();
    }
    if (iterations) *iterations = iters;
    return FIE_OK;
}

int fie_canny_rgb_device_u8(fie_ctx* ctx, const uint8_t* rgb, int H, int W, int low, int high, void* workspace,
                            uint8_t* edges_rgb, int* iterations) {
    FIE_REQUIRE(ctx && rgb && workspace && edges_rgb && H > 0 && W > 0, "fie_canny_rgb_device_u8: bad argument");
    int host_flags[kPasses] = {0, 0, 0, 0};               // pageable: the copy is then synchronous with respect to the host, which this entry is anyway
    int rc = fie_canny_rgb_device_begin_u8(ctx, rgb, H, W, low, high, 1, workspace, edges_rgb, host_flags);
    if (rc != FIE_OK) return rc;
    int more = 0;
    rc = fie_canny_rgb_device_finish_u8(ctx, H, W, workspace, edges_rgb, host_flags, &more);
    if (iterations) *iterations = kPasses + more;
    return rc;
}

}  // extern "C"
