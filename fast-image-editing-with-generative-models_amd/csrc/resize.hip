// K13: LANCZOS resize of a u8 RGB image on the device, bit-exact with Pillow's ImagingResample (8 bits per channel).
// Replaces `image.resize((1024, 1024), Image.LANCZOS)` at src/pipeline.py:251 of the reference (a Pillow call; the algorithm
// restated here is Pillow's src/libImaging/Resample.c: separable, horizontal pass first, 22-bit fixed-point coefficients,
// each pass rounded and clipped to u8).  The coefficient / bounds tables are Pillow's precompute_coeffs +
// normalize_coeffs_8bpc restated on the host (fie_amd/resize.py) and uploaded once per (in, out) size pair.
#include "fie_internal.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= kPrecisionBits;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// out[y][ox][c] = clip8(half + sum_x in[y][xmin + x][c] * k[ox][x])
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* in, int H, int W, int OW, const int* kk, const int* bounds, int ksize,
                                                       uint8_t* out) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (ox >= OW) return;
    const int xmin = bounds[2 * ox], n = bounds[2 * ox + 1];
    const int* k = kk + (size_t)ox * ksize;
    const uint8_t* row = in + ((size_t)y * W + xmin) * 3;
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        const int c = k[x];
        s0 += row[3 * x] * c;
        s1 += row[3 * x + 1] * c;
        s2 += row[3 * x + 2] * c;
    }
    uint8_t* o = out + ((size_t)y * OW + ox) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// out[oy][x][c] = clip8(half + sum_y in[ymin + y][x][c] * k[oy][y])
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* in, int W, int OH, const int* kk, const int* bounds, int ksize,
                                                       uint8_t* out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
    if (x >= W) return;
    const int ymin = bounds[2 * oy], n = bounds[2 * oy + 1];
    const int* k = kk + (size_t)oy * ksize;
    const uint8_t* col = in + ((size_t)ymin * W + x) * 3;
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
        const int c = k[y];
        const uint8_t* px = col + (size_t)y * W * 3;
        s0 += px[0] * c;
        s1 += px[1] * c;
        s2 += px[2] * c;
    }
    uint8_t* o = out + ((size_t)oy * W + x) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

}  // namespace

extern "C" int fie_resize_rgb_u8(fie_ctx* ctx, const uint8_t* src, int H, int W, uint8_t* dst, int OH, int OW, const int* kx,
                                 const int* bx, int ksx, const int* ky, const int* by, int ksy, uint8_t* tmp) {
    FIE_REQUIRE(ctx && src && dst && H > 0 && W > 0 && OH > 0 && OW > 0, "fie_resize_rgb_u8: bad argument");
    FIE_REQUIRE((W == OW || (kx && bx && ksx > 0)) && (H == OH || (ky && by && ksy > 0)), "fie_resize_rgb_u8: missing coefficient table");
    FIE_REQUIRE((W == OW || H == OH) || tmp, "fie_resize_rgb_u8: two passes need the [H, OW, 3] scratch image");
    const uint8_t* vin = src;
    if (W != OW) {                         // horizontal pass first, as Pillow does
        uint8_t* hout = H == OH ? dst : tmp;
        fie_launch(ctx, resize_h_kernel, dim3((OW + 255) / 256, H), dim3(256), 0, src, H, W, OW, kx, bx, ksx, hout);
        FIE_LAUNCH_CHECK();
        vin = hout;
    }
    if (H != OH) {
        fie_launch(ctx, resize_v_kernel, dim3((OW + 255) / 256, OH), dim3(256), 0, vin, OW, OH, ky, by, ksy, dst);
        FIE_LAUNCH_CHECK();
    }
    if (W == OW && H == OH) FIE_REQUIRE(hipMemcpyAsync(dst, src, (size_t)H * W * 3, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess,
                                        "fie_resize_rgb_u8: copy failed");
    return FIE_OK;
}
