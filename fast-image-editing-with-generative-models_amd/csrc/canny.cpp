// K11: RGB -> gray -> Canny(low, high, aperture 3, L1 gradient) on the host, integer exact (include/fie.h:
// fie_canny_rgb_u8).  Replaces cv2.cvtColor(COLOR_RGB2GRAY) + cv2.Canny + 3-channel stack at
// /root/reference/src/pipeline.py:200,205,208.  The reference also runs this step on the host CPU (OpenCV), before
// anything is handed to the device, so this is host logic on both sides of the boundary, not a device fallback.
//
// Algorithm (OpenCV 4.x imgproc semantics): 15-bit fixed-point luma; 3x3 Sobel with replicated
// borders; |dx|+|dy| magnitude with a zero ring; fixed-point tan(22.5 deg) sector test; asymmetric (>, >=)
// non-maximum comparisons; hysteresis as an explicit-stack flood fill from strong pixels.
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "fie_internal.h"

namespace {

constexpr int kShift = 15;
constexpr int kTg22 = (int)(0.4142135623730950488016887242097 * (1 << kShift) + 0.5);

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace

extern "C" int fie_canny_rgb_u8(const uint8_t* rgb, int H, int W, int low, int high, uint8_t* edges_rgb) {
    FIE_REQUIRE(rgb && edges_rgb && H > 0 && W > 0, "fie_canny_rgb_u8: bad argument");
    if (low > high) { int t = low; low = high; high = t; }
    const size_t n = (size_t)H * W;
    std::vector<uint8_t> gray(n);
    for (size_t i = 0; i < n; ++i)
        gray[i] = (uint8_t)((rgb[3 * i] * 9798 + rgb[3 * i + 1] * 19235 + rgb[3 * i + 2] * 3735 + (1 << 14)) >> 15);

    std::vector<int16_t> dx(n), dy(n);
    const int MS = W + 2;                                  // magnitude plane with a zero ring
    std::vector<int32_t> mag((size_t)(H + 2) * MS, 0);
    for (int y = 0; y < H; ++y) {
        const uint8_t* r0 = &gray[(size_t)clampi(y - 1, 0, H - 1) * W];
        const uint8_t* r1 = &gray[(size_t)y * W];
        const uint8_t* r2 = &gray[(size_t)clampi(y + 1, 0, H - 1) * W];
        for (int x = 0; x < W; ++x) {
            const int xl = clampi(x - 1, 0, W - 1), xr = clampi(x + 1, 0, W - 1);
            const int gx = (r0[xr] + 2 * r1[xr] + r2[xr]) - (r0[xl] + 2 * r1[xl] + r2[xl]);
            const int gy = (r2[xl] + 2 * r2[x] + r2[xr]) - (r0[xl] + 2 * r0[x] + r0[xr]);
            dx[(size_t)y * W + x] = (int16_t)gx;
            dy[(size_t)y * W + x] = (int16_t)gy;
            mag[(size_t)(y + 1) * MS + x + 1] = abs(gx) + abs(gy);
        }
    }

    // map: 0 = not an edge, 1 = candidate (weak), 2 = edge
    std::vector<uint8_t> map((size_t)(H + 2) * MS, 0);
    std::vector<int32_t> stack;
    stack.reserve(n / 8 + 16);
    for (int y = 0; y < H; ++y) {
        const int32_t* mp = &mag[(size_t)y * MS + 1];      // previous row
        const int32_t* mc = mp + MS;
        const int32_t* mn = mc + MS;
        uint8_t* mrow = &map[(size_t)(y + 1) * MS + 1];
        for (int x = 0; x < W; ++x) {
            const int m = mc[x];
            if (m <= low) continue;
            const int xs = dx[(size_t)y * W + x], ys = dy[(size_t)y * W + x];
            const int64_t ax = abs(xs), ay = (int64_t)abs(ys) << kShift;
            const int64_t tg22x = ax * kTg22;
            bool keep;
            if (ay < tg22x) {
                keep = m > mc[x - 1] && m >= mc[x + 1];
            } else {
                const int64_t tg67x = tg22x + (ax << (kShift + 1));
                if (ay > tg67x) {
                    keep = m > mp[x] && m >= mn[x];
                } else {
                    const int s = (xs ^ ys) < 0 ? -1 : 1;
                    keep = m > mp[x - s] && m > mn[x + s];
                }
            }
            if (!keep) continue;
            if (m > high) {
                mrow[x] = 2;
                stack.push_back((int32_t)((y + 1) * MS + x + 1));
            } else {
                mrow[x] = 1;
            }
        }
    }
    static const int dxy[8][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};
    while (!stack.empty()) {
        const int32_t pos = stack.back();
        stack.pop_back();
        for (int k = 0; k < 8; ++k) {
            const int32_t q = pos + dxy[k][0] * MS + dxy[k][1];
            if (map[q] == 1) {
                map[q] = 2;
                stack.push_back(q);
            }
        }
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const uint8_t e = map[(size_t)(y + 1) * MS + x + 1] == 2 ? 255 : 0;
            uint8_t* o = edges_rgb + ((size_t)y * W + x) * 3;
            o[0] = o[1] = o[2] = e;
        }
    return FIE_OK;
}
