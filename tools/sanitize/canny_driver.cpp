// ASan / UBSan driver for the host Canny entry (csrc/canny.cpp): odd sizes, 1-pixel borders, flat and noisy images.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fie.h"
int main() {
    unsigned seed = 12345;
    const int sizes[][2] = {{1, 1}, {2, 3}, {3, 2}, {7, 5}, {8, 8}, {31, 33}, {64, 64}, {97, 131}, {256, 255}, {1024, 1024}};
    for (const auto& s : sizes) {
        const int h = s[0], w = s[1];
        std::vector<uint8_t> rgb((size_t)h * w * 3), out((size_t)h * w * 3, 7);
        for (auto& v : rgb) { seed = seed * 1664525u + 1013904223u; v = (uint8_t)(seed >> 24); }
        for (int flat = 0; flat < 2; ++flat) {
            if (flat) std::fill(rgb.begin(), rgb.end(), (uint8_t)128);
            const int rc = fie_canny_rgb_u8(rgb.data(), h, w, 100, 200, out.data());
            size_t edges = 0;
            for (size_t i = 0; i < out.size(); i += 3) edges += out[i] == 255;
            printf("%dx%d %s: rc %d, %zu edge pixels\n", h, w, flat ? "flat" : "noise", rc, edges);
        }
    }
    return 0;
}
