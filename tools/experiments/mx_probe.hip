// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit scales (E8M0 127): which k does byte b of lane l hold?
// Hypothesis H1: lane l holds row (l & 15), k = 32 * (l >> 4) + b, b = 0..31 (register r = bytes 4r..4r+3).
// Hypothesis H2: lane l holds k = 16 * (l >> 4) + b for b < 16 and 64 + 16 * (l >> 4) + (b - 16) for b >= 16.
// Build: hipcc --offload-arch=gfx950 -O2 tools/experiments/mx_probe.hip -o gpurun_out/mx_probe   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const unsigned char* a, const unsigned char* b, float* c, int scale) {
    v8i x, y;
    memcpy(&x, a + threadIdx.x * 32, 32);
    memcpy(&y, b + threadIdx.x * 32, 32);
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, y, acc, 0, 0, 0, scale, 0, scale);
    for (int r = 0; r < 4; ++r) c[threadIdx.x * 4 + r] = acc[r];
}

static unsigned char e4m3(int v) {      // small non-negative integers 0..8 exactly
    static const unsigned char t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
    return t[v];
}

int main() {
    int A[16][128], B[16][128];
    srand(1);
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 128; ++kk) { A[i][kk] = rand() % 9; B[i][kk] = rand() % 9; }
    float ref[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 128; ++kk) s += A[i][kk] * B[j][kk]; ref[i][j] = s; }
    for (int hyp = 1; hyp <= 2; ++hyp) {
        unsigned char ha[64 * 32], hb[64 * 32];
        for (int l = 0; l < 64; ++l) for (int bt = 0; bt < 32; ++bt) {
            const int kb = l >> 4, kk = hyp == 1 ? 32 * kb + bt : (bt < 16 ? 16 * kb + bt : 64 + 16 * kb + bt - 16);
            ha[l * 32 + bt] = e4m3(A[l & 15][kk]);
            hb[l * 32 + bt] = e4m3(B[l & 15][kk]);
        }
        unsigned char *da, *db; float* dc; float hc[256];
        hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, sizeof(hc));
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        for (int scale : {0x7F7F7F7F, 0x7F}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, scale);
            hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
            // C/D layout of the 16x16 forms: col = lane & 15, row = (lane >> 4) * 4 + reg; with A as the first operand rows index A
            int bad = 0, badT = 0;
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
                const int row = (l >> 4) * 4 + r, col = l & 15;
                if (hc[l * 4 + r] != ref[row][col]) ++bad;
                if (hc[l * 4 + r] != ref[col][row]) ++badT;
            }
            printf("hypothesis %d scale %#x: C[row=A][col=B] mismatches %d, transposed mismatches %d (sample %g vs %g)\n", hyp, scale, bad, badT, hc[0], ref[0][0]);
        }
    }
    return 0;
}
