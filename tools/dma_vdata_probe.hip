// Does an LDS-DMA load (buffer_load_dwordx4 ... offen lds: no VGPR destination; the MUBUF VDATA field is encoded 0) ever write v0..v3?
// One asm statement owns v0..v3: known values in, N x (4 LDS-DMA pieces from cold memory), vmcnt(0), values out.  Many blocks, cold misses.
// build: hipcc --offload-arch=gfx950 -O2 tools/dma_vdata_probe.hip -o build/probe/dma_vdata_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void probe(const char* src, size_t bytes_per_block, unsigned* bad, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * bytes_per_block), 0, (int)bytes_per_block, 0x00020000);
    const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 4096);
    unsigned voff = lane * 16 + wave * 4096;
    unsigned o0, o1, o2, o3;
    unsigned nbad = 0;
    for (int it = 0; it < (int)(bytes_per_block / 16384); ++it) {
        unsigned m0a = lds0, m0b = lds0 + 1024, m0c = lds0 + 2048, m0d = lds0 + 3072;
        asm volatile(
            "v_mov_b32 v0, 0x11111111\n\tv_mov_b32 v1, 0x22222222\n\tv_mov_b32 v2, 0x33333333\n\tv_mov_b32 v3, 0x44444444\n\t"
            "s_mov_b32 m0, %6\n\ts_nop 4\n\tbuffer_load_dwordx4 %4, %5, 0 offen lds\n\t"
            "s_mov_b32 m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, 0 offen offset:1024 lds\n\t"
            "s_mov_b32 m0, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, 0 offen offset:2048 lds\n\t"
            "s_mov_b32 m0, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, 0 offen offset:3072 lds\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\tv_mov_b32 %2, v2\n\tv_mov_b32 %3, v3\n\t"
            : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
            : "v"(voff), "s"(rs), "s"(m0a), "s"(m0b), "s"(m0c), "s"(m0d)
            : "v0", "v1", "v2", "v3", "memory");
        nbad += (o0 != 0x11111111u) + (o1 != 0x22222222u) + (o2 != 0x33333333u) + (o3 != 0x44444444u);
        voff += 16384;
        __syncthreads();
        sink[(blockIdx.x * 256 + threadIdx.x) & 1023] = *(unsigned*)(smem + threadIdx.x * 4);
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    const size_t per_block = 1 << 20, blocks = 2048;
    char* src; unsigned *bad, *sink;
    hipMalloc(&src, per_block * blocks); hipMemset(src, 0x5a, per_block * blocks);
    hipMalloc(&bad, 4); hipMemset(bad, 0, 4); hipMalloc(&sink, 4096);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 16384, 0, src, per_block, bad, sink);
    hipDeviceSynchronize();
    unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("LDS-DMA loads with v0..v3 holding known values: %u clobbered register reads in %zu blocks x 64 rounds x 3 launches (%s)\n", h, blocks, hipGetErrorString(hipGetLastError()));
    return 0;
}
