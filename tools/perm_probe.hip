#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  unsigned x = threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
  unsigned y = threadIdx.x + 100;
  auto s = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
  auto t = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  o[256 + threadIdx.x] = t[0]; o[320 + threadIdx.x] = t[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 384 * 4); k<<<1, 64>>>(d); unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = {"p16(x,x)[0]", "p16(x,x)[1]", "p16(x,y)[0]", "p16(x,y)[1]", "p32(x,y)[0]", "p32(x,y)[1]"};
  for (int a = 0; a < 6; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 8) printf(" %u", h[a * 64 + i]); printf("\n"); }
  return 0;
}
