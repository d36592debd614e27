// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds u16 value = its own index; lane l reads at byte address
// a(l) = (l & 15) * row_stride + (l >> 4) * 8  (pattern 0)  or  l * 8 (pattern 1); prints the 4 values each lane gets.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void probe(unsigned short* out, int pattern, int row_stride) {
  __shared__ unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int a = pattern == 0 ? (l & 15) * row_stride + (l >> 4) * 8 : (pattern == 1 ? l * 8 : ((l & 3) * 8 + (l >> 2) * row_stride));
  auto p = (__attribute__((address_space(3))) h4*)((__attribute__((address_space(3))) char*)lds + a);
  h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
  unsigned short r[4];
  __builtin_memcpy(r, &v, 8);
  for (int k = 0; k < 4; k++) out[l * 4 + k] = r[k];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  const int strides[3] = {64, 128, 32};
  for (int pattern = 0; pattern < 3; pattern++)
    for (int si = 0; si < (pattern == 1 ? 1 : 3); si++) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pattern, strides[si]);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("pattern %d row_stride %d bytes (values are u16 element indices = byte address / 2)\n", pattern, strides[si]);
      for (int l = 0; l < 64; l++) printf("  lane %2d: %5d %5d %5d %5d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : " |");
    }
  return 0;
}
