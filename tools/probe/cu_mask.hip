// cu_mask.hip — which physical CUs does bit i of a hipExtStreamCreateWithCUMask mask enable on gfx950 (8 XCDs x 32 CUs)?
// Launches a spin kernel of one-wave workgroups on a masked stream and reports where they ran (HW_ID / XCC_ID).
// build: hipcc --offload-arch=gfx950 -O2 -o cu_mask.bin cu_mask.hip ; run: ./cu_mask.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <map>
__global__ __launch_bounds__(64) void where(unsigned* out, int spin) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(10);
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
  }
}
static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  const int nb = 256 * 24;
  unsigned* d; (void)hipMalloc(&d, nb * 8);
  hipLaunchKernelGGL(where, dim3(nb), dim3(64), 0, s, d, 2000);
  (void)hipStreamSynchronize(s);
  std::vector<unsigned> h(2 * nb); (void)hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
  std::map<int, std::set<int>> per_xcc;
  for (int i = 0; i < nb; i++) {
    const unsigned hw = h[2 * i], xc = h[2 * i + 1] & 15;
    const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xc].insert((se * 2 + sh) * 16 + cu);
  }
  int total = 0;
  printf("%s:", name);
  for (auto& kv : per_xcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  total CUs %d\n", total);
  if (total <= 64) for (auto& kv : per_xcc) { printf("   xcc%d (se*32+sh*16+cu):", kv.first); for (int c : kv.second) printf(" %d", c); printf("\n"); }
  (void)hipFree(d); (void)hipStreamDestroy(s);
}
int main() {
  std::vector<uint32_t> all(8, 0xFFFFFFFFu);
  run("all 256 bits", all);
  { auto m = all; m[0] = 0; run("bits 0..31 cleared", m); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0xFFFFFFFFu; run("only bits 0..31", m); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0xFFu; run("only bits 0..7", m); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0x01010101u; run("only bits 0,8,16,24", m); }
  { std::vector<uint32_t> m(8, 0); m[7] = 0xFFFFFFFFu; run("only bits 224..255", m); }
  { std::vector<uint32_t> m(8, 0xFFFFFFFFu); m[7] = 0; run("bits 224..255 cleared", m); }
  return 0;
}
