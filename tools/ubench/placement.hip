// Where do the 512 workgroups of a 2048-trajectory unet_kernel launch land?  Each workgroup (256 threads, 77 KB of LDS: two per
// CU) records HW_ID / XCC_ID of its first wave and its start time, then spins ~20 us so that all of them are resident together.
// Prints how many CUs hold two workgroups, the WAVE_ID (wave slot) parity of the pair, and their start-time skew.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, long long* t) {
  __shared__ float lds[77000 / 4];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID, all 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID[3:0]
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; t[blockIdx.x] = t0; }
  while (wall_clock64() - t0 < 2000) { __builtin_amdgcn_s_sleep(8); }              // 20 us at 100 MHz
  if (lds[threadIdx.x] < 0.f) out[0] = 0;
}
int main() {
  const int nb = 512;
  unsigned* out; long long* t;
  hipMalloc(&out, nb * 8); hipMalloc(&t, nb * 8);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, out, t); hipDeviceSynchronize(); }
  std::vector<unsigned> h(nb * 2); std::vector<long long> ht(nb);
  hipMemcpy(h.data(), out, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, nb * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < nb; ++b) {
    const unsigned hw = h[b * 2], key = ((h[b * 2 + 1] & 15) << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 15);
    cu[key].push_back(b);
  }
  int two = 0, other = 0, parity_differs = 0; double skew = 0, maxskew = 0;
  for (auto& kv : cu) {
    if (kv.second.size() == 2) {
      ++two;
      const int a = kv.second[0], b = kv.second[1];
      if ((h[a * 2] & 1) != (h[b * 2] & 1)) ++parity_differs;
      const double d = (double)llabs(ht[a] - ht[b]) * 0.01; skew += d; if (d > maxskew) maxskew = d;
    } else ++other;
  }
  printf("CUs seen %zu: %d with two workgroups (WAVE_ID bit 0 differs in %d), %d others; start skew of a pair: mean %.2f us max %.2f us\n",
         cu.size(), two, parity_differs, other, two ? skew / two : 0.0, maxskew);
  for (int b = 0; b < 16; ++b)
    printf("block %3d: xcc %u se %u sh %u cu %2u simd %u wave %2u  t %+.2f us | block %3d: xcc %u cu %2u wave %2u t %+.2f\n", b, h[b * 2 + 1] & 15, (h[b * 2] >> 13) & 7,
           (h[b * 2] >> 12) & 1, (h[b * 2] >> 8) & 15, (h[b * 2] >> 4) & 3, h[b * 2] & 15, (double)(ht[b] - ht[0]) * 0.01,
           b + 256, h[(b + 256) * 2 + 1] & 15, (h[(b + 256) * 2] >> 8) & 15, h[(b + 256) * 2] & 15, (double)(ht[b + 256] - ht[0]) * 0.01);
  return 0;
}
