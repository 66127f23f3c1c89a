// LDS read bandwidth of one CU as the direct f16x2 convs use it: every lane of a wave reads 16 B (ds_read_b128; lanes of a
// 16-lane row contiguous, the four rows a multiple of 256 B apart -- rd_load_a's pattern, conflict free), 8 or 16 reads in
// flight per wave, for 1 / 2 / 4 / 8 waves per CU (one workgroup of 1 / 2 / 4 waves, or two of 4), alone and with MFMAs
// (v_mfma_f32_16x16x32_f16, 3 per read: ups.0's ratio is 1.5, downs.2's 3) issued by the same wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_read.hip -o lds_read.  Prints bytes / clock / CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int WIDTH, int MFMA_PER_READ>
__global__ __launch_bounds__(256) void lds_loop(unsigned* out, long long* cycles, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[16384];          // 64 KB
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane group g: 5376 B apart (RdGeo<128>::G), row n: 16 B apart; a wave's fragments of one step 320 B apart (a sample's rows)
  const char* base = reinterpret_cast<const char*>(lds) + (lane >> 4) * 5376 + (lane & 15) * 16 + wave * 64;
  u32x4 acc = {0u, 0u, 0u, 0u};
  f32x4 c[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const char* p = base + ((it + k) & 7) * 320 + (k & 1) * 21504;
      if (WIDTH == 16) v[k] = *reinterpret_cast<const u32x4*>(p);
      else {
        const uint2 a = *reinterpret_cast<const uint2*>(p - (lane & 15) * 8);
        v[k] = u32x4{a.x, a.y, 0u, 0u};
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MFMA_PER_READ == 0) acc ^= v[k];
#pragma unroll
      for (int m = 0; m < MFMA_PER_READ; ++m)
        c[(k + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, v[k]), __builtin_bit_cast(f16x8, v[(k + 1) & 7]), c[(k + m) & 3], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  if (MFMA_PER_READ) acc = __builtin_bit_cast(u32x4, c[0] + c[1] + c[2] + c[3]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int WIDTH, int MPR>
static void run(int waves_per_wg, int wgs_per_cu, unsigned* out, long long* cyc) {
  const int iters = 2000, n_cu = 256, nb = n_cu * wgs_per_cu;
  hipLaunchKernelGGL((lds_loop<WIDTH, MPR>), dim3(nb), dim3(64 * waves_per_wg), 0, 0, out, cyc, 10);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((lds_loop<WIDTH, MPR>), dim3(nb), dim3(64 * waves_per_wg), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(nb * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int b = 0; b < nb; ++b) for (int w = 0; w < waves_per_wg; ++w) mean += (double)h[b * 4 + w];
  mean /= (double)nb * waves_per_wg;
  // clock64 = s_memtime: shader clock; bytes per CU = waves x iters x 8 reads x 64 lanes x WIDTH
  const double bytes_cu = (double)waves_per_wg * wgs_per_cu * iters * 8 * 64 * WIDTH;
  printf("b%-3d mfma/read %d  waves/CU %d (%d WG x %d): %7.1f cycles per read per wave, %6.1f B/clk/CU (kernel %.3f ms -> %.1f B/ns/CU)\n",
         WIDTH * 8, MPR, waves_per_wg * wgs_per_cu, wgs_per_cu, waves_per_wg, mean / (iters * 8.0), bytes_cu / mean, ms, bytes_cu / (ms * 1e6));
}

int main() {
  unsigned* out; long long* cyc;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 4 * 8);
  const int cfg[4][2] = {{1, 1}, {2, 1}, {4, 1}, {4, 2}};
  for (auto& c : cfg) run<16, 0>(c[0], c[1], out, cyc);
  for (auto& c : cfg) run<8, 0>(c[0], c[1], out, cyc);
  for (auto& c : cfg) run<16, 1>(c[0], c[1], out, cyc);
  for (auto& c : cfg) run<16, 3>(c[0], c[1], out, cyc);
  return 0;
}
