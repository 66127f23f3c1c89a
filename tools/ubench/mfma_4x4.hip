// v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4x4 outer products, K = 1): issue cost, and is a chain of four of them bitwise
// equal to one v_mfma_f32_16x16x4_f32 (i.e. is the 16x16x4 accumulation the k-ascending fmaf chain)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void timing(float* out, long long* cyc, int iters) {
  f32x4 m[16];
  for (int i = 0; i < 16; ++i) m[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, m[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// A [16 rows][4 k], B [4 k][16 cols] -> D 16x16 by one 16x16x4, and by 4x4x1_16B: block (mb, nb) = rows 4mb.., cols 4nb..
__global__ void compare(const float* A, const float* B, float* D1, float* D2) {
  const int l = threadIdx.x;
  // 16x16x4: A operand lane (i = l & 15, k = l >> 4); B operand lane (n = l & 15, k = l >> 4); D: lane = 16 * (row / 4) + col, reg = row % 4
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D1[(4 * (l >> 4) + r) * 16 + (l & 15)] = d[r];
  // 4x4x1_16B: block b = l >> 2; A lane holds a[row = 4 mb + (l & 3)], B lane holds b[col = 4 nb + (l & 3)]; D lane (b, j): col j, reg = row i
  const int blk = l >> 2, mb = blk >> 2, nb = blk & 3, j = l & 3;
  f32x4 e = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < 4; ++k) e = __builtin_amdgcn_mfma_f32_4x4x1f32(A[(4 * mb + j) * 4 + k], B[k * 16 + 4 * nb + j], e, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D2[(4 * mb + i) * 16 + 4 * nb + j] = e[i];
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 64 * 4); hipMalloc(&cyc, 256 * 8);
  hipLaunchKernelGGL(timing, dim3(256), dim3(64), 0, 0, out, cyc, 2000);
  hipLaunchKernelGGL(timing, dim3(256), dim3(64), 0, 0, out, cyc, 2000);
  hipDeviceSynchronize();
  std::vector<long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  printf("4x4x1_16B: %.2f cycles per MFMA (one wave per SIMD, 16 independent accumulators)\n", s / 256 / 2000 / 16);
  std::vector<float> A(64), B(64), d1(256), d2(256);
  unsigned x = 12345;
  auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xFFFF) / 65536.0f * 2.f - 1.f + ((x >> 3) & 0xFF) * 1e-7f; };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd();
  float *dA, *dB, *dD1, *dD2;
  hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD1, 1024); hipMalloc(&dD2, 1024);
  hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(compare, dim3(1), dim3(64), 0, 0, dA, dB, dD1, dD2);
  hipDeviceSynchronize();
  hipMemcpy(d1.data(), dD1, 1024, hipMemcpyDeviceToHost); hipMemcpy(d2.data(), dD2, 1024, hipMemcpyDeviceToHost);
  int diff = 0, wrong = 0;
  for (int i = 0; i < 256; ++i) {
    diff += memcmp(&d1[i], &d2[i], 4) != 0;
    float ref = 0.f; for (int k = 0; k < 4; ++k) ref = fmaf(A[(i / 16) * 4 + k], B[k * 16 + i % 16], ref);
    wrong += memcmp(&d1[i], &ref, 4) != 0;
  }
  printf("16x16x4 vs 4 x 4x4x1 chain: %d of 256 elements differ bitwise; 16x16x4 vs host k-ascending fmaf chain: %d differ\n", diff, wrong);
  return 0;
}
