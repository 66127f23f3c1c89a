// Sustained fp32 MFMA rate on real (random) operands: 256 CUs x N waves, register-only v_mfma_f32_32x32x2_f32 chains.
// Calibrates the roofline denominator: the 157.3 TF spec assumes 2.4 GHz; the achievable number under DVFS is lower.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float a0 = in[tid], a1 = in[tid + 1], b0 = in[tid + 2], b1 = in[tid + 3];
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c3, 0, 0, 0);
    a0 += 1e-9f; b1 -= 1e-9f;   // keep operands changing
  }
  float s = 0;
  for (int k = 0; k < 16; ++k) s += c0[k] + c1[k] + c2[k] + c3[k];
  out[tid] = s;
}

int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 1;
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves = one per SIMD
  const size_t n = (size_t)blocks * 256 + 8;
  std::vector<float> h(n);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *din, *dout;
  hipMalloc(&din, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 2);
    printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s  (implied clock %.2f GHz at 64 FLOP/clk/SIMD x 1024 SIMDs)\n",
           waves_per_simd, ms, flops / ms / 1e9, flops / ms / 1e9 * 1e12 / (64.0 * 1024) / 1e9);
  }
  return 0;
}
