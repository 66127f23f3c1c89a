// Is the fp32 MFMA rate power/clock limited?  Measures the shader clock (s_memtime cycles / wall_clock64 at 100 MHz)
// while every CU runs (a) a dense v_mfma_f32_32x32x2_f32 stream, (b) the same stream at a 50 % duty cycle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, long long* stamps,
                                         int iters, int sleep_every, int sleep_len) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float a0 = in[tid], a1 = in[tid + 1], b0 = in[tid + 2], b1 = in[tid + 3];
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const long long w0 = wall_clock64(), s0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c3, 0, 0, 0);
    a0 += 1e-9f; b1 -= 1e-9f;
    if (sleep_every && (i % sleep_every) == sleep_every - 1)
      for (int s = 0; s < sleep_len; ++s) __builtin_amdgcn_s_sleep(127);
  }
  const long long w1 = wall_clock64(), s1 = __builtin_readcyclecounter();
  float s = 0;
  for (int q = 0; q < 16; ++q) s += c0[q] + c1[q] + c2[q] + c3[q];
  out[tid] = s;
  if (tid == 0) { stamps[0] = w1 - w0; stamps[1] = s1 - s0; }
}

int main() {
  const int blocks = 512, iters = 20000;
  const size_t n = (size_t)blocks * 256 + 8;
  std::vector<float> h(n);
  for (auto& v : h) v = (float)rand() / (float)RAND_MAX - 0.5f;
  float *din, *dout; long long* st;
  (void)hipMalloc(&din, n * 4); (void)hipMalloc(&dout, n * 4); (void)hipMalloc(&st, 16);
  (void)hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  // sleep_every 32 iterations (128 MFMAs = 8192 cycles): s_sleep(127) = 127*64 = 8128 cycles -> ~50 % duty
  const int cfg[3][2] = {{0, 0}, {32, 1}, {32, 3}};
  for (auto& c : cfg)
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, din, dout, st, iters, c[0], c[1]);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      long long s[2]; (void)hipMemcpy(s, st, 16, hipMemcpyDeviceToHost);
      const double flops = (double)blocks * 4 * iters * 4 * 4096.0;
      printf("sleep_every=%d len=%d: %.3f ms, %.1f TFLOP/s avg, shader clock %.2f GHz (cycles %lld / wall %lld @100MHz)\n",
             c[0], c[1], ms, flops / ms / 1e9, (double)s[1] / ((double)s[0] * 10.0) , s[1], s[0]);
    }
  return 0;
}
