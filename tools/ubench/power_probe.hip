// What does an instruction cost in ENERGY?  unet_kernel runs at the package power limit (rocm-smi: 1.34 kW, shader clock
// 2.06 GHz instead of 2.4), so its time is set by joules, not by issue slots.  One instruction class per run, back to back for
// SECONDS of wall time on every CU at the kernel's occupancy (two 4-wave workgroups per CU) with random operand bits, while the
// calling script samples `rocm-smi --showclocks --showpower`; prints the sustained rate.  Classes: mfma16 (v_mfma_f32_16x16x32_f16),
// mfma32 (v_mfma_f32_32x32x16_f16), lds (ds_read_b128, conflict free), valu (v_pk_fma_f32), l2 (global_load_dwordx4 of a 2 MB
// buffer: L2 hits), l1 (the same over 16 KB: L1 hits), idle (s_sleep).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/power_probe.hip -o power_probe;  run: power_probe <class> [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
// random fp16 pairs with exponents in [-8, 0): finite, no denormals
__device__ __forceinline__ unsigned rnd_h2(unsigned& s) { return (rnd(s) & 0x83ff83ffu) | 0x2c002c00u | ((rnd(s) & 0x0c000c00u)); }
__device__ __forceinline__ f16x8 rnd_frag(unsigned& s) {
  u32x4 v = {rnd_h2(s), rnd_h2(s), rnd_h2(s), rnd_h2(s)};
  return __builtin_bit_cast(f16x8, v);
}

__global__ __launch_bounds__(256, 2) void k_mfma16(float* out, int iters) {
  unsigned s = threadIdx.x * 977u + blockIdx.x * 131u + 7u;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd_frag(s); b[i] = rnd_frag(s); }
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 1) & 3], c[i], 0, 0, 0);
  }
  f32x4 t = c[0];
  for (int i = 1; i < 8; ++i) t += c[i];
  out[blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
}
__global__ __launch_bounds__(256, 2) void k_mfma32(float* out, int iters) {
  unsigned s = threadIdx.x * 977u + blockIdx.x * 131u + 7u;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd_frag(s); b[i] = rnd_frag(s); }
  f32x16 c[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i >> 1) & 3], c[i], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) t += c[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
__global__ __launch_bounds__(256, 2) void k_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[16384];          // 64 KB
  unsigned s = threadIdx.x * 977u + blockIdx.x * 131u + 7u;
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = rnd_h2(s);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = reinterpret_cast<const char*>(lds) + (lane >> 4) * 5376 + (lane & 15) * 16 + wave * 64;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const u32x4*>(base + ((it + k) & 7) * 320 + (k & 1) * 21504);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k];
  }
  out[blockIdx.x * 256 + threadIdx.x] = __builtin_bit_cast(float, acc.x ^ acc.y ^ acc.z ^ acc.w);
}
__global__ __launch_bounds__(256, 2) void k_valu(float* out, int iters) {
  unsigned s = threadIdx.x * 977u + blockIdx.x * 131u + 7u;
  f32x2 x[8], m, a;
  for (int i = 0; i < 8; ++i) x[i] = f32x2{(float)(rnd(s) & 1023) * 1e-3f, (float)(rnd(s) & 1023) * 1e-3f};
  m = f32x2{0.999f + (float)(rnd(s) & 7) * 1e-4f, 1.0001f};
  a = f32x2{1e-3f, -1e-3f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], m, a);
  }
  f32x2 t = x[0];
  for (int i = 1; i < 8; ++i) t += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = t[0] + t[1];
}
__global__ __launch_bounds__(256, 2) void k_l2(float* out, const u32x4* buf, int iters, unsigned mask) {
  // a wave reads 1 KB contiguous (global_load_dwordx4), 8 loads in flight, walking (mask + 1) KB of the buffer: 2 MB = L2 hits
  // (a CU's 8 waves touch far more than its 32 KB L1 between reuses), 16 KB = L1 hits
  const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  u32x4 acc = {0u, 0u, 0u, 0u};
  unsigned pos = wid * 8;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = buf[(((pos + k) & mask) << 6) + lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k];
    pos += 8 * 37;
  }
  out[blockIdx.x * 256 + threadIdx.x] = __builtin_bit_cast(float, acc.x ^ acc.y ^ acc.z ^ acc.w);
}
__global__ __launch_bounds__(256, 2) void k_idle(float* out, int iters) {
  for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(127);
  out[blockIdx.x * 256 + threadIdx.x] = 0.f;
}

int main(int argc, char** argv) {
  const char* cls = argc > 1 ? argv[1] : "mfma16";
  const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
  const int nb = 512;
  float* out;
  u32x4* buf;
  hipMalloc(&out, nb * 256 * 4);
  hipMalloc(&buf, 2048 * 64 * 16);
  std::vector<unsigned> h(2048 * 64 * 4);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s & 0x83ff83ffu) | 0x2c002c00u; }
  hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  int iters = 20000;
  double per_iter = 8;                                         // instructions per wave per iteration
  const char* unit = "";
  auto launch = [&]() {
    if (!strcmp(cls, "mfma16")) { hipLaunchKernelGGL(k_mfma16, dim3(nb), dim3(256), 0, 0, out, iters); unit = "v_mfma_f32_16x16x32_f16"; }
    else if (!strcmp(cls, "mfma32")) { hipLaunchKernelGGL(k_mfma32, dim3(nb), dim3(256), 0, 0, out, iters); per_iter = 4; unit = "v_mfma_f32_32x32x16_f16"; }
    else if (!strcmp(cls, "lds")) { hipLaunchKernelGGL(k_lds, dim3(nb), dim3(256), 0, 0, out, iters); unit = "ds_read_b128"; }
    else if (!strcmp(cls, "valu")) { hipLaunchKernelGGL(k_valu, dim3(nb), dim3(256), 0, 0, out, iters); unit = "v_pk_fma_f32"; }
    else if (!strcmp(cls, "l2")) { hipLaunchKernelGGL(k_l2, dim3(nb), dim3(256), 0, 0, out, buf, iters / 4, 2047u); unit = "global_load_dwordx4 (L2 hit)"; }
    else if (!strcmp(cls, "l1")) { hipLaunchKernelGGL(k_l2, dim3(nb), dim3(256), 0, 0, out, buf, iters / 4, 15u); unit = "global_load_dwordx4 (L1 hit)"; }
    else { hipLaunchKernelGGL(k_idle, dim3(nb), dim3(256), 0, 0, out, iters / 16); unit = "s_sleep"; }
  };
  launch();
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  do {
    for (int i = 0; i < 4; ++i) launch();
    hipDeviceSynchronize();
    launches += 4;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } while (el < seconds);
  const double it = !strcmp(cls, "l2") || !strcmp(cls, "l1") ? iters / 4 : !strcmp(cls, "mfma16") || !strcmp(cls, "mfma32") || !strcmp(cls, "lds") || !strcmp(cls, "valu") ? iters : iters / 16;
  const double wave_instr = (double)launches * nb * 4 * it * per_iter;
  printf("%s: %.3f s, %ld launches, %.4g wave-instructions/s of %s (%.2f per CU per ns)\n", cls, el, launches, wave_instr / el, unit,
         wave_instr / el / 256 / 1e9);
  return 0;
}
