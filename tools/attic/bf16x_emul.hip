// fp32 GEMM emulated on the bf16 matrix pipe: a = a1 + a2 + a3 (three bf16 pieces, exact), same for b; a*b ~ sum of 6 (or all 9)
// piece products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Measures (1) the error of the x6 / x9 schemes and of
// the native fp32 MFMA chain against an fp64 host reference, (2) cycles per bf16 MFMA, (3) whether independent VALU of the
// SAME wave hides in the bf16 MFMA's shadow (it does not for the fp32 MFMA: mfma_valu_overlap.hip).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// truncation split: p1 = top 16 bits of a, p2 = top 16 bits of (a - p1), p3 = a - p1 - p2 (<= 8 significant bits: exact)
__device__ __host__ inline void split3(float a, unsigned short p[3]) {
  unsigned u; memcpy(&u, &a, 4);
  unsigned u1 = u & 0xFFFF0000u; float f1; memcpy(&f1, &u1, 4);
  float r1 = a - f1; unsigned v; memcpy(&v, &r1, 4);
  unsigned u2 = v & 0xFFFF0000u; float f2; memcpy(&f2, &u2, 4);
  float r2 = r1 - f2; unsigned w; memcpy(&w, &r2, 4);
  p[0] = u1 >> 16; p[1] = u2 >> 16; p[2] = w >> 16;
}

// A [16][K] row-major, B [K][16] row-major (B[k][n]); D [16][16].  mode 0: fp32 16x16x4 chain; 6 / 9: bf16 piece products,
// low-order products first inside every K=32 chunk; 16: x6 with the five low-order products in a second accumulator
__global__ void gemm(const float* A, const float* B, float* D, int K, int mode) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, lo = {0.f, 0.f, 0.f, 0.f};
  if (mode == 0) {
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k + g], B[(k + g) * 16 + i], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 32) {
      s16x8 a[3], b[3];
      for (int j = 0; j < 8; ++j) {
        unsigned short p[3];
        split3(A[i * K + k0 + g * 8 + j], p);
        for (int q = 0; q < 3; ++q) a[q][j] = (short)p[q];
        split3(B[(k0 + g * 8 + j) * 16 + i], p);
        for (int q = 0; q < 3; ++q) b[q][j] = (short)p[q];
      }
#define MM(x, y, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[x]), __builtin_bit_cast(bf16x8, b[y]), c, 0, 0, 0)
      if (mode == 9) { MM(2, 2, acc); MM(2, 1, acc); MM(1, 2, acc); }
      if (mode == 16) { MM(2, 0, lo); MM(1, 1, lo); MM(0, 2, lo); MM(1, 0, lo); MM(0, 1, lo); MM(0, 0, acc); }
      else { MM(2, 0, acc); MM(1, 1, acc); MM(0, 2, acc); MM(1, 0, acc); MM(0, 1, acc); MM(0, 0, acc); }
    }
    if (mode == 16) acc += lo;
  }
  for (int r = 0; r < 4; ++r) D[(g * 4 + r) * 16 + i] = acc[r];
}

template <int NV>
__global__ __launch_bounds__(256) void timing(int iters, float* out, long long* cyc) {
  f32x4 m[16];
  float v[16];
  for (int i = 0; i < 16; ++i) { m[i] = f32x4{0.f, 0.f, 0.f, 0.f}; v[i] = threadIdx.x * 1e-3f + i; }
  s16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + threadIdx.x + j); b[j] = (short)(0x3f00 + j); }
  const float c = 1.0001f, d = 1e-4f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      m[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), m[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(i * NV + j) & 15] = __builtin_fmaf(v[(i * NV + j) & 15], c, d);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += m[i][0] + m[i][1] + m[i][2] + m[i][3] + v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// KIND 0: v_mfma_f32_32x32x16_bf16, 1: v_mfma_f32_32x32x2_f32; 4 independent accumulators, NV fillers after each MFMA
template <int KIND, int NV>
__global__ __launch_bounds__(256) void timing32(int iters, float* out, long long* cyc) {
  f32x16 m[4];
  float v[16];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) m[i][j] = 0.f;
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  s16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + threadIdx.x + j); b[j] = (short)(0x3f00 + j); }
  const float c = 1.0001f, d = 1e-4f, fa = 1.0f + threadIdx.x * 1e-3f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) m[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), m[i & 3], 0, 0, 0);
      else m[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, c, m[i & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(i * NV + j) & 15] = __builtin_fmaf(v[(i * NV + j) & 15], c, d);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += m[i][j];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static double gauss() {
  double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
  return sqrt(-2 * log(u)) * cos(6.283185307179586 * v);
}

int main() {
  srand(1);
  float *dA, *dB, *dD, *out; long long* cyc;
  const int KMAX = 1280;
  hipMalloc(&dA, 16 * KMAX * 4); hipMalloc(&dB, 16 * KMAX * 4); hipMalloc(&dD, 256 * 4);
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
  for (int K : {128, 640, 1280}) {
    double e[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0}, ref2 = 0;
    const int trials = 64;
    for (int t = 0; t < trials; ++t) {
      std::vector<float> A(16 * K), B(K * 16), D(256);
      for (auto& x : A) x = (float)(gauss() * (t & 1 ? 1.0 : 3.0) + (t & 2 ? 0.5 : 0.0));
      for (auto& x : B) x = (float)(gauss() * 0.05);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      std::vector<double> R(256, 0.0);
      for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 16 + n]; R[i * 16 + n] = s; ref2 += s * s; }
      const int modes[4] = {0, 6, 9, 16};
      for (int q = 0; q < 4; ++q) {
        hipLaunchKernelGGL(gemm, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, modes[q]);
        hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
        for (int x = 0; x < 256; ++x) { double d = D[x] - R[x]; e[q] += d * d; if (fabs(d) > m[q]) m[q] = fabs(d); }
      }
    }
    const double rms = sqrt(ref2 / (256.0 * trials));
    printf("K=%4d  rms(D)=%.3f  error vs fp64 (rms / max, in units of rms(D)):  fp32 MFMA chain %.3e / %.3e   bf16 x6 %.3e / %.3e   bf16 x9 %.3e / %.3e   bf16 x6 + separate low accumulator %.3e / %.3e\n",
           K, rms, sqrt(e[0] / (256.0 * trials)) / rms, m[0] / rms, sqrt(e[1] / (256.0 * trials)) / rms, m[1] / rms,
           sqrt(e[2] / (256.0 * trials)) / rms, m[2] / rms, sqrt(e[3] / (256.0 * trials)) / rms, m[3] / rms);
  }
  std::vector<long long> h(256 * 4);
  const int iters = 2000;
#define TIME(NV) { hipLaunchKernelGGL(timing<NV>, dim3(256), dim3(256), 0, 0, iters, out, cyc); hipLaunchKernelGGL(timing<NV>, dim3(256), dim3(256), 0, 0, iters, out, cyc); \
    hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, 256 * 4 * 8, hipMemcpyDeviceToHost); double s = 0; for (auto x : h) s += x; \
    printf("one wave per SIMD: v_mfma_f32_16x16x32_bf16 + %d independent v_fma_f32 each: %.2f cycles per MFMA\n", NV, s / (256.0 * 4) / iters / 16); }
  TIME(0) TIME(1) TIME(2) TIME(3) TIME(4) TIME(6)
#define TIME32(KIND, NV) { hipLaunchKernelGGL((timing32<KIND, NV>), dim3(256), dim3(256), 0, 0, iters, out, cyc); hipLaunchKernelGGL((timing32<KIND, NV>), dim3(256), dim3(256), 0, 0, iters, out, cyc); \
    hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, 256 * 4 * 8, hipMemcpyDeviceToHost); double s = 0; for (auto x : h) s += x; \
    printf("one wave per SIMD: %s + %d independent v_fma_f32 each: %.2f cycles per MFMA\n", KIND ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_32x32x16_bf16", NV, s / (256.0 * 4) / iters / 16); }
  TIME32(0, 0) TIME32(0, 2) TIME32(0, 4) TIME32(0, 6) TIME32(0, 8) TIME32(0, 12)
  TIME32(1, 0) TIME32(1, 4) TIME32(1, 8) TIME32(1, 12)
  return 0;
}
