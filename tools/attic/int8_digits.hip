// Groundwork for an integer digit split of the fp32 GEMMs: a = s_a(row) * sum_i A_i 2^(-8 (i + 1)),
// w = s_w(col) * sum_j W_j 2^(-8 (j + 1)) with int8 digits (s = power of two >= the row's / column's max magnitude), the
// products of order i + j <= ORD accumulated EXACTLY in int32 on v_mfma_i32_16x16x64_i8 and combined in fp32 at the end.
// Measures the error against fp64 next to the fp32 MFMA chain and the bf16x3 split, for rows of mixed magnitude, and the
// instruction's issue rate.  Result (profiles/r02_ubench_int8_digits.txt): three digits a side are NOT enough (six
// products: 10x the fp32 chain's error, all nine: 2-5x), four a side with the ten products of order <= 3 are 7x better
// than fp32; the instruction takes 27.3 cycles (1.53x v_mfma_f32_16x16x32_bf16 for twice the K), so ten of them per K = 64
// cost 1.28x the bf16x3 split's twelve bf16 MFMAs -- what the scheme would buy is 4 instead of 6 bytes per weight only.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __host__ inline void split3(float a, unsigned short p[3]) {
  unsigned u; memcpy(&u, &a, 4);
  unsigned u1 = u & 0xFFFF0000u; float f1; memcpy(&f1, &u1, 4);
  float r1 = a - f1; unsigned v; memcpy(&v, &r1, 4);
  unsigned u2 = v & 0xFFFF0000u; float f2; memcpy(&f2, &u2, 4);
  float r2 = r1 - f2; unsigned w; memcpy(&w, &r2, 4);
  p[0] = u1 >> 16; p[1] = u2 >> 16; p[2] = w >> 16;
}
// ND balanced radix-256 digits of x / s (|x| < s / 4, s a power of two): x = s * sum_i d[i] 2^(-8 (i + 1)) up to
// s 2^(-8 ND - 1); the digits of the integer round(x / s * 2^(8 ND)), lowest first, each in [-128, 127] with carry
__device__ __host__ inline void digits(float x, float s, int nd, signed char* d) {
  long long n = llrint((double)x / (double)s * (double)(1ll << (8 * nd)));
  for (int i = nd - 1; i >= 0; --i) {
    const long long q = ((n + 128) & 255) - 128;
    d[i] = (signed char)q;
    n = (n - q) >> 8;
  }
}
__device__ __host__ inline float pow2_ge(float m) { int e; frexpf(m, &e); return m > 0.f ? ldexpf(1.f, e + 2) : 1.f; }   // in (4 m, 8 m]: the top digit stays below 64 even after a carry

// A [16][K], B [K][16]; mode 0: fp32 MFMA chain, 1: bf16x3 (six products), 2: int8 digits NA x NW, orders <= ORD
__global__ void gemm(const float* A, const float* B, float* D, int K, int mode, int NA, int NW, int ORD) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  if (mode == 0) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k + g], B[(k + g) * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(g * 4 + r) * 16 + i] = acc[r];
  } else if (mode == 1) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
      s16x8 a[3], b[3];
      for (int j = 0; j < 8; ++j) {
        unsigned short p[3];
        split3(A[i * K + k0 + g * 8 + j], p);
        for (int q = 0; q < 3; ++q) a[q][j] = (short)p[q];
        split3(B[(k0 + g * 8 + j) * 16 + i], p);
        for (int q = 0; q < 3; ++q) b[q][j] = (short)p[q];
      }
#define MM(x, y) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[x]), __builtin_bit_cast(bf16x8, b[y]), acc, 0, 0, 0)
      MM(2, 0); MM(1, 1); MM(0, 2); MM(1, 0); MM(0, 1); MM(0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(g * 4 + r) * 16 + i] = acc[r];
  } else {
    // scales: row i of A (this lane's A row), column i of B (this lane's B column); the output lane holds rows 4 g + r of
    // column i: row scales come through LDS
    __shared__ float sa[16], sw[16];
    if (g == 0) {
      float ma = 0.f, mw = 0.f;
      for (int k = 0; k < K; ++k) { ma = fmaxf(ma, fabsf(A[i * K + k])); mw = fmaxf(mw, fabsf(B[k * 16 + i])); }
      sa[i] = pow2_ge(ma); sw[i] = pow2_ge(mw);
    }
    __syncthreads();
    i32x4 acc[7];
    for (int o = 0; o < 7; ++o) acc[o] = i32x4{0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 64) {
      i32x4 a[4], b[4];
      signed char da[4][16], db[4][16];
      for (int j = 0; j < 16; ++j) {
        signed char d[4];
        digits(A[i * K + k0 + g * 16 + j], sa[i], NA, d);
        for (int q = 0; q < NA; ++q) da[q][j] = d[q];
        digits(B[(k0 + g * 16 + j) * 16 + i], sw[i], NW, d);
        for (int q = 0; q < NW; ++q) db[q][j] = d[q];
      }
      for (int q = 0; q < NA; ++q) memcpy(&a[q], da[q], 16);
      for (int q = 0; q < NW; ++q) memcpy(&b[q], db[q], 16);
      for (int x = 0; x < NA; ++x)
        for (int y = 0; y < NW; ++y)
          if (x + y <= ORD) acc[x + y] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[x], b[y], acc[x + y], 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) {
      double v = 0.0;   // exact combination reference for the int32 sums; the fp32 combination is what a kernel would do
      float f = 0.f;
      for (int o = ORD; o >= 0; --o) f = f + (float)acc[o][r] * ldexpf(1.f, -8 * (o + 2));
      (void)v;
      D[(g * 4 + r) * 16 + i] = f * sa[g * 4 + r] * sw[i];
    }
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void timing(int iters, int* out, long long* cyc) {
  i32x4 m[16];
  for (int i = 0; i < 16; ++i) m[i] = i32x4{0, 0, 0, 0};
  i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {5, 6, 7, (int)threadIdx.x};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, m[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  int s = 0;
  for (int i = 0; i < 16; ++i) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static double gauss() {
  double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
  return sqrt(-2 * log(u)) * cos(6.283185307179586 * v);
}

int main() {
  srand(2);
  float *dA, *dB, *dD; int* out; long long* cyc;
  const int KMAX = 1280;
  hipMalloc(&dA, 16 * KMAX * 4); hipMalloc(&dB, 16 * KMAX * 4); hipMalloc(&dD, 256 * 4);
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
  struct Cfg { int mode, na, nw, ord; const char* name; };
  const Cfg cfgs[] = {{0, 0, 0, 0, "fp32 MFMA chain"}, {1, 0, 0, 0, "bf16x3 (6 products)"}, {2, 3, 3, 2, "int8 3x3 digits, order <= 2 (6)"},
                      {2, 3, 3, 4, "int8 3x3 digits, all 9"}, {2, 4, 3, 3, "int8 4x3 digits, order <= 3 (9)"}, {2, 4, 4, 3, "int8 4x4 digits, order <= 3 (10)"}};
  for (int spread = 0; spread < 2; ++spread)
    for (int K : {128, 640}) {
      const int NC = sizeof(cfgs) / sizeof(cfgs[0]);
      double e[8] = {0}, m[8] = {0}, ref2 = 0;
      const int trials = 48;
      for (int t = 0; t < trials; ++t) {
        std::vector<float> A(16 * K), B(K * 16), D(256);
        for (int i = 0; i < 16; ++i)
          for (int k = 0; k < K; ++k) {
            // spread 1: a few large entries per row (what a fixed-point split is worst at) on top of small ones
            const double sc = spread ? ((rand() % 32 == 0) ? 30.0 : 0.3) : 1.0;
            A[i * K + k] = (float)(gauss() * sc);
          }
        for (auto& x : B) x = (float)(gauss() * 0.05 * (spread ? ((rand() % 16 == 0) ? 10.0 : 1.0) : 1.0));
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<double> R(256, 0.0);
        for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 16 + n]; R[i * 16 + n] = s; ref2 += s * s; }
        for (int q = 0; q < NC; ++q) {
          hipLaunchKernelGGL(gemm, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, cfgs[q].mode, cfgs[q].na, cfgs[q].nw, cfgs[q].ord);
          hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
          for (int x = 0; x < 256; ++x) { double d = D[x] - R[x]; e[q] += d * d; if (fabs(d) > m[q]) m[q] = fabs(d); }
        }
      }
      const double rms = sqrt(ref2 / (256.0 * trials));
      printf("K=%4d %s: rms(D)=%.3f; error vs fp64, rms / max in units of rms(D):\n", K, spread ? "heavy-tailed rows / columns" : "gaussian", rms);
      for (int q = 0; q < NC; ++q) printf("    %-36s %.3e / %.3e\n", cfgs[q].name, sqrt(e[q] / (256.0 * trials)) / rms, m[q] / rms);
    }
  std::vector<long long> h(256 * 4);
  const int iters = 2000;
  hipLaunchKernelGGL(timing<0>, dim3(256), dim3(256), 0, 0, iters, out, cyc);
  hipLaunchKernelGGL(timing<0>, dim3(256), dim3(256), 0, 0, iters, out, cyc);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), cyc, 256 * 4 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : h) s += x;
  printf("one wave per SIMD: v_mfma_i32_16x16x64_i8, 16 independent accumulators: %.2f cycles per MFMA (K = 64: 32768 MAC)\n", s / (256.0 * 4) / iters / 16);
  return 0;
}
