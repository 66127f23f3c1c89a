// fp32 GEMM emulated on the fp16 matrix pipe with TWO pieces per operand: x = x0 + x1, x0 = RN16(x), x1 = RN16(x - x0)
// (|x - x0 - x1| <= 2^-24 |x| as long as x1 is not below fp16's denormal step), a*b ~ a1 b0 + a0 b1 + a0 b0 on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation: 3 MFMAs and 4 bytes per weight instead of bf16x3's 6 and 6.
// fp16 has 5 exponent bits, so B's columns are scaled by a power of two (max |b| of a column in [2^14, 2^15)) and A by
// 2^SA; what happens to the low pieces that fall below 2^-14 depends on whether the MFMA honours fp16 denormals --
// probed first.  Measures (1) denormal handling, (2) error of fp32 chain / bf16x3 (6 products) / f16x2 (3 products, one
// accumulator) / f16x2 with the two low products in their own accumulator / f16x2 with all 4 products against fp64 for
// activations of rms 1, 0.1, 0.01, 30, (3) cycles per v_mfma_f32_16x16x32_f16 and per dependent triple.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float a, unsigned short p[3]) {
  unsigned u; memcpy(&u, &a, 4);
  unsigned u1 = u & 0xFFFF0000u; float f1; memcpy(&f1, &u1, 4);
  float r1 = a - f1; unsigned v; memcpy(&v, &r1, 4);
  unsigned u2 = v & 0xFFFF0000u; float f2; memcpy(&f2, &u2, 4);
  float r2 = r1 - f2; unsigned w; memcpy(&w, &r2, 4);
  p[0] = u1 >> 16; p[1] = u2 >> 16; p[2] = w >> 16;
}
__device__ inline void split2(float a, _Float16 p[2]) {
  p[0] = (_Float16)a;                 // v_cvt_f16_f32: round to nearest even
  p[1] = (_Float16)(a - (float)p[0]);
}

// A [16][K], B [K][16] (already column-scaled by the host: bs[n]); D [16][16] in true units.
// mode 0: fp32 16x16x4 chain; 6: bf16x3; 3: f16x2, 3 products, one accumulator; 13: f16x2, low products in their own
// accumulator; 4: f16x2, all four products
__global__ void gemm(const float* A, const float* B, const float* bs, float* D, int K, int mode, float sa) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, lo = {0.f, 0.f, 0.f, 0.f};
  if (mode == 0) {
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k + g], B[(k + g) * 16 + i], acc, 0, 0, 0);
  } else if (mode == 6) {
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
      MM(2, 0, acc); MM(1, 1, acc); MM(0, 2, acc); MM(1, 0, acc); MM(0, 1, acc); MM(0, 0, acc);
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += 32) {
      f16x8 a[2], b[2];
      for (int j = 0; j < 8; ++j) {
        _Float16 p[2];
        split2(A[i * K + k0 + g * 8 + j] * sa, p);
        a[0][j] = p[0]; a[1][j] = p[1];
        split2(B[(k0 + g * 8 + j) * 16 + i], p);
        b[0][j] = p[0]; b[1][j] = p[1];
      }
#define MH(x, y, c) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[x], b[y], c, 0, 0, 0)
      if (mode == 4) MH(1, 1, acc);
      if (mode == 13) { MH(1, 0, lo); MH(0, 1, lo); MH(0, 0, acc); }
      else { MH(1, 0, acc); MH(0, 1, acc); MH(0, 0, acc); }
    }
    if (mode == 13) acc += lo;
  }
  const float s = mode == 0 || mode == 6 ? 1.f / bs[i] : 1.f / (bs[i] * sa);
  for (int r = 0; r < 4; ++r) D[(g * 4 + r) * 16 + i] = acc[r] * s;
}

// denormal probe: every A element = va, every B element = vb, K = 32: D = 32 va vb if the inputs are honoured
__global__ void probe(float va, float vb, float* D) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)va; b[j] = (_Float16)vb; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) D[0] = acc[0];
}

// NDEP MFMAs in a row on the SAME accumulator, NACC accumulators round robin
template <int NACC, int NDEP>
__global__ __launch_bounds__(256) void timing(int iters, float* out, long long* cyc) {
  f32x4 m[NACC];
  for (int i = 0; i < NACC; ++i) m[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(1.f + 0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.5f + 0.01f * j); }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
      for (int d = 0; d < NDEP; ++d) {
        m[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, m[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static double gauss() {
  double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
  return sqrt(-2 * log(u)) * cos(6.283185307179586 * v);
}

int main() {
  srand(1);
  float *dA, *dB, *dS, *dD, *out; long long* cyc;
  const int KMAX = 1280;
  hipMalloc(&dA, 16 * KMAX * 4); hipMalloc(&dB, 16 * KMAX * 4); hipMalloc(&dD, 256 * 4); hipMalloc(&dS, 64);
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
  {
    float d;
    const float pa[4] = {ldexpf(1.f, -20), ldexpf(1.f, -24), ldexpf(1.f, -15), ldexpf(1.f, -14)};
    for (float va : pa) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, va, 1024.f, dD);
      hipMemcpy(&d, dD, 4, hipMemcpyDeviceToHost);
      printf("denormal probe: A = %.3e (fp16 %s), B = 1024, K = 32: D = %.6e, expected %.6e -> %s\n", va,
             va < ldexpf(1.f, -14) ? "denormal" : "normal", d, 32.0 * va * 1024.0, d == 32.f * va * 1024.f ? "honoured" : "FLUSHED/other");
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, 1024.f, va, dD);
      hipMemcpy(&d, dD, 4, hipMemcpyDeviceToHost);
      printf("denormal probe: B = %.3e, A = 1024: D = %.6e -> %s\n", va, d, d == 32.f * va * 1024.f ? "honoured" : "FLUSHED/other");
    }
  }
  for (int K : {128, 640, 1280}) {
    for (double amag : {1.0, 0.1, 0.01, 30.0}) {
      for (float sa : {1.f, 8.f}) {
        const int NM = 5;
        const int modes[NM] = {0, 6, 3, 13, 4};
        double e[NM] = {0}, m[NM] = {0}, ref2 = 0;
        const int trials = 32;
        for (int t = 0; t < trials; ++t) {
          std::vector<float> A(16 * K), B(K * 16), BS(K * 16), D(256), S(16);
          for (auto& x : A) x = (float)((gauss() * (t & 1 ? 1.0 : 3.0) + (t & 2 ? 0.5 : 0.0)) * amag);
          for (auto& x : B) x = (float)(gauss() * 0.05 * exp(-8.0 * rand() / RAND_MAX));   // 3.5 decades inside a column
          for (int n = 0; n < 16; ++n) {
            float mx = 0.f;
            for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(B[k * 16 + n]));
            int ex; frexpf(mx, &ex);                 // mx = f * 2^ex, f in [0.5, 1)
            S[n] = ldexpf(1.f, 15 - ex);             // scaled max in [2^14, 2^15)
            for (int k = 0; k < K; ++k) BS[k * 16 + n] = B[k * 16 + n] * S[n];
          }
          hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
          hipMemcpy(dB, BS.data(), BS.size() * 4, hipMemcpyHostToDevice);
          hipMemcpy(dS, S.data(), 64, hipMemcpyHostToDevice);
          std::vector<double> R(256, 0.0);
          for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 16 + n]; R[i * 16 + n] = s; ref2 += s * s; }
          for (int q = 0; q < NM; ++q) {
            hipLaunchKernelGGL(gemm, dim3(1), dim3(64), 0, 0, dA, dB, dS, dD, K, modes[q], sa);
            hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
            for (int x = 0; x < 256; ++x) { double d = D[x] - R[x]; e[q] += d * d; if (fabs(d) > m[q]) m[q] = fabs(d); }
          }
        }
        const double rms = sqrt(ref2 / (256.0 * trials));
        printf("K=%4d rms(A)~%5.2f SA=%g  rms error / rms(D):  fp32 chain %.2e  bf16x3 %.2e  f16x2(3) %.2e  f16x2(3, low acc) %.2e  f16x2(4) %.2e   max: %.2e %.2e %.2e %.2e %.2e\n",
               K, 2.3 * amag, sa, sqrt(e[0] / (256.0 * trials)) / rms, sqrt(e[1] / (256.0 * trials)) / rms, sqrt(e[2] / (256.0 * trials)) / rms,
               sqrt(e[3] / (256.0 * trials)) / rms, sqrt(e[4] / (256.0 * trials)) / rms, m[0] / rms, m[1] / rms, m[2] / rms, m[3] / rms, m[4] / rms);
      }
    }
  }
  std::vector<long long> h(256 * 4);
  const int iters = 2000;
#define TIME(NACC, NDEP) { hipLaunchKernelGGL((timing<NACC, NDEP>), dim3(256), dim3(256), 0, 0, iters, out, cyc); hipLaunchKernelGGL((timing<NACC, NDEP>), dim3(256), dim3(256), 0, 0, iters, out, cyc); \
    hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, 256 * 4 * 8, hipMemcpyDeviceToHost); double s = 0; for (auto x : h) s += x; \
    printf("one wave per SIMD: v_mfma_f32_16x16x32_f16, %d accumulators x %d dependent in a row: %.2f cycles per MFMA\n", NACC, NDEP, s / (256.0 * 4) / iters / (NACC * NDEP)); }
  TIME(16, 1) TIME(8, 1) TIME(4, 1) TIME(2, 1) TIME(1, 1) TIME(8, 3) TIME(2, 3) TIME(4, 3)
  return 0;
}
