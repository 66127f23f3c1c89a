// Do the two waves of a SIMD overlap an fp32 MFMA stream with a VALU stream?  One 512-thread workgroup per CU: waves 0..3
// land on SIMDs 0..3 and waves 4..7 on the same SIMDs again.  Role A (waves 0..3) / role B (waves 4..7) in {idle, MFMA
// (v_mfma_f32_16x16x4_f32, 16 independent accumulators), VALU (independent v_fma_f32 chains), TRANS (v_exp_f32)}; each
// active wave runs a fixed amount of work and reports its own cycles (s_memtime).  Usage: mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { IDLE = 0, MFMA = 1, VALU = 2, TRANS = 3, PK = 4, RCP = 5, LDSR = 6 };
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// -DF16MFMA: the MFMA role issues v_mfma_f32_16x16x32_f16 (the matrix core proper; 16 cycles) instead of v_mfma_f32_16x16x4_f32
// (32 cycles, fp32 at the vector rate)
// -DF16MFMA32: v_mfma_f32_32x32x16_f16 (8 passes = 32 cycles, 16 accumulator registers; 4 accumulators x 4 per iteration)
#if defined(F16MFMA32)
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ACC_T f32x16
#define NACC 4
#define ACC_ZERO(x) for (int q_ = 0; q_ < 16; ++q_) x[q_] = 0.f
#define MFMA_OP(acc) __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0)
#define MFMA_DECL f16x8 ah, bh; for (int q = 0; q < 8; ++q) { ah[q] = (_Float16)(1.0f + threadIdx.x * 1e-3f); bh[q] = (_Float16)0.5f; }
#elif defined(F16MFMA)
#define MFMA_OP(acc) __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0)
#define MFMA_DECL f16x8 ah, bh; for (int q = 0; q < 8; ++q) { ah[q] = (_Float16)(1.0f + threadIdx.x * 1e-3f); bh[q] = (_Float16)0.5f; }
#else
#define MFMA_OP(acc) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)
#define MFMA_DECL
#endif
#ifndef ACC_T
#define ACC_T f32x4
#define NACC 16
#define ACC_ZERO(x) x = f32x4{0.f, 0.f, 0.f, 0.f}
#endif

__device__ __forceinline__ void run_mfma(int iters, float* out) {
  ACC_T m[NACC];
  for (int i = 0; i < NACC; ++i) ACC_ZERO(m[i]);
  float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  (void)a; (void)b;
  MFMA_DECL
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      m[i % NACC] = MFMA_OP(m[i % NACC]);
#ifdef MFMA_NOP
      // yield the issue port while the matrix pipe is busy: does the other wave's VALU stream get the cycles?
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop %0" ::"n"(MFMA_NOP));
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
  out[threadIdx.x] = s;
}
__device__ __forceinline__ void run_valu(int iters, float* out) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float c = 1.0001f, d = 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], c, d);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[threadIdx.x] = s;
}
__device__ __forceinline__ void run_trans(int iters, float* out) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.01f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.25f;
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[threadIdx.x] = s;
}

__device__ __forceinline__ void run_pk(int iters, float* out) {   // 16 v_pk_fma_f32 = 32 fp32 FMAs per iteration
  f32x2 v[16];
  for (int i = 0; i < 16; ++i) v[i] = f32x2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f - i};
  const f32x2 c = {1.0001f, 0.9999f}, d = {1e-4f, -1e-4f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_elementwise_fma(v[i], c, d);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
  out[threadIdx.x] = s;
}
__device__ __forceinline__ void run_rcp(int iters, float* out) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + threadIdx.x * 1e-3f + i * 0.01f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_rcpf(v[i]);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[threadIdx.x] = s;
}
__device__ __forceinline__ void run_ldsr(int iters, float* out) {   // 16 ds_read_b128 per iteration
  __shared__ float4 buf[8][64 * 17];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int i = 0; i < 17; ++i) buf[w][l * 17 + i] = make_float4(l, i, 0.f, 1.f);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 t = buf[w][l * 17 + ((i + it) & 15)];
      acc.x += t.x; acc.y += t.y;
    }
  }
  out[threadIdx.x] = acc.x + acc.y;
}

// MIX: one wave interleaves NV independent v_fma_f32 after every MFMA (does VALU issue in the MFMA's shadow?)
template <int NV>
__device__ __forceinline__ void run_mix(int iters, float* out) {
  ACC_T m[NACC];
  float v[16];
  for (int i = 0; i < NACC; ++i) ACC_ZERO(m[i]);
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  (void)a; (void)b;
  MFMA_DECL
  const float c = 1.0001f, d = 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      m[i % NACC] = MFMA_OP(m[i % NACC]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(i * NV + j) & 15] = __builtin_fmaf(v[(i * NV + j) & 15], c, d);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void kmix(int nv, int iters, float* out, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (nv == 2) run_mix<2>(iters, out + blockIdx.x * 512);
    else if (nv == 4) run_mix<4>(iters, out + blockIdx.x * 512);
    else if (nv == 6) run_mix<6>(iters, out + blockIdx.x * 512);
    else if (nv == 8) run_mix<8>(iters, out + blockIdx.x * 512);
  }
  const long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// itersB: role B's own iteration count (so that B's work can be sized to finish inside A's run); prioB: s_setprio of role B
__global__ __launch_bounds__(512) void k(int roleA, int roleB, int iters, float* out, long long* cyc, int itersB = 0,
                                         int prioB = 0) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? roleA : roleB;
  if (wave >= 4) {
    if (itersB) iters = itersB;
    if (prioB == 1) __builtin_amdgcn_s_setprio(1);
    if (prioB == 3) __builtin_amdgcn_s_setprio(3);
  }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (role == MFMA) run_mfma(iters, out + blockIdx.x * 512);
  else if (role == VALU) run_valu(iters, out + blockIdx.x * 512);
  else if (role == TRANS) run_trans(iters, out + blockIdx.x * 512);
  else if (role == PK) run_pk(iters, out + blockIdx.x * 512);
  else if (role == RCP) run_rcp(iters, out + blockIdx.x * 512);
  else if (role == LDSR) run_ldsr(iters, out + blockIdx.x * 512);
  const long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  const int nb = 256, iters = 2000;
  float* out; long long* cyc;
  hipMalloc(&out, nb * 512 * sizeof(float));
  hipMalloc(&cyc, nb * 8 * sizeof(long long));
  const char* names[] = {"idle", "MFMA", "VALU", "TRANS", "PK", "RCP", "LDSR"};
  std::vector<long long> h(nb * 8);
  for (int a = 0; a < 7; ++a)
    for (int b = 0; b < 7; ++b) {
      if (a == IDLE && b == IDLE) continue;
      if (a != IDLE && a != MFMA && b != IDLE && b != MFMA && a != b) continue;   // alone, paired with MFMA, or with itself
      hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, a, b, iters, out, cyc);
      hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, a, b, iters, out, cyc);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), cyc, nb * 8 * sizeof(long long), hipMemcpyDeviceToHost);
      double ca = 0, cb = 0;
      for (int i = 0; i < nb; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? ca : cb) += h[i * 8 + w];
      ca /= nb * 4; cb /= nb * 4;
      printf("A=%-5s B=%-5s : A %8.1f cycles per 16-op iteration, B %8.1f\n", names[a], names[b], a ? ca / iters : 0.0, b ? cb / iters : 0.0);
    }
  // priority / starvation probes: B does 8x the iterations (so B alone = 8 * 60 = 483 cycles per A iteration, about A's
  // 512) and we report both waves' TOTAL cycles: "time-shared" => A ~ 512 + 483, "B hidden in A's shadow" => both ~ 512
  for (int prio = 0; prio <= 3; prio += (prio ? 2 : 1)) {
    for (int role = VALU; role <= LDSR; ++role) {
      const int mult = role == VALU ? 8 : role == TRANS ? 2 : role == PK ? 4 : role == RCP ? 3 : 1;
      hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, MFMA, role, iters, out, cyc, iters * mult, prio);
      hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, MFMA, role, iters, out, cyc, iters * mult, prio);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), cyc, nb * 8 * sizeof(long long), hipMemcpyDeviceToHost);
      double ca = 0, cb = 0;
      for (int i = 0; i < nb; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? ca : cb) += h[i * 8 + w];
      ca /= nb * 4; cb /= nb * 4;
      printf("prio(B)=%d A=MFMA x%d, B=%-5s x%d : A total %9.0f  B total %9.0f  (per A iteration: A %.1f, B %.1f)\n", prio, iters,
             names[role], iters * mult, ca, cb, ca / iters, cb / iters);
    }
  }
  for (int nv = 2; nv <= 8; nv += 2) {   // (nv = 1, 3 with the f16 MFMA: see kmix)
    hipLaunchKernelGGL(kmix, dim3(nb), dim3(512), 0, 0, nv, iters, out, cyc);
    hipLaunchKernelGGL(kmix, dim3(nb), dim3(512), 0, 0, nv, iters, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, nb * 8 * sizeof(long long), hipMemcpyDeviceToHost);
    double ca = 0;
    for (int i = 0; i < nb; ++i) for (int w = 0; w < 4; ++w) ca += h[i * 8 + w];
    printf("same wave: 16 x (MFMA + %d v_fma_f32): %.1f cycles per iteration (MFMA alone 512, VALU alone %.0f)\n", nv,
           ca / (nb * 4) / iters, 3.77 * 16 * nv);
  }
  return 0;
}
