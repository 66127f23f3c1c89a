// Would ONE 512-register wave per SIMD that interleaves the MFMAs of one 4-sample half with the GroupNorm + Mish epilogue of the
// other beat TWO 256-register workgroups per CU running in lockstep?    The loop body is downs.2's
// 128 -> 128 conv exactly as unet_kernel runs it (rd_store2 -> barrier -> rd_taps -> rd_gn_mish, this file includes unet.hip):
//   base: 512 workgroups of 4 samples, 2 per CU  (hipcc ... -o fat_base)
//   fat : 256 workgroups of 8 samples = two halves; per phase the taps of one half and the epilogue + slab store of the other sit
//         in one basic block and a sched_group_barrier pipeline asks for 3 MFMAs : 4 VALU  (hipcc -DFAT ... -o fat_fat)
// Prints the time per conv of 8 samples per CU.  Build (both): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DFAT]
//   tools/ubench/fatwave_conv.hip mmd_amd/csrc/{guide,api,multi_agent,postprocess}.hip -o build_tmp/fat_{base,fat}
#ifdef FAT
#define MMD_VB3_LOOSE
#define MMD_NO_PIN
#endif
#include "../../mmd_amd/csrc/unet.hip"

#include <cstdio>
#include <random>

namespace mmd {
using G128 = RdGeo<128>;

struct ConvP { const uint4* w; const float* par; };   // par: bias[128] gamma[128] beta[128] tb[128] isc[128]

template <class ACC>
__device__ __forceinline__ void epilogue(ACC& acc, const Epi<2>& e) {
  const float one4[4] = {1.f, 1.f, 1.f, 1.f};
  const float t0 = e.tb[0], t1 = e.tb[1];
  rd_gn_mish<2, 256, true>(acc, e.b, e.g, e.be, e.is, one4, act_scale(1.f), [&](int, int t, int) { return t ? t1 : t0; });
}

#ifndef FAT
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_loop(ConvP p, float* out, int nconv) {
  __shared__ __attribute__((aligned(16))) float lds[G128::BYTES / 4 + 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 15, g = lane >> 4;
  const int c0 = 32 * wave + 2 * n;
  char* const slab = reinterpret_cast<char*>(lds);
  const char* const va = slab + g * G128::G + n * 16;
  char* const vs = slab + wave * G128::G + (n >> 2) * G128::BX + (2 + 4 * g) * 16 + (n & 3) * 4;
  f32x4 acc[4][2];
  for (int s = 0; s < 4; ++s) for (int t = 0; t < 2; ++t) for (int r = 0; r < 4; ++r) acc[s][t][r] = 0.01f * ((threadIdx.x * 7 + s * 3 + t + r + blockIdx.x) % 97) - 0.5f;
  rd_zero_halo<G128>(slab);
  const u32x4* wp[2] = {reinterpret_cast<const u32x4*>(p.w) + (size_t)(2 * wave) * G128::FRAGS5 * 64 + lane,
                        reinterpret_cast<const u32x4*>(p.w) + (size_t)(2 * wave + 1) * G128::FRAGS5 * 64 + lane};
  __syncthreads();
  for (int k = 0; k < nconv; ++k) {
    const Epi<2> e = epi_load<2>(p.par, p.par + 128, p.par + 256, p.par + 384, p.par + 512, c0);
    u32x4 ring[3][2][2];
    rd_ring_load<G128, 2, 3>(ring, wp);
    rd_store2<G128>(vs, acc);
    __syncthreads();
    rd_taps<G128, 2, 0, 5, true, false, 4, 3>(acc, acc, va, wp, wp, ring);
    epilogue(acc, e);
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int t = 0; t < 2; ++t) s += acc[i][t][0] + acc[i][t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
constexpr int SAMPLES_PER_WG = 4;
#else
// one scheduling pipeline for a phase: per step (20) and half step (2): 4 DS reads, then 4 x (3 MFMAs, 4 VALU); per step 4 VMEM reads
__device__ __forceinline__ void phase_pipeline() {
#ifdef NO_SGB
  return;
#endif
#pragma unroll
  for (int st = 0; st < 20; ++st) {
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
  }
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_loop(ConvP p, float* out, int nconv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];     // two halves: 2 x G128::BYTES
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 15, g = lane >> 4;
  const int c0 = 32 * wave + 2 * n;
  char* const slabA = reinterpret_cast<char*>(lds);
  char* const slabB = slabA + G128::BYTES;
  const int va_off = g * G128::G + n * 16, vs_off = wave * G128::G + (n >> 2) * G128::BX + (2 + 4 * g) * 16 + (n & 3) * 4;
  f32x4 accA[4][2], accB[4][2];
  for (int s = 0; s < 4; ++s) for (int t = 0; t < 2; ++t) for (int r = 0; r < 4; ++r) {
    accA[s][t][r] = 0.01f * ((threadIdx.x * 7 + s * 3 + t + r + blockIdx.x) % 97) - 0.5f;
    accB[s][t][r] = 0.01f * ((threadIdx.x * 5 + s * 2 + t + r + blockIdx.x) % 89) - 0.4f;
  }
  rd_zero_halo<G128>(slabA);
  rd_zero_halo<G128>(slabB);
  const u32x4* wp[2] = {reinterpret_cast<const u32x4*>(p.w) + (size_t)(2 * wave) * G128::FRAGS5 * 64 + lane,
                        reinterpret_cast<const u32x4*>(p.w) + (size_t)(2 * wave + 1) * G128::FRAGS5 * 64 + lane};
  __syncthreads();
  rd_store2<G128>(slabA + vs_off, accA);
  __syncthreads();
  for (int k = 0; k < nconv; ++k) {
    const Epi<2> e = epi_load<2>(p.par, p.par + 128, p.par + 256, p.par + 384, p.par + 512, c0);
    {   // phase 1: taps of half A; epilogue + slab store of half B (its previous conv)
      u32x4 ring[3][2][2];
      rd_ring_load<G128, 2, 3>(ring, wp);
      rd_taps<G128, 2, 0, 5, true, false, 4, 3>(accA, accA, slabA + va_off, wp, wp, ring);
      epilogue(accB, e);
      rd_store2<G128>(slabB + vs_off, accB);
      phase_pipeline();
      __syncthreads();
    }
    {   // phase 2: taps of half B; epilogue + slab store of half A
      u32x4 ring[3][2][2];
      rd_ring_load<G128, 2, 3>(ring, wp);
      rd_taps<G128, 2, 0, 5, true, false, 4, 3>(accB, accB, slabB + va_off, wp, wp, ring);
      epilogue(accA, e);
      rd_store2<G128>(slabA + vs_off, accA);
      phase_pipeline();
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int t = 0; t < 2; ++t) s += accA[i][t][0] + accA[i][t][3] + accB[i][t][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
constexpr int SAMPLES_PER_WG = 8;
#endif
}  // namespace mmd

int main() {
  using namespace mmd;
  const int nconv = 64, nb = 2048 / SAMPLES_PER_WG;
  const size_t wbytes = (size_t)(8 * G128::FRAGS5 + 8) * 64 * 16;
  std::vector<uint16_t> hw(wbytes / 2);
  std::mt19937 rng(1);
  for (auto& v : hw) { const _Float16 f = (_Float16)(((int)(rng() % 2001) - 1000) * 1e-4f); memcpy(&v, &f, 2); }
  std::vector<float> par(5 * 128);
  for (int i = 0; i < 128; ++i) { par[i] = 0.01f * (i % 7); par[128 + i] = 1.f + 0.01f * (i % 5); par[256 + i] = 0.02f * (i % 3); par[384 + i] = 0.05f; par[512 + i] = 1.f; }
  uint4* dw; float* dpar; float* dout;
  hipMalloc(&dw, wbytes); hipMalloc(&dpar, par.size() * 4); hipMalloc(&dout, (size_t)nb * 256 * 4);
  hipMemcpy(dw, hw.data(), wbytes, hipMemcpyHostToDevice); hipMemcpy(dpar, par.data(), par.size() * 4, hipMemcpyHostToDevice);
  ConvP p{dw, dpar};
#ifdef FAT
  const size_t shm = 2 * G128::BYTES + 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_loop), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
#else
  const size_t shm = 0;
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(conv_loop, dim3(nb), dim3(256), shm, 0, p, dout, nconv);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> ho(8); hipMemcpy(ho.data(), dout, 32, hipMemcpyDeviceToHost);
    printf("%s: %d workgroups x %d samples, %d convs: %.1f us -> %.2f us per conv of 8 samples per CU  (check %.4f, err %s)\n",
#ifdef FAT
           "fat ",
#else
           "base",
#endif
           nb, SAMPLES_PER_WG, nconv, ms * 1e3, ms * 1e3 / nconv, ho[3], hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
