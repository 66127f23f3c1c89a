// Would ONE 512-register wave per SIMD that interleaves the MFMAs of one 4-sample half with the GroupNorm + Mish epilogue of the
// other beat TWO 256-register workgroups per CU running in lockstep?  (HISTORY.md section 7, item 1.)  The loop body is downs.2's
// 128 -> 128 conv exactly as unet_kernel runs it (rd_store2 -> barrier -> rd_taps -> rd_gn_mish, this file includes unet.hip):
//   base: 512 workgroups of 4 samples, 2 per CU  (hipcc ... -o fat_base)
//   fat : 256 workgroups of 8 samples = two halves; per phase the taps of one half and the epilogue + slab store of the other sit
//         in one basic block and a sched_group_barrier pipeline asks for 3 MFMAs : 4 VALU  (hipcc -DFAT ... -o fat_fat)
// Prints the time per conv of 8 samples per CU.  Build (both): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DFAT]
//   tools/ubench/fatwave_conv.hip mmd_amd/csrc/{unet_layers,guide,api,multi_agent,postprocess}.hip -o build_tmp/fat_{base,fat}
#ifdef FAT
#define MMD_VB3_LOOSE
#define MMD_NO_PIN
#endif
#ifndef VPT
#define VPT 4                 // VALU instructions asked for after each MFMA triple (HOOK variant)
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

#ifdef DPPTAPS
// The taps of a k = 5 conv at L = 16 are row-shifted views of the slab: M tile = sample = 16 positions = ONE 16-lane DPP row of
// the A fragment, so the fragment of tap d is the centre fragment shifted by d lanes inside every row (row_shr / row_shl with
// zero fill = the zero halo).  One ds_read_b128 per (chunk, M tile, piece) + 16 v_mov_dpp per other tap instead of five reads:
// LDS -> VGPR traffic / 5 (the conv loop is bound by exactly that traffic competing with the MFMAs for the register file,
// profiles/r03_ubench_conv_parts.txt).  Steps run chunk-major: (kc, tap).
template <int D>
__device__ __forceinline__ u32x4 dpp_rows(const u32x4& v) {
  if constexpr (D == 0) return v;
  constexpr int ctrl = D > 0 ? 0x100 + D : 0x110 - D;          // row_shl:D (lane i <- lane i + D) / row_shr:-D
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[i], ctrl, 0xf, 0xf, true);
  return r;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <class GEO, int NT, int MT, int RD>
__device__ __forceinline__ void rd_taps_dpp(f32x4 (&acc)[MT][NT], const char* va, const u32x4* const (&w)[NT], u32x4 (&b)[RD][NT][2]) {
  constexpr int KC = GEO::KC, STEPS = 5 * KC;
  // centre fragments of chunk kc, per M-tile PAIR: the pair that is not multiplying is re-loaded for the next chunk
  u32x4 c[MT][2];
  auto load_c = [&](int m0, int kc) {
#pragma unroll
    for (int m = m0; m < m0 + 2; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q) c[m][q] = *reinterpret_cast<const u32x4*>(va + q * GEO::PS + kc * GEO::BX + (GEO::tile_row(m) + 2) * 16);
  };
  load_c(0, 0);
  load_c(2, 0);
  MMD_PIN_LOADS();
  // order inside a chunk: the five taps of pair 0, then the five taps of pair 1; pair 0's next-chunk fragments are requested
  // when pair 1 starts, pair 1's when the next chunk's pair 0 starts.  Weight steps: 10 per chunk (pair, tap) -- the B fragment
  // of a tap is loaded twice per chunk ... no: a weight step is shared by both pairs, so the taps of both pairs run inside it.
  static_for<0, KC>([&](auto kcc) {
    constexpr int kc = decltype(kcc)::value;
    static_for<0, 5>([&](auto tc) {
      constexpr int tap = decltype(tc)::value, q = kc * 5 + tap, ri = q % RD;
      constexpr bool zero = q == 0;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const u32x4 a[2] = {dpp_rows<tap - 2>(c[m][0]), dpp_rows<tap - 2>(c[m][1])};
#pragma unroll
        for (int t = 0; t < NT; ++t) vb_three<zero>(acc[m][t], a, b[ri][t]);
        // after the LAST tap's use of a pair its registers take the next chunk's fragments
        if constexpr (tap == 4 && kc + 1 < KC) {
          if (m == 1) { load_c(0, kc + 1); MMD_PIN_LOADS(); }
          if (m == 3) { load_c(2, kc + 1); MMD_PIN_LOADS(); }
        }
      }
      if constexpr (q + RD < STEPS) rd_load_b<GEO, NT>(b[ri], w, q + RD);   // (the pack would be chunk-major: sequential)
      MMD_PIN_LOADS();
    });
  });
}
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
  int woff[2] = {(2 * wave) * G128::FRAGS5 * 64 + lane, (2 * wave + 1) * G128::FRAGS5 * 64 + lane};
  __syncthreads();
  for (int k = 0; k < nconv; ++k) {
    // (per-conv pointers, as in the kernel: no address hoisting out of the loop; the barrier sits on the OFFSETS so that the
    // loads stay global_load -- behind an opaque pointer they become flat_load, which also counts on lgkmcnt)
    asm volatile("" : "+v"(woff[0]), "+v"(woff[1]));
    const u32x4* wp[2] = {reinterpret_cast<const u32x4*>(p.w) + woff[0], reinterpret_cast<const u32x4*>(p.w) + woff[1]};
    const Epi<2> e = epi_load<2>(p.par, p.par + 128, p.par + 256, p.par + 384, p.par + 512, c0);
    u32x4 ring[3][2][2];
    // chunk-major step order: sequence index q = 5 kc + tap reads weight step tap KC + kc
#pragma unroll
    for (int q = 0; q < 3; ++q) rd_load_b<G128, 2>(ring[q], wp, q);
    rd_store2<G128>(vs, acc);
    __syncthreads();
    rd_taps_dpp<G128, 2, 4, 3>(acc, va, wp, ring);
    epilogue(acc, e);
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int t = 0; t < 2; ++t) s += acc[i][t][0] + acc[i][t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
constexpr int SAMPLES_PER_WG = 4;
#elif defined(PARTS)
// What bounds the conv?  The base loop with parts switched off (-DPARTS=<mask>): 1 MFMAs, 2 A fragments from the LDS, 4 weight
// fragments from L2 / L1, 8 epilogue (GroupNorm + Mish), 16 slab store, 32 the two barriers.  Loads that feed no MFMA are folded
// into the accumulators with one xor each so that they stay.
// -DADEPTH=<n> (default 2): the A fragments of a half step are requested n - 1 half steps ahead (n buffers of 16 registers; the
// kernel double-buffers: is the LDS latency of one half step exposed when eight waves read at once?)
#ifndef ADEPTH
#define ADEPTH 2
#endif
#ifdef SPREAD
// -DSPREAD: the operand fetches of a half step are NOT issued as a bunch in front of its twelve MFMAs: one A read (ds_read_b128) goes in
// front of every MFMA triple (the four fragments of the NEXT half step), one weight fragment (global_load_dwordx4) behind every second
// triple (the ring slot freed by the PREVIOUS step is refilled during this one: the same distance as the kernel's refill after the
// step).  A single wave per SIMD has nobody to fill the issue cycles a bunch of eight 64-lane memory instructions takes.
template <class GEO, int NT, int MT, int RD>
__device__ __forceinline__ void rd_taps_parts(f32x4 (&acc)[MT][NT], const char* va, const u32x4* const (&w)[NT], u32x4 (&b)[RD][NT][2]) {
  constexpr int KC = GEO::KC, STEPS = 5 * KC, HP = MT / 2, HS = STEPS * HP;
  static_assert(NT == 2 && ADEPTH == 2, "spread form: four triples per half step, double-buffered A");
  u32x4 a[2][2][2];
  auto a_ptr = [&](int hs, int sm, int q) {
    const int st = hs / HP, hp = hs % HP, tap = st / KC, kc = st % KC;
    return reinterpret_cast<const u32x4*>(va + q * GEO::PS + kc * GEO::BX + (GEO::tile_row(2 * hp + sm) + tap) * 16);
  };
#pragma unroll
  for (int sm = 0; sm < 2; ++sm)
#pragma unroll
    for (int q = 0; q < 2; ++q) a[0][sm][q] = *a_ptr(0, sm, q);
  MMD_PIN_LOADS();
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const int ri = st % RD;
    const bool zero = st == 0;
#pragma unroll
    for (int hp = 0; hp < HP; ++hp) {
      const int hs = st * HP + hp, cur = hs & 1, nx = hs + 1 < HS ? hs + 1 : HS - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {                       // triple j = (sm, t) = (j >> 1, j & 1)
        if (PARTS & 2) a[cur ^ 1][j >> 1][j & 1] = *a_ptr(nx, j >> 1, j & 1);      // the next half step's fragment (sm, q) = (j >> 1, j & 1)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (zero) vb_three<true>(acc[2 * hp + (j >> 1)][j & 1], a[cur][j >> 1], b[ri][j & 1]);
        else vb_three<false>(acc[2 * hp + (j >> 1)][j & 1], a[cur][j >> 1], b[ri][j & 1]);
        // refill the slot of step st - 1 with step st - 1 + RD: fragment (t, q) = the k-th of this step's four load slots
        const int k = hp * 2 + (j >> 1);                  // 0 .. 3 over the step's eight triples (behind triples 1, 3, 5, 7)
        if ((PARTS & 4) && (j & 1) && st >= 1 && st - 1 + RD < STEPS && (HP == 2 || true)) {
          const int rs = (st - 1) % RD, t = k >> 1, q = k & 1;
          if (HP == 2) b[rs][t][q] = w[t][((st - 1 + RD) * 2 + q) * 64];
        }
      }
    }
    MMD_PIN_LOADS();
  }
}
#else
template <class GEO, int NT, int MT, int RD>
__device__ __forceinline__ void rd_taps_parts(f32x4 (&acc)[MT][NT], const char* va, const u32x4* const (&w)[NT], u32x4 (&b)[RD][NT][2],
                                              const char* g_ldsb = nullptr) {
  constexpr int KC = GEO::KC, STEPS = 5 * KC, HP = MT / 2, HS = STEPS * HP;
  u32x4 a[ADEPTH][2][2];
  auto load_hs = [&](int slot, int hs) {     // half step hs = (tap, kc, hp); past the end: a valid, unused read
    const int st = hs / HP, hp = hs % HP, tap = st / KC, kc = st % KC;
    rd_load_a<GEO>(a[slot], va, tap, kc, hp);
  };
  if (PARTS & 2) {
#pragma unroll
    for (int i = 0; i + 1 < ADEPTH; ++i) load_hs(i, i);
  } else {
    for (int i = 0; i < ADEPTH; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 2; ++q) a[i][j][q] = u32x4{1u + i, 2u + j, 3u + q, 4u};
  }
  MMD_PIN_LOADS();
#pragma unroll
  for (int tap = 0; tap < 5; ++tap)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int st = tap * KC + kc, ri = st % RD;
      const bool zero = st == 0;
#pragma unroll
      for (int hp = 0; hp < HP; ++hp) {
        const int hs = st * HP + hp, cur = hs % ADEPTH;
        if (PARTS & 2) load_hs((hs + ADEPTH - 1) % ADEPTH, hs + ADEPTH - 1 < HS ? hs + ADEPTH - 1 : HS - 1);
        MMD_PIN_LOADS();
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (PARTS & 1) {
#ifdef AGPR_ACC   // the accumulators in the AGPR half of the register file (inline asm: timing experiment only, no hazard padding)
              f32x4& c = acc[2 * hp + sm][t];
              if (zero) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a[cur][sm][1]), "v"(b[ri][t][0]));
              else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a[cur][sm][1]), "v"(b[ri][t][0]));
              asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a[cur][sm][0]), "v"(b[ri][t][1]));
              asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a[cur][sm][0]), "v"(b[ri][t][0]));
#else
              if (zero) vb_three<true>(acc[2 * hp + sm][t], a[cur][sm], b[ri][t]);
              else vb_three<false>(acc[2 * hp + sm][t], a[cur][sm], b[ri][t]);
#endif
            } else {
              const unsigned x = a[cur][sm][0][0] ^ a[cur][sm][1][3] ^ b[ri][t][0][1] ^ b[ri][t][1][2];
              acc[2 * hp + sm][t][0] += __builtin_bit_cast(float, (x & 0x007fffffu) | 0x3f800000u) * 1e-9f;
            }
          }
      }
#ifdef LDSB   // (X1: the kernel's 2 x 4 tile, but the B fragments come from a static LDS stage instead of L2: is it the global loads?)
      if ((PARTS & 4) && st + RD < STEPS) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < 2; ++q) b[ri][t][q] = *reinterpret_cast<const u32x4*>(g_ldsb + (((st + RD) & 1) * 4 + 2 * t + q) * 1024);
      }
#else
      if ((PARTS & 4) && st + RD < STEPS) rd_load_b<GEO, NT>(b[ri], w, st + RD);
#endif
      MMD_PIN_LOADS();
    }
}
#endif   // SPREAD
// -DSKEW=<n>: the SECOND workgroup of a CU (blocks >= 256) starts n x 64 cycles late, so that its MFMA phases meet the first
// one's epilogues (do the two workgroups of a CU drift into lockstep?)
#ifndef SKEW
#define SKEW 0
#endif
// -DUB_RD=<n>: weight ring depth (the kernel: 3); -DUB_NB=256 (main): ONE workgroup per CU, the regime of launches of <= 1024
// trajectories (a single wave per SIMD: nothing fills the gaps of another wave's operand waits)
#ifndef UB_RD
#define UB_RD 3
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_loop(ConvP p, float* out, int nconv) {
#ifdef LDSB
  __shared__ __attribute__((aligned(16))) float lds[G128::BYTES / 4 + 64 + 4 * 8 * 1024 / 4];
#else
  __shared__ __attribute__((aligned(16))) float lds[G128::BYTES / 4 + 64];
#endif
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 15, g = lane >> 4;
  const int c0 = 32 * wave + 2 * n;
  char* const slab = reinterpret_cast<char*>(lds);
  const char* const va = slab + g * G128::G + n * 16;
  char* const vs = slab + wave * G128::G + (n >> 2) * G128::BX + (2 + 4 * g) * 16 + (n & 3) * 4;
  f32x4 acc[4][2];
  for (int s = 0; s < 4; ++s) for (int t = 0; t < 2; ++t) for (int r = 0; r < 4; ++r) acc[s][t][r] = 0.01f * ((threadIdx.x * 7 + s * 3 + t + r + blockIdx.x) % 97) - 0.5f;
  rd_zero_halo<G128>(slab);
  int woff[2] = {(2 * wave) * G128::FRAGS5 * 64 + lane, (2 * wave + 1) * G128::FRAGS5 * 64 + lane};
#ifdef LDSB
  char* const ldsb = slab + G128::BYTES + 256 + wave * 8 * 1024 + lane * 16;      // the wave's 8 static fragments (2 steps x 2 tiles x 2 pieces)
  for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(ldsb + i * 1024) = p.w[(wave * 8 + i) * 64 + lane];
#else
  const char* const ldsb = nullptr;
#endif
  __syncthreads();
  if (SKEW && blockIdx.x >= 256) {
    for (int i = 0; i < SKEW; ++i) __builtin_amdgcn_s_sleep(1);
  }
  for (int k = 0; k < nconv; ++k) {
    // (per-conv pointers, as in the kernel: no address hoisting out of the loop; the barrier sits on the OFFSETS so that the
    // loads stay global_load -- behind an opaque pointer they become flat_load, which also counts on lgkmcnt)
    asm volatile("" : "+v"(woff[0]), "+v"(woff[1]));
    const u32x4* wp[2] = {reinterpret_cast<const u32x4*>(p.w) + woff[0], reinterpret_cast<const u32x4*>(p.w) + woff[1]};
    const Epi<2> e = epi_load<2>(p.par, p.par + 128, p.par + 256, p.par + 384, p.par + 512, c0);
    u32x4 ring[UB_RD][2][2];
    rd_ring_load<G128, 2, UB_RD>(ring, wp);
    if (PARTS & 16) rd_store2<G128>(vs, acc);
    if (PARTS & 32) __syncthreads();
    rd_taps_parts<G128, 2, 4, UB_RD>(acc, va, wp, ring, ldsb);
    if (PARTS & 8) epilogue(acc, e);
    if (PARTS & 32) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int t = 0; t < 2; ++t) s += acc[i][t][0] + acc[i][t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
constexpr int SAMPLES_PER_WG = 4;
#elif defined(M32)
// downs.2's 128 -> 128 conv on v_mfma_f32_32x32x16_f16 (rm_taps / rm_gn_mish / rm_store below; M tile = sample pair, N tile =
// the wave's 32 channels, K step = 16 channels): the same loop as `base`, with parts switched off by -DM32=<mask> like -DPARTS
// (1 MFMAs, 2 A fragments from the LDS, 4 weight fragments, 8 epilogue, 16 slab store, 32 barriers; 63 = everything).
// Measured slower than the 16x16x32 form in every arm (profiles/r04_ubench_conv_loop.txt): these building blocks live here, not
// in the product kernel.
// ----------------------------------------------------------------------------------------------------------------
// The L = 16 stages on v_mfma_f32_32x32x16_f16 (8 passes: the same FLOP rate as 16x16x32, HALF the A / B operand reads from the
// register file per FLOP and half the MFMA instructions).  An M tile is a sample PAIR: matrix row r < 16 = position r of sample
// 2 mt, row r >= 16 = position (r - 20) mod 16 of sample 2 mt + 1 -- rotated by four rows, because the second sample's slab rows
// start RPS = 20 rows = 320 B after the first's and a ds_read_b128 is served in groups of 16 lanes that mix the two 16-lane rows
// (rows {0-3, 12-15} with {20-27}: with the rotation the 16 lanes of every group hit 16 different 16-byte bank groups of the same
// Rd slab the 16x16 forms use; unrotated the second sample's rows collide 2-way with the first's).  An N tile is the wave's 32
// channels (column n = channel 32 wave + n), a K step 16 channels: lane (row r = lane & 31, h = lane >> 5) reads the 8 channels of
// block 2 ks + h.  C/D layout: lane (n, h) holds for channel n the matrix rows 8 (i >> 2) + 4 h + (i & 3), i = 0 .. 15, i.e. of
// the even sample positions 4 h + (i & 3) (i < 4) and 8 + 4 h + (i & 3) (4 <= i < 8), of the odd sample (i >= 8) positions
// (12 - 12 h) + (i & 3) and 4 + 4 h + (i & 3): both samples own the same two position sets {0-3, 8-11} / {4-7, 12-15}, one per
// wave half, so every per-sample reduction below (four-value chains per position set, combined by commutative additions) is the
// same arithmetic for an even and an odd sample -- results stay bitwise independent of the batch position.
// Weights: per n-tile of 32 columns [tap][K step ks][piece][lane] x 16 B (pack_rm).
// ----------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma_w(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <bool ZERO>
__device__ __forceinline__ void vw_three(f32x16& x, const u32x4 (&a)[2], const u32x4 (&b)[2]) {
  f32x16 c;
  if constexpr (ZERO) {
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
  } else {
    c = x;
  }
  c = mfma_w(a[1], b[0], c);
  c = mfma_w(a[0], b[1], c);
  c = mfma_w(a[0], b[0], c);
  x = c;
  __builtin_amdgcn_sched_barrier(0);
}
template <class GEO> struct RmGeo : GEO {
  static constexpr int KS = GEO::KC * 2;                     // K steps of 16 channels
  static constexpr int FRAGS5 = 5 * KS * 2;                  // weight fragments per 32-column n-tile of a k = 5 conv
  // slab offset of channel block cb = 2 ks + h (the two blocks of a K step are neighbours inside one lane group's region)
  static constexpr int blk(int cb) { return (cb / GEO::KC) * GEO::G + (cb % GEO::KC) * GEO::BX; }
};
// the lane's A offset inside the slab: (row of its matrix row, block half h); M tile mt adds 2 mt RPS rows, K step ks blk(2 ks)
template <class GEO>
__device__ __forceinline__ int rm_lane_a(int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int row = r < 16 ? r : GEO::RPS + ((r - 20) & 15);
  return row * 16 + h * GEO::BX;
}
template <class GEO, int MT>
__device__ __forceinline__ void rm_load_a(u32x4 (&a)[MT][2], const char* va, int rowoff, int ks) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      a[mt][q] = *reinterpret_cast<const u32x4*>(va + q * GEO::PS + RmGeo<GEO>::blk(2 * ks) + (2 * mt * GEO::RPS + rowoff) * 16);
}
__device__ __forceinline__ void rm_load_b(u32x4 (&b)[2], const u32x4* w, int step) {
#pragma unroll
  for (int q = 0; q < 2; ++q) b[q] = w[(step * 2 + q) * 64];
}
template <int RD>
__device__ __forceinline__ void rm_ring_load(u32x4 (&b)[RD][2], const u32x4* w) {
#pragma unroll
  for (int i = 0; i < RD; ++i) rm_load_b(b[i], w, i);
  MMD_PIN_LOADS();
}
// acc[mt] (+)= conv over TAPS taps (slab rows TAP0 .. relative to the output position) x the C channels of the slab; va = slab +
// rm_lane_a; w = the wave's n-tile pack + lane; b = ring pre-loaded with the first RD steps.  RES: the stage's 1x1 residual conv
// rides on the centre tap's A fragments (wr = [K step][piece] + lane).  FRESH: start from zero.
template <class GEO, int TAP0, int TAPS, bool FRESH, bool RES, int MT, int RD>
__device__ __forceinline__ void rm_taps(f32x16 (&acc)[MT], f32x16 (&res)[MT], const char* va, const u32x4* w, const u32x4* wr,
                                        u32x4 (&b)[RD][2]) {
  constexpr int KS = RmGeo<GEO>::KS, STEPS = TAPS * KS;
  u32x4 a[2][MT][2];                                         // double-buffered by step
  rm_load_a<GEO, MT>(a[0], va, TAP0, 0);
  constexpr int C0 = (2 - TAP0) * KS;                        // the centre tap's first step
  constexpr int RES_LOOK = C0 < 6 ? C0 : 6;
  u32x4 brp[RES ? KS : 1][2];
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int st = tap * KS + ks, ri = st % RD, cur = st & 1;
      const bool zero = FRESH && st == 0, last_ks = ks + 1 == KS, with_res = RES && TAP0 + tap == 2;
      if constexpr (RES) {
        if (st + RES_LOOK >= C0 && st + RES_LOOK < C0 + KS) rm_load_b(brp[st + RES_LOOK - C0], wr, st + RES_LOOK - C0);
      }
      // the next step's A fragments (past the last step: a valid, unused read)
      rm_load_a<GEO, MT>(a[cur ^ 1], va, last_ks ? TAP0 + tap + 1 : TAP0 + tap, last_ks ? 0 : ks + 1);
      MMD_PIN_LOADS();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (zero) vw_three<true>(acc[mt], a[cur][mt], b[ri]);
        else vw_three<false>(acc[mt], a[cur][mt], b[ri]);
      }
      if constexpr (RES) {
        if (with_res) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (FRESH && ks == 0) vw_three<true>(res[mt], a[cur][mt], brp[ks]);
            else vw_three<false>(res[mt], a[cur][mt], brp[ks]);
          }
        }
      }
      if (st + RD < STEPS) rm_load_b(b[ri], w, st + RD);
      MMD_PIN_LOADS();
    }
}
// sum over a lane's eight values of sample sl (0: even, 1: odd) of an M tile, f(value, running sum) folded over the four
// registers of each position set separately and the two chains added: an odd sample's lanes hold the position sets of the even
// sample's OTHER wave half (in the other register order), and a + b = b + a, so both samples get the same arithmetic.
template <class F>
__device__ __forceinline__ float rm_sum8(const f32x16& t, int sl, F f) {
  float c0 = 0.f, c1 = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    c0 = f(t[8 * sl + r], c0);
    c1 = f(t[8 * sl + 4 + r], c1);
  }
  return c0 + c1;
}
__device__ __forceinline__ float row_sum16(float v) {        // sum over the 16 lanes of a DPP row, in all of them
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  return v;
}
// GroupNorm + Mish of the 32x32 tiles acc[mt] (raw f16x2 conv output: true value = acc * isc * inv[sample]) + add(mt, i); a
// group = 16 adjacent channels (one DPP row) x the sample's 16 positions (both wave halves), NG = 256 values; sample = 2 mt + (i >> 3)
template <int MT, bool ACT, class ADD>
__device__ __forceinline__ void rm_gn_mish(f32x16 (&acc)[MT], float bias, float gamma, float beta, float isc,
                                           const float (&inv)[2 * MT], const ActScale& as, ADD add) {
  constexpr float inv_n = 1.f / 256.f;
  const float bmean = row_sum16(bias) * 16.f * inv_n;
  float k[2 * MT], sum[2 * MT], dm[2 * MT], sq[2 * MT];
#pragma unroll
  for (int s = 0; s < 2 * MT; ++s) {
    k[s] = isc * inv[s];
    sum[s] = row_sum16(rm_sum8(acc[s >> 1], s & 1, [](float x, float c) { return c + x; }) * k[s]);
  }
#pragma unroll
  for (int s = 0; s < 2 * MT; ++s) sum[s] = add_xor32(sum[s]);
#pragma unroll
  for (int s = 0; s < 2 * MT; ++s) {
    const float mean = fmaf(sum[s], inv_n, bmean);
    dm[s] = mean - bias;
    const float ks = k[s], d0 = dm[s];
    sq[s] = row_sum16(rm_sum8(acc[s >> 1], s & 1, [&](float x, float c) {
      const float d = fmaf(x, ks, -d0);
      return fmaf(d, d, c);
    }));
  }
#pragma unroll
  for (int s = 0; s < 2 * MT; ++s) sq[s] = add_xor32(sq[s]);
#pragma unroll
  for (int s = 0; s < 2 * MT; ++s) {
    const float rstd = __builtin_amdgcn_rsqf(fmaf(sq[s], inv_n, 1e-5f));
    GnCoef cf = gn_coef(dm[s], rstd, gamma, beta);
    cf.sa *= k[s];
    f32x16& t = acc[s >> 1];
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
      const int i = 8 * (s & 1) + r;
      const f32x2_t o = gn_mish2<ACT>(f32x2_t{t[i], t[i + 1]}, cf, f32x2_t{add(s >> 1, i), add(s >> 1, i + 1)}, as);
      t[i] = o.x;
      t[i + 1] = o.y;
    }
  }
}
// per-sample |x| maxima of the tiles -> mx region 0 (and region2 if > 0): slots 2 wave + {0, 1} of MX_SLOTS = 8 (the wave's two
// halves; the two 16-lane rows of a half are combined first)
template <int MT>
__device__ __forceinline__ void rm_dyn_out(const f32x16 (&acc)[MT], float* mx, int wave, int lane, int region2) {
#pragma unroll
  for (int s = 0; s < 2 * MT; ++s) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) m = fmaxf(m, fabsf(acc[s >> 1][8 * (s & 1) + r]));
    m = row_max16(m);
    m = max_xor16(m);
    if ((lane & 31) == 0) {
      mx[s * MX_SLOTS + 2 * wave + (lane >> 5)] = m;
      if (region2) mx[region2 * MX_REGION + s * MX_SLOTS + 2 * wave + (lane >> 5)] = m;
    }
  }
}
// the lane's two store offsets inside the slab for channel c = 32 wave + n of a C-channel Rd slab: vs[0] = even sample, rows 2 + 4 h
// + r (registers r = 0 .. 3; registers 4 .. 7: eight rows further); vs[1] = odd sample, registers 8 .. 11 at rows 2 + (12 - 12 h) + r
// (registers 12 .. 15 at the even sample's row offset + RPS + 4 rows)
template <class GEO>
__device__ __forceinline__ void rm_lane_s(int (&vs)[2], int wave, int lane) {
  const int n = lane & 31, h = lane >> 5, c = 32 * wave + n, cb = c >> 3;
  const int base = (cb / GEO::KC) * GEO::G + (cb % GEO::KC) * GEO::BX + (c & 7) * 2;
  vs[0] = base + (2 + 4 * h) * 16;
  vs[1] = base + (GEO::RPS + 2 + 12 - 12 * h) * 16;
}
// 32x32 tiles -> the slab, one fp16 per lane and value (a lane owns ONE channel: the two pieces of a pair of its values are split
// together and stored as halves, ds_write_b16 / ds_write_b16_d16_hi)
template <class GEO, int MT>
__device__ __forceinline__ void rm_store(char* slab, const int (&vs)[2], const f32x16 (&acc)[MT]) {
  auto st2 = [&](char* p0, char* p1, float v0, float v1) {   // v0 -> p0, v1 -> p1 (both pieces)
    const F16Pair f = f16_split2(v0, v1);
    *reinterpret_cast<unsigned short*>(p0) = (unsigned short)f.hi;
    *reinterpret_cast<unsigned short*>(p1) = (unsigned short)(f.hi >> 16);
    *reinterpret_cast<unsigned short*>(p0 + GEO::PS) = (unsigned short)f.lo;
    *reinterpret_cast<unsigned short*>(p1 + GEO::PS) = (unsigned short)(f.lo >> 16);
  };
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    char* const e = slab + vs[0] + 2 * mt * GEO::RPS * 16;
    char* const o = slab + vs[1] + 2 * mt * GEO::RPS * 16;
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      st2(e + r * 16, e + (r + 1) * 16, acc[mt][r], acc[mt][r + 1]);
      st2(e + (8 + r) * 16, e + (9 + r) * 16, acc[mt][4 + r], acc[mt][5 + r]);
      st2(o + r * 16, o + (r + 1) * 16, acc[mt][8 + r], acc[mt][9 + r]);
      st2(e + (GEO::RPS + 4 + r) * 16, e + (GEO::RPS + 5 + r) * 16, acc[mt][12 + r], acc[mt][13 + r]);
    }
  }
}


template <class GEO, int MT, int RD>
__device__ __forceinline__ void rm_taps_parts(f32x16 (&acc)[MT], const char* va, const u32x4* w, u32x4 (&b)[RD][2]) {
  constexpr int KS = RmGeo<GEO>::KS, STEPS = 5 * KS;
  u32x4 a[2][MT][2];
  if (M32 & 2) rm_load_a<GEO, MT>(a[0], va, 0, 0);
  else for (int i = 0; i < 2; ++i) for (int j = 0; j < MT; ++j) for (int q = 0; q < 2; ++q) a[i][j][q] = u32x4{1u + i, 2u + j, 3u + q, 4u};   // (distinct per M tile: no CSE of the chains)
  MMD_PIN_LOADS();
#pragma unroll
  for (int tap = 0; tap < 5; ++tap)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int st = tap * KS + ks, ri = st % RD, cur = st & 1;
      const bool zero = st == 0, last_ks = ks + 1 == KS;
      if (M32 & 2) rm_load_a<GEO, MT>(a[cur ^ 1], va, last_ks ? tap + 1 : tap, last_ks ? 0 : ks + 1);
      MMD_PIN_LOADS();
#ifdef M32IL    // the two M tiles' chains interleaved: no MFMA depends on the one right before it
      if (M32 & 1) {
        static_assert(MT == 2, "interleave of two accumulators");
        f32x16 c0 = acc[0], c1 = acc[1];
        if (zero) { for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; } }
        c0 = mfma_w(a[cur][0][1], b[ri][0], c0);
        c1 = mfma_w(a[cur][1][1], b[ri][0], c1);
        c0 = mfma_w(a[cur][0][0], b[ri][1], c0);
        c1 = mfma_w(a[cur][1][0], b[ri][1], c1);
        c0 = mfma_w(a[cur][0][0], b[ri][0], c0);
        c1 = mfma_w(a[cur][1][0], b[ri][0], c1);
        acc[0] = c0; acc[1] = c1;
        __builtin_amdgcn_sched_barrier(0);
      }
#else
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (M32 & 1) {
          if (zero) vw_three<true>(acc[mt], a[cur][mt], b[ri]);
          else vw_three<false>(acc[mt], a[cur][mt], b[ri]);
        } else {
          const unsigned x = a[cur][mt][0][0] ^ a[cur][mt][1][3] ^ b[ri][0][1] ^ b[ri][1][2];
          acc[mt][0] += __builtin_bit_cast(float, (x & 0x007fffffu) | 0x3f800000u) * 1e-9f;
        }
      }
#endif
      if ((M32 & 4) && st + RD < STEPS) rm_load_b(b[ri], w, st + RD);
      MMD_PIN_LOADS();
    }
}
#ifndef M32_RD
#define M32_RD 6
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_loop(ConvP p, float* out, int nconv) {
  __shared__ __attribute__((aligned(16))) float lds[G128::BYTES / 4 + 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = 32 * wave + (lane & 31);
  char* const slab = reinterpret_cast<char*>(lds);
  const char* const va = slab + rm_lane_a<G128>(lane);
  int vs[2];
  rm_lane_s<G128>(vs, wave, lane);
  f32x16 acc[2];
  for (int m = 0; m < 2; ++m) for (int i = 0; i < 16; ++i) acc[m][i] = 0.01f * ((threadIdx.x * 7 + m * 3 + i + blockIdx.x) % 97) - 0.5f;
  rd_zero_halo<G128>(slab);
  int woff = wave * RmGeo<G128>::FRAGS5 * 64 + lane;
  __syncthreads();
  const float one4[4] = {1.f, 1.f, 1.f, 1.f};
  for (int k = 0; k < nconv; ++k) {
    asm volatile("" : "+v"(woff));
    const u32x4* wp = reinterpret_cast<const u32x4*>(p.w) + woff;
    const Epi<1> e = epi_load<1>(p.par, p.par + 128, p.par + 256, p.par + 384, p.par + 512, c);
    u32x4 ring[M32_RD][2];
    rm_ring_load<M32_RD>(ring, wp);
    if (M32 & 16) rm_store<G128, 2>(slab, vs, acc);
    if (M32 & 32) __syncthreads();
    rm_taps_parts<G128, 2, M32_RD>(acc, va, wp, ring);
    if (M32 & 8) {
      const float tb = e.tb[0];
      rm_gn_mish<2, true>(acc, e.b[0], e.g[0], e.be[0], e.is[0], one4, act_scale(1.f), [&](int, int) { return tb; });
    }
    if (M32 & 32) __syncthreads();
  }
  float s = 0.f;
  for (int m = 0; m < 2; ++m) s += acc[m][0] + acc[m][3] + acc[m][9] + acc[m][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
constexpr int SAMPLES_PER_WG = 4;
#elif defined(W8)
// VERDICT r4 #1 / DESIGN 7.1a, the lever nobody had run: ONE 8-wave workgroup of 8 samples per CU instead of two 4-wave
// workgroups of 4, a wave = (n-tile quad nq = wave & 1: channels 64 nq .. 64 nq + 63, sample pair sp = wave >> 1) with a register
// tile of 4 n-tiles x 2 samples (the kernel: 2 x 4) -- half the A-fragment LDS reads per MFMA -- and the conv's weights staged
// ONCE per workgroup into an LDS ring by LDS-DMA (global_load_lds_dwordx4), read from there by the four waves that share them.
// -DW8=<mask>: 1 MFMAs, 2 A fragments from the LDS, 4 B fragments from the LDS stage, 8 epilogue, 16 slab store, 32 the conv's
// two barriers, 64 the stage ring is really refilled (LDS-DMA + one barrier per W8_SB steps); without 64 the ring holds static
// data: the loop's best case.  -DW8_SB=<1|2>: steps per refill barrier (ring = 2 W8_SB steps of 16 KB).
// -DW8_RING=<n> (with bit 64): the stage is a ring of n single steps instead of two halves; the refill of step st + n - 1 is issued at
// the top of step st (behind the step's barrier: every wave is past step st - 1, whose slot it takes) and a wave waits only for the
// DMAs of the step it is about to read -- s_waitcnt vmcnt(2 (n - 2)): two fragments per wave and step stay in flight per younger
// step -- so a fragment has n - 2 steps (~0.4 us each) to arrive instead of one refill group.
#ifndef W8_SB
#define W8_SB 2
#endif
#ifdef W8_RING
#undef W8_SB
#define W8_SB 1
#endif
struct Rd8 {                     // RdGeo<128> for EIGHT samples: [piece][lane group][chunk][row = 20 s + 2 + position][8 ch]
  static constexpr int KC = 4, RPS = 20, BX = 8 * RPS * 16 + 32, G = (KC * BX + 255) / 256 * 256, PS = 4 * G, BYTES = 2 * PS;
  static constexpr int tile_row(int m) { return m * RPS; }
};
constexpr int W8_STEP_BYTES = 8 * 2 * 1024;                    // one (tap, chunk) step of all 8 n-tiles, two pieces
#ifdef W8_RING
constexpr int W8_SLOTS = W8_RING;
#else
constexpr int W8_SLOTS = 2 * W8_SB;
#endif
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_loop(ConvP p, float* out, int nconv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 15, g = lane >> 4;
  const int nq = wave & 1, sp = wave >> 1;
  char* const slab = reinterpret_cast<char*>(lds);
  char* const stage = slab + Rd8::BYTES;
  const char* const va = slab + g * Rd8::G + n * 16;
  f32x4 acc[2][4];
  for (int sm = 0; sm < 2; ++sm) for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) acc[sm][t][r] = 0.01f * ((threadIdx.x * 7 + sm * 3 + t + r + blockIdx.x) % 97) - 0.5f;
  for (int i = threadIdx.x; i < Rd8::BYTES / 16; i += 512) reinterpret_cast<uint4*>(slab)[i] = make_uint4(0u, 0u, 0u, 0u);
  // the stage: static pseudo-random fp16 content (all W8_SLOTS steps) unless it is refilled
  for (int i = threadIdx.x; i < W8_SLOTS * W8_STEP_BYTES / 16; i += 512) reinterpret_cast<uint4*>(stage)[i] = p.w[i];
  const int c0[2] = {64 * nq + 2 * n, 64 * nq + 32 + 2 * n};
  __syncthreads();
  // DMA of the fragments of steps [s0, s0 + W8_SB) into ring half `half`: 16 W8_SB fragments of 1 KiB dealt to the 8 waves
  auto refill = [&](int s0, int half) {
#pragma unroll
    for (int i = 0; i < 2 * W8_SB; ++i) {
      const int f = wave + 8 * i;                             // fragment within the W8_SB steps
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.w + ((size_t)s0 * 16 + f) * 64 + lane),
                                       (__attribute__((address_space(3))) void*)(stage + (half * W8_SB * 16 + f) * 1024), 16, 0, 0);
    }
  };
  for (int k = 0; k < nconv; ++k) {
    const Epi<2> e0 = epi_load<2>(p.par, p.par + 128, p.par + 256, p.par + 384, p.par + 512, c0[0]);
    const Epi<2> e1 = epi_load<2>(p.par, p.par + 128, p.par + 256, p.par + 384, p.par + 512, c0[1]);
#ifdef W8_RING
    auto refill1 = [&](int st) {                               // the 16 fragments of step st -> slot st % W8_RING
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int f = wave + 8 * i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.w + ((size_t)st * 16 + f) * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(stage + ((st % W8_RING) * 16 + f) * 1024), 16, 0, 0);
      }
    };
    if (W8 & 64) {
#pragma unroll
      for (int st = 0; st < W8_RING - 1; ++st) refill1(st);
    }
#else
    if (W8 & 64) refill(0, 0);
#endif
    if (W8 & 16) {
      // the lane's channel pairs: tile pair tp of the quad = block 4 (2 nq + tp) + (n >> 2), rows of samples 2 sp, 2 sp + 1
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        char* const vs = slab + (2 * nq + tp) * Rd8::G + (n >> 2) * Rd8::BX + (2 * sp * Rd8::RPS + 2 + 4 * g) * 16 + (n & 3) * 4;
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const F16Pair f = f16_split2(acc[sm][2 * tp][r], acc[sm][2 * tp + 1][r]);
            *reinterpret_cast<unsigned*>(vs + (sm * Rd8::RPS + r) * 16) = f.hi;
            *reinterpret_cast<unsigned*>(vs + Rd8::PS + (sm * Rd8::RPS + r) * 16) = f.lo;
          }
      }
    }
#ifdef W8_RING
    if (W8 & 64) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (W8_RING - 2)) : "memory");
#else
    if (W8 & 64) staged_weights_landed();
#endif
    if (W8 & 32) __syncthreads();
    // ---- taps: 20 steps of 2 samples x 4 n-tiles x 3 MFMAs; A double-buffered by step from the slab, B by step from the stage
    u32x4 a[2][2][2], b[2][4][2];
    auto load_a = [&](int buf, int st) {
      if (W8 & 2) rd_load_a<Rd8, 2>(a[buf], va, st / 4, st % 4, sp);
    };
    auto load_b = [&](int buf, int st) {
#ifdef W8_GLOBB   // (X2: the 4 x 2 tile with its B fragments straight from L2, eight 64-lane global loads per step and wave)
      if (W8 & 4) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int q = 0; q < 2; ++q) b[buf][t][q] = reinterpret_cast<const u32x4*>(p.w)[(((size_t)st * 8 + 4 * nq + t) * 2 + q) * 64 + lane];
        return;
      }
#endif
      if (W8 & 4) {
        const char* src = stage + (st % W8_SLOTS) * W8_STEP_BYTES + (4 * nq) * 2048 + lane * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int q = 0; q < 2; ++q) b[buf][t][q] = *reinterpret_cast<const u32x4*>(src + (2 * t + q) * 1024);
      }
    };
    if (!(W8 & 2))
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 2; ++q) a[i][j][q] = u32x4{1u + i, 2u + j, 3u + q, 4u};
    if (!(W8 & 4))
      for (int i = 0; i < 2; ++i) for (int t = 0; t < 4; ++t) for (int q = 0; q < 2; ++q) b[i][t][q] = u32x4{5u + i, 6u + t, 7u + q, 8u};
    load_a(0, 0);
    load_b(0, 0);
    MMD_PIN_LOADS();
#pragma unroll
    for (int st = 0; st < 20; ++st) {
      const int cur = st & 1;
#ifdef W8_RING
      if (W8 & 64) {
        if (st > 0) {
          // step st's fragments (issued W8_RING - 1 steps ago) have landed: at most the W8_RING - 2 younger steps' stay in flight
          if (st + W8_RING - 2 < 20) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (W8_RING - 2)) : "memory");
          else staged_weights_landed();
          __syncthreads();
        }
        if (st + W8_RING - 1 < 20) refill1(st + W8_RING - 1);
      }
      if (false) {
#else
      if ((W8 & 64) && st % W8_SB == 0) {
#endif
        // top of a refill group: this wave's DMA of the group has landed, every wave has landed its own and is past the previous
        // group (barrier): the other ring half is free for the group after this one
        if (st > 0) {
          staged_weights_landed();
          __syncthreads();
        }
        if (st + W8_SB < 20) refill(st + W8_SB, ((st / W8_SB) + 1) & 1);
      }
      load_a(cur ^ 1, st + 1 < 20 ? st + 1 : 19);
      load_b(cur ^ 1, st + 1 < 20 ? st + 1 : 19);
      MMD_PIN_LOADS();
#pragma unroll
      for (int sm = 0; sm < 2; ++sm)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (W8 & 1) {
            if (st == 0) vb_three<true>(acc[sm][t], a[cur][sm], b[cur][t]);
            else vb_three<false>(acc[sm][t], a[cur][sm], b[cur][t]);
          } else {
            const unsigned x = a[cur][sm][0][0] ^ a[cur][sm][1][3] ^ b[cur][t][0][1] ^ b[cur][t][1][2];
            acc[sm][t][0] += __builtin_bit_cast(float, (x & 0x007fffffu) | 0x3f800000u) * 1e-9f;
          }
        }
    }
    if (W8 & 8) {
      const float one2[2] = {1.f, 1.f};
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        const Epi<2>& e = tp ? e1 : e0;
        f32x4 tile[2][2] = {{acc[0][2 * tp], acc[0][2 * tp + 1]}, {acc[1][2 * tp], acc[1][2 * tp + 1]}};
        const float t0 = e.tb[0], t1 = e.tb[1];
        rd_gn_mish<2, 256, true>(tile, e.b, e.g, e.be, e.is, one2, act_scale(1.f), [&](int, int t, int) { return t ? t1 : t0; });
        acc[0][2 * tp] = tile[0][0]; acc[0][2 * tp + 1] = tile[0][1]; acc[1][2 * tp] = tile[1][0]; acc[1][2 * tp + 1] = tile[1][1];
      }
    }
    if (W8 & 32) __syncthreads();
  }
  float s = 0.f;
  for (int sm = 0; sm < 2; ++sm) for (int t = 0; t < 4; ++t) s += acc[sm][t][0] + acc[sm][t][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
constexpr int SAMPLES_PER_WG = 8;
#define W8_LAUNCH 1
#elif !defined(FAT)
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
  int woff[2] = {(2 * wave) * G128::FRAGS5 * 64 + lane, (2 * wave + 1) * G128::FRAGS5 * 64 + lane};
  __syncthreads();
  for (int k = 0; k < nconv; ++k) {
    // (per-conv pointers, as in the kernel: no address hoisting out of the loop; the barrier sits on the OFFSETS so that the
    // loads stay global_load -- behind an opaque pointer they become flat_load, which also counts on lgkmcnt)
    asm volatile("" : "+v"(woff[0]), "+v"(woff[1]));
    const u32x4* wp[2] = {reinterpret_cast<const u32x4*>(p.w) + woff[0], reinterpret_cast<const u32x4*>(p.w) + woff[1]};
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
#elif defined(MICRO)
// compile-time loops: every index below is a constant expression, so the epilogue state stays in registers
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// rd_taps with a hook after EVERY MFMA triple (8 per step, 160 per conv), strict program order (sched_barrier after each):
// hook(m) issues a <= 4-instruction micro-slice of the OTHER half's epilogue into the triple's shadow
template <class GEO, int NT, int MT, int RD, class HOOKF>
__device__ __forceinline__ void rd_taps_micro(f32x4 (&acc)[MT][NT], const char* va, const u32x4* const (&w)[NT], u32x4 (&b)[RD][NT][2], HOOKF hook) {
  constexpr int KC = GEO::KC, STEPS = 5 * KC, HP = MT / 2;
  u32x4 a[2][2][2];
  rd_load_a<GEO>(a[0], va, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  static_for<0, STEPS>([&](auto stc) {
    constexpr int st = decltype(stc)::value, tap = st / KC, kc = st % KC, ri = st % RD;
    constexpr bool zero = st == 0, last_kc = kc + 1 == KC;
    static_for<0, HP>([&](auto hpc) {
      constexpr int hp = decltype(hpc)::value, cur = (st * HP + hp) & 1;
      if constexpr (hp + 1 < HP) rd_load_a<GEO>(a[cur ^ 1], va, tap, kc, hp + 1);
      else rd_load_a<GEO>(a[cur ^ 1], va, last_kc ? tap + 1 : tap, last_kc ? 0 : kc + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 2 * NT>([&](auto qc) {
        constexpr int q = decltype(qc)::value, sm = q / NT, t = q % NT;
        vb_three<zero>(acc[2 * hp + sm][t], a[cur][sm], b[ri][t]);
        __builtin_amdgcn_sched_barrier(0);
        hook(std::integral_constant<int, ((st * HP + hp) * 2 + sm) * NT + t>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    if constexpr (st + RD < STEPS) rd_load_b<GEO, NT>(b[ri], w, st + RD);
    __builtin_amdgcn_sched_barrier(0);
  });
}
struct MicroState {
  float k[4][2], sum[4], dm[4][2], sq[4], bmean, v, rstd;
  GnCoef cf[2];
  f32x2_t yl, e2, n2, q2;
};
// micro-slice M (0 .. 159) of rd_gn_mish<2, 256, true> + rd_store2 for the tile in acc; 140 are used
template <int M>
__device__ __forceinline__ void epi_micro(f32x4 (&acc)[4][2], const Epi<2>& e, MicroState& S, char* vs) {
  constexpr float inv_n = 1.f / 256.f, LOG2E = 1.44269504088896341f;
  if constexpr (M < 8) {                         // A: sums, one tile per slice
    constexpr int sm = M >> 1, t = M & 1;
    if constexpr (M == 0) S.bmean = e.b[0] + e.b[1];
    S.k[sm][t] = e.is[t];
    const float st = (acc[sm][t][0] + acc[sm][t][1]) + (acc[sm][t][2] + acc[sm][t][3]);
    S.v = fmaf(st, S.k[sm][t], t ? S.v : 0.f);
    if constexpr (t == 1) S.sum[sm] = S.v;
  } else if constexpr (M < 16) {                 // B: reductions of the sums
    constexpr int sm = (M - 8) >> 1, h = (M - 8) & 1;
    if constexpr (h == 0) S.sum[sm] = group_colsum<8>(S.sum[sm]);
    else S.sum[sm] = add_xor32(add_xor16(S.sum[sm]));
    if constexpr (M == 15) S.bmean = group_colsum<8>(S.bmean) * 16.f * inv_n;
  } else if constexpr (M < 36) {                 // C: centred squares, 5 slices per sample
    constexpr int sm = (M - 16) / 5, j = (M - 16) % 5;
    if constexpr (j == 0) {
      const float mean = fmaf(S.sum[sm], inv_n, S.bmean);
      S.dm[sm][0] = mean - e.b[0];
      S.dm[sm][1] = mean - e.b[1];
      S.v = 0.f;
    } else {
      constexpr int t = (j - 1) >> 1, r0 = ((j - 1) & 1) * 2;
#pragma unroll
      for (int r = r0; r < r0 + 2; ++r) {
        const float d = fmaf(acc[sm][t][r], S.k[sm][t], -S.dm[sm][t]);
        S.v = fmaf(d, d, S.v);
      }
      if constexpr (j == 4) S.sq[sm] = S.v;
    }
  } else if constexpr (M < 44) {                 // D: reductions of the squares
    constexpr int sm = (M - 36) >> 1, h = (M - 36) & 1;
    if constexpr (h == 0) S.sq[sm] = group_colsum<8>(S.sq[sm]);
    else S.sq[sm] = add_xor32(add_xor16(S.sq[sm]));
  } else if constexpr (M < 124) {                // E + F per sample: rstd, 2 x coef, then 4 pairs x 3 slices
    constexpr int sm = (M - 44) / 20, j = (M - 44) % 20;
    if constexpr (j == 0) S.rstd = __builtin_amdgcn_rsqf(fmaf(S.sq[sm], inv_n, 1e-5f));
    else if constexpr (j < 3) {
      constexpr int t = j - 1;
      S.cf[t] = gn_coef(S.dm[sm][t], S.rstd, e.g[t], e.be[t]);
      S.cf[t].sa *= S.k[sm][t];
    } else if constexpr (j < 15) {
      constexpr int pr = (j - 3) / 3, ph = (j - 3) % 3, t = pr >> 1, r = (pr & 1) * 2;
      if constexpr (ph == 0) {
        S.yl = __builtin_elementwise_fma(f32x2_t{acc[sm][t][r], acc[sm][t][r + 1]}, f32x2_t{S.cf[t].sa, S.cf[t].sa}, f32x2_t{S.cf[t].sb, S.cf[t].sb});
        S.e2 = f32x2_t{__builtin_amdgcn_exp2f(fminf(S.yl.x, 20.f * LOG2E)), __builtin_amdgcn_exp2f(fminf(S.yl.y, 20.f * LOG2E))};
      } else if constexpr (ph == 1) {
        S.n2 = S.e2 * (S.e2 + f32x2_t{2.f, 2.f});
        const f32x2_t den = __builtin_elementwise_fma(S.n2, f32x2_t{LOG2E, LOG2E}, f32x2_t{2.f * LOG2E, 2.f * LOG2E});
        S.q2 = f32x2_t{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
      } else {
        const f32x2_t o = __builtin_elementwise_fma(S.yl, S.n2 * S.q2, f32x2_t{e.tb[t], e.tb[t]});
        acc[sm][t][r] = o.x;
        acc[sm][t][r + 1] = o.y;
      }
    }
  } else if constexpr (M < 140) {                // G: split + store, one (sample, r) per slice
    constexpr int sm = (M - 124) >> 2, r = (M - 124) & 3;
    const F16Pair f = f16_split2(acc[sm][0][r], acc[sm][1][r]);
    *reinterpret_cast<unsigned*>(vs + (sm * G128::RPS + r) * 16) = f.hi;
    *reinterpret_cast<unsigned*>(vs + G128::PS + (sm * G128::RPS + r) * 16) = f.lo;
  }
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_loop(ConvP p, float* out, int nconv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
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
  int woff[2] = {(2 * wave) * G128::FRAGS5 * 64 + lane, (2 * wave + 1) * G128::FRAGS5 * 64 + lane};
  __syncthreads();
  rd_store2<G128>(slabA + vs_off, accA);
  __syncthreads();
  MicroState S;
  for (int k = 0; k < nconv; ++k) {
    // (per-conv pointers, as in the kernel: no address hoisting out of the loop; the barrier sits on the OFFSETS so that the
    // loads stay global_load -- behind an opaque pointer they become flat_load, which also counts on lgkmcnt)
    asm volatile("" : "+v"(woff[0]), "+v"(woff[1]));
    const u32x4* wp[2] = {reinterpret_cast<const u32x4*>(p.w) + woff[0], reinterpret_cast<const u32x4*>(p.w) + woff[1]};
    const Epi<2> e = epi_load<2>(p.par, p.par + 128, p.par + 256, p.par + 384, p.par + 512, c0);
    {
      u32x4 ring[3][2][2];
      rd_ring_load<G128, 2, 3>(ring, wp);
      rd_taps_micro<G128, 2, 4, 3>(accA, slabA + va_off, wp, ring, [&](auto mc) { epi_micro<decltype(mc)::value>(accB, e, S, slabB + vs_off); });
      __syncthreads();
    }
    {
      u32x4 ring[3][2][2];
      rd_ring_load<G128, 2, 3>(ring, wp);
      rd_taps_micro<G128, 2, 4, 3>(accB, slabB + va_off, wp, ring, [&](auto mc) { epi_micro<decltype(mc)::value>(accA, e, S, slabA + vs_off); });
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int t = 0; t < 2; ++t) s += accA[i][t][0] + accA[i][t][3] + accB[i][t][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
constexpr int SAMPLES_PER_WG = 8;
#elif defined(HOOK)
// rd_taps with a per-step hook: step st's code, then hook(st) -- an epilogue slice of the OTHER half -- then a small
// sched_group_barrier pipeline for this region (8 MFMA triples, a few VALU after each) and a region barrier
template <class GEO, int NT, int MT, int RD, class HOOKF>
__device__ __forceinline__ void rd_taps_hook(f32x4 (&acc)[MT][NT], const char* va, const u32x4* const (&w)[NT], u32x4 (&b)[RD][NT][2], HOOKF hook) {
  constexpr int KC = GEO::KC, STEPS = 5 * KC, HP = MT / 2;
  u32x4 a[2][2][2];
  rd_load_a<GEO>(a[0], va, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int tap = 0; tap < 5; ++tap)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int st = tap * KC + kc, ri = st % RD;
      const bool zero = st == 0, last_kc = kc + 1 == KC;
#pragma unroll
      for (int hp = 0; hp < HP; ++hp) {
        const int cur = (st * HP + hp) & 1;
        if (hp + 1 < HP) rd_load_a<GEO>(a[cur ^ 1], va, tap, kc, hp + 1);
        else rd_load_a<GEO>(a[cur ^ 1], va, last_kc ? tap + 1 : tap, last_kc ? 0 : kc + 1, 0);
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (zero) vb_three<true>(acc[2 * hp + sm][t], a[cur][sm], b[ri][t]);
            else vb_three<false>(acc[2 * hp + sm][t], a[cur][sm], b[ri][t]);
          }
      }
      if (st + RD < STEPS) rd_load_b<GEO, NT>(b[ri], w, st + RD);
      hook(st);
      // region pipeline: DS reads first, then 8 x (3 MFMAs, VPT VALU), the VMEM reads last
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPT, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
}
struct EpiState { float k[4][2], sum[4], dm[4][2], sq[4], bmean; };
// slice st (0 .. 19) of rd_gn_mish<2, 256, true> + rd_store2 for the tile in acc
__device__ __forceinline__ void epi_slice(int st, f32x4 (&acc)[4][2], const Epi<2>& e, EpiState& S, char* vs) {
  constexpr float inv_n = 1.f / 256.f;
  if (st < 4) {
    const int sm = st;
    if (st == 0) S.bmean = group_colsum<8>(e.b[0] + e.b[1]) * 16.f * inv_n;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      S.k[sm][t] = e.is[t];
      v = fmaf((acc[sm][t][0] + acc[sm][t][1]) + (acc[sm][t][2] + acc[sm][t][3]), S.k[sm][t], v);
    }
    S.sum[sm] = add_xor32(add_xor16(group_colsum<8>(v)));
  } else if (st < 8) {
    const int sm = st - 4;
    const float mean = fmaf(S.sum[sm], inv_n, S.bmean);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      S.dm[sm][t] = mean - e.b[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = fmaf(acc[sm][t][r], S.k[sm][t], -S.dm[sm][t]);
        v = fmaf(d, d, v);
      }
    }
    S.sq[sm] = add_xor32(add_xor16(group_colsum<8>(v)));
  } else if (st < 16) {
    const int sm = (st - 8) >> 1, t = (st - 8) & 1;
    const float rstd = __builtin_amdgcn_rsqf(fmaf(S.sq[sm], inv_n, 1e-5f));
    GnCoef cf = gn_coef(S.dm[sm][t], rstd, e.g[t], e.be[t]);
    cf.sa *= S.k[sm][t];
    const ActScale as = act_scale(1.f);
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      const f32x2_t o = gn_mish2<true>(f32x2_t{acc[sm][t][r], acc[sm][t][r + 1]}, cf, f32x2_t{e.tb[t], e.tb[t]}, as);
      acc[sm][t][r] = o.x;
      acc[sm][t][r + 1] = o.y;
    }
  } else {
    const int sm = st - 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const F16Pair f = f16_split2(acc[sm][0][r], acc[sm][1][r]);
      *reinterpret_cast<unsigned*>(vs + (sm * G128::RPS + r) * 16) = f.hi;
      *reinterpret_cast<unsigned*>(vs + G128::PS + (sm * G128::RPS + r) * 16) = f.lo;
    }
  }
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_loop(ConvP p, float* out, int nconv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
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
  int woff[2] = {(2 * wave) * G128::FRAGS5 * 64 + lane, (2 * wave + 1) * G128::FRAGS5 * 64 + lane};
  __syncthreads();
  rd_store2<G128>(slabA + vs_off, accA);
  __syncthreads();
  EpiState S;
  for (int k = 0; k < nconv; ++k) {
    // (per-conv pointers, as in the kernel: no address hoisting out of the loop; the barrier sits on the OFFSETS so that the
    // loads stay global_load -- behind an opaque pointer they become flat_load, which also counts on lgkmcnt)
    asm volatile("" : "+v"(woff[0]), "+v"(woff[1]));
    const u32x4* wp[2] = {reinterpret_cast<const u32x4*>(p.w) + woff[0], reinterpret_cast<const u32x4*>(p.w) + woff[1]};
    const Epi<2> e = epi_load<2>(p.par, p.par + 128, p.par + 256, p.par + 384, p.par + 512, c0);
    {
      u32x4 ring[3][2][2];
      rd_ring_load<G128, 2, 3>(ring, wp);
      rd_taps_hook<G128, 2, 4, 3>(accA, slabA + va_off, wp, ring, [&](int st) { epi_slice(st, accB, e, S, slabB + vs_off); });
      __syncthreads();
    }
    {
      u32x4 ring[3][2][2];
      rd_ring_load<G128, 2, 3>(ring, wp);
      rd_taps_hook<G128, 2, 4, 3>(accB, slabB + va_off, wp, ring, [&](int st) { epi_slice(st, accA, e, S, slabA + vs_off); });
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int t = 0; t < 2; ++t) s += accA[i][t][0] + accA[i][t][3] + accB[i][t][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
constexpr int SAMPLES_PER_WG = 8;
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
  int woff[2] = {(2 * wave) * G128::FRAGS5 * 64 + lane, (2 * wave + 1) * G128::FRAGS5 * 64 + lane};
  __syncthreads();
  rd_store2<G128>(slabA + vs_off, accA);
  __syncthreads();
  for (int k = 0; k < nconv; ++k) {
    // (per-conv pointers, as in the kernel: no address hoisting out of the loop; the barrier sits on the OFFSETS so that the
    // loads stay global_load -- behind an opaque pointer they become flat_load, which also counts on lgkmcnt)
    asm volatile("" : "+v"(woff[0]), "+v"(woff[1]));
    const u32x4* wp[2] = {reinterpret_cast<const u32x4*>(p.w) + woff[0], reinterpret_cast<const u32x4*>(p.w) + woff[1]};
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
#ifdef UB_NB
  const int nconv = 64, nb = UB_NB;
#else
  const int nconv = 64, nb = 2048 / SAMPLES_PER_WG;
#endif
  const size_t wbytes = (size_t)(8 * G128::FRAGS5 + 8) * 64 * 16;
  std::vector<uint16_t> hw(wbytes / 2);
  std::mt19937 rng(1);
  for (auto& v : hw) { const _Float16 f = (_Float16)(((int)(rng() % 2001) - 1000) * 1e-4f); memcpy(&v, &f, 2); }
  std::vector<float> par(5 * 128);
  for (int i = 0; i < 128; ++i) { par[i] = 0.01f * (i % 7); par[128 + i] = 1.f + 0.01f * (i % 5); par[256 + i] = 0.02f * (i % 3); par[384 + i] = 0.05f; par[512 + i] = 1.f; }
  uint4* dw; float* dpar; float* dout;
  hipMalloc(&dw, wbytes); hipMalloc(&dpar, par.size() * 4); hipMalloc(&dout, (size_t)nb * 512 * 4);
  hipMemcpy(dw, hw.data(), wbytes, hipMemcpyHostToDevice); hipMemcpy(dpar, par.data(), par.size() * 4, hipMemcpyHostToDevice);
  ConvP p{dw, dpar};
#ifdef FAT
  const size_t shm = 2 * G128::BYTES + 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_loop), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
#elif defined(W8_LAUNCH)
  const size_t shm = Rd8::BYTES + W8_SLOTS * W8_STEP_BYTES;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_loop), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
#else
  const size_t shm = 0;
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
#ifdef W8_LAUNCH
    hipLaunchKernelGGL(conv_loop, dim3(nb), dim3(512), shm, 0, p, dout, nconv);
#else
    hipLaunchKernelGGL(conv_loop, dim3(nb), dim3(256), shm, 0, p, dout, nconv);
#endif
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> ho(8); hipMemcpy(ho.data(), dout, 32, hipMemcpyDeviceToHost);
    printf("%s: %d workgroups x %d samples, %d convs: %.1f us -> %.2f us per conv of the workgroups a CU holds  (check %.4f, err %s)\n",
#ifdef FAT
           "fat ",
#else
           "base",  // (PARTS builds print base too; the mask is in the file name)
#endif
           nb, SAMPLES_PER_WG, nconv, ms * 1e3, ms * 1e3 / nconv, ho[3], hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
