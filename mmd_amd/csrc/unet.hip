// TemporalUnet forward for gfx950 (MI355X), the whole network in ONE launch (unet_kernel).  Every Conv1d /
// ConvTranspose1d of the reference network (mmd/models/diffusion_models/temporal_unet.py:121-174,
// mmd/models/layers/layers.py:261-358) is an fp32-accurate GEMM on the matrix pipe, with GroupNorm + Mish + time-bias /
// residual fused into the epilogue:
//   * f16x2: an fp32 operand is split into two fp16 pieces (round to nearest, twice) and a product is three
//     v_mfma_f32_16x16x32_f16 (a1*w0 + a0*w1 + a0*w0) with fp32 accumulation -- 1/5 of the fp32 MFMA's pipe time, at
//     least its accuracy.  Power-of-two scales keep the pieces inside fp16's range: per output channel for weights
//     (host), static for the inputs of an RTB's second conv (bounded by GroupNorm), dynamic per sample for the
//     residual stream (dyn_scale); all of them leave through the GroupNorm epilogue's coefficients.
//   * every conv is a DIRECT convolution (taps = row-shifted views of an fp16 slab): downs.0 and ups.1 + final block
//     wave-private (wave = sample, no workgroup barriers inside the stage), downs.1 / downs.2 + mid / ups.0 on workgroup slabs
//     (a wave owns 1-2 n-tiles x 2-4 samples); strided tails read their slab at stride 2, transposed tails = two parity passes.
//   * no register spills (a reload waits for every weight load in flight): downs.2's residual tile is parked
//     lane-privately in LDS; epilogue parameters and the residual conv's weights are requested ahead of their use.
//
// Layout.  The trajectory tensor is channels-last [n_traj, 64, 4] fp32 in HBM on both sides (no transposes).  A
// workgroup (4 waves) owns 4 whole samples for the entire forward: activations live in LDS slabs (fp32 row form
// [sample][L+4][C+2] for downs.2's input; fp16 row form [piece][lane group][K chunk][row][8 ch], RdGeo / RlGeo / RwGeo) and
// in register tiles; the two skip connections wait in registers for the up path; nothing but the input, the output
// and the weights touches HBM/L2.  In the 16x16 C/D layout a lane holds 4 consecutive positions of one channel per tile, so a GroupNorm group is a few lanes of one DPP row (x row blocks): the statistics are
// in-register + cross-lane reductions.  Weights are pre-packed on the host in MFMA B-fragment order (fp16 pairs) and fetched straight from L2 through a register ring (no LDS
// staging: a B element is used once per workgroup).  HISTORY.md section 3.1 has the measurements behind each choice.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/mmd_amd.h"
#include "../../include/mmd_amd_debug.h"
#include "common.h"
#include "f16x2.h"
#include "gn_mish.h"
#include "guide_dev.h"
#include "unet_spec.h"

namespace mmd {


enum { RES_NONE = 0, RES_IDENT = 1, RES_CONV = 2 };

// final Conv1dBlock(32->32, k5) + Conv1d(32->4, k1) of the network
struct FinalArgs {
  float* out;             // eps [n, 64, 4]
  const uint4* w5;        // f16x2 pack of the k5 conv (interleaved column pairs)
  const float* isc;       // [32] inverse channel scales of w5
  const float* bias;      // [32]
  const float* gamma;     // [32] GroupNorm weight
  const float* beta;      // [32] GroupNorm bias
  float act;              // static power-of-two scale of the block's output activations = the 1x1 conv's f16x2 input
  const uint4* w1_bf;     // f16x2 pack of the 1x1 conv (one n-tile, columns >= 4 zero)
  const float* is1;       // [4] inverse channel scales of w1_bf / act
  const float* w1_bias;   // [4]
};

// GroupNorm-epilogue parameters of a conv for the lane's NT adjacent channels.  They are REQUESTED BEFORE the conv's taps (the
// stage bodies call epi_load ahead of the weight ring): read inside the epilogue they cost every conv an exposed L2 round trip
// (~0.5 us, 25 times per forward) -- nothing else is in flight at that point and the statistics need the bias at once.
template <int NT> struct Epi { float b[NT], g[NT], be[NT], is[NT], tb[NT]; };
template <int NT>
__device__ __forceinline__ Epi<NT> epi_load(const float* b, const float* g, const float* be, const float* tb, const float* isc, int c0) {
  Epi<NT> e;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    e.b[t] = b[c0 + t];
    e.g[t] = g[c0 + t];
    e.be[t] = be[c0 + t];
    e.is[t] = isc ? isc[c0 + t] : 1.f;
    e.tb[t] = tb ? tb[c0 + t] : 0.f;
  }
  return e;
}

// The lane's value combined with the same lane of the neighbouring 16-lane row (xor 16) / of the other wave half (xor 32):
// v_permlane16_swap / v_permlane32_swap (gfx950) exchange the odd rows of one operand with the even rows of the other, so with
// both operands = v the two results are (row 0, row 0, row 2, row 2) and (row 1, row 1, row 3, row 3) -- one VALU
// instruction instead of a ds_bpermute round trip through the LDS in the dependent chain of every GroupNorm reduction.
struct RowPair { float a, b; };
__device__ __forceinline__ RowPair rows_xor16(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return RowPair{__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1])};
}
__device__ __forceinline__ RowPair rows_xor32(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return RowPair{__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1])};
}
__device__ __forceinline__ float add_xor16(float v) { const RowPair r = rows_xor16(v); return r.a + r.b; }
__device__ __forceinline__ float add_xor32(float v) { const RowPair r = rows_xor32(v); return r.a + r.b; }
__device__ __forceinline__ float max_xor16(float v) { const RowPair r = rows_xor16(v); return fmaxf(r.a, r.b); }
__device__ __forceinline__ float max_xor32(float v) { const RowPair r = rows_xor32(v); return fmaxf(r.a, r.b); }

template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)));
}
__device__ __forceinline__ float row_max16(float v) {      // max over the 16 lanes of a DPP row, in all of them (v >= 0)
  v = dpp_max<0xB1>(v);
  v = dpp_max<0x4E>(v);
  v = dpp_max<0x141>(v);
  v = dpp_max<0x140>(v);
  return v;
}
constexpr int MX_SLOTS = 8;                                  // partial maxima per sample (waves x lane groups sharing a sample)
constexpr int MX_REGION = 4 * MX_SLOTS;                      // region 0: the conv being prepared; 1 / 2: downs.2's skip2 / the
constexpr int MX_FLOATS = 3 * MX_REGION;                     // mid blocks' output, kept for ups.0's conv A

// acc[mt] += A(slab rows, taps x CP channels) * B(packed).  abase[mt] is the lane's slab offset of (row, k=lane>>5)
// for tap 0; tap t reads STR floats further.  wp points at this lane's float4 of the first k-group.
// B fragments are prefetched FOUR k-groups (32 MFMAs = 2048 cycles) ahead through a 4-register ring so the L2
// latency of a weight fetch never sits in front of the MFMA that consumes it; pack_b pads every packed tensor with 4
// zero groups so the ring may over-read unconditionally.
// A compiler-level memory barrier right after a ring refill: the weight loads are read-only, so LLVM is otherwise free to
// sink them down to their first use (one k-group later: the L2 latency then sits in front of the MFMA again).
// The machine scheduler gets a full barrier at the same point, or it hoists the VALU consumers of an LDS read up to the
// read (and with them the s_waitcnt), which exposes the LDS latency once per k-step.
#ifdef MMD_NO_PIN                        // (tools/ubench/fatwave_conv.hip: the scheduler is steered by sched_group_barrier there)
#define MMD_PIN_LOADS() do { } while (0)
#else
#define MMD_PIN_LOADS()                 \
  do {                                  \
    asm volatile("" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);  \
  } while (0)
#endif

// Phase tracing (side builds with -DMMD_TRACE only; tools/dbg/trace_phases.py): lane 0 of every wave stamps the 100 MHz
// wall clock at tagged points into a [block][wave][256] table set with mmd_debug_set_trace().
#ifdef MMD_TRACE
__device__ unsigned long long* g_trace = nullptr;
#define TR(tag)                                                                                                       \
  do {                                                                                                                \
    if (g_trace && (threadIdx.x & 63) == 0)                                                                           \
      g_trace[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 256 + (tag)] = wall_clock64();                          \
  } while (0)
#else
#define TR(tag) do { } while (0)
#endif

// The thread index through an opaque copy, for addresses that depend on nothing but the thread: inside the persistent kernel's step
// loop they are loop invariant, and hoisted out of the loop they were kept -- spilled to scratch -- across the whole forward (a
// reload waits for every weight load in flight).  Recomputing them where they are used costs a few VALU instructions.
__device__ __forceinline__ int opaque_tid() {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  return tid;
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {   // v + v[DPP-permuted lane] in one VALU op
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// ----------------------------------------------------------------------------------------------------------------
enum { TAIL_NONE = 0, TAIL_DOWN = 1, TAIL_UP = 2 };
constexpr int MAX_IDENT = 3;

struct RtbPtrs {
  const float* ba; const float* ga; const float* bea; const float* tb;   // conv A: bias, GroupNorm weight / bias, time bias [C_out]
  const float* bb; const float* gb; const float* beb;                    // conv B
  const uint4* wa_bf; const uint4* wb_bf;   // f16x2 packs of the two convs
  const float* isa; const float* isb;       // [C_out] inverse per-channel weight scales of the f16x2 packs (isb: / act_a)
  float act_a;                              // static power-of-two scale of conv A's output activations = conv B's f16x2 input
};

struct ChainArgs {
  const float* in0;                      // [n, L, C0] network input (first chain only)
  RtbPtrs r0;
  const uint4* wa0_c1_bf;                // conv A pack of the second input chunk (up stages: cat(x, skip))
  const float* br;                       // bias of the 1x1 residual conv
  const float* isr;                      // [C_out] inverse scale of the residual weights in their f16x2 pack
  const uint4* wres_bf;                  // f16x2 pack of the 1x1 residual conv (first input chunk)
  const uint4* wres_c1_bf;               // ... of the second input chunk (up stages)
  const uint4* wt_bf0; const uint4* wt_bf1;   // f16x2 pack(s) of the tail conv: Downsample1d, or the two parity passes of Upsample1d
  const float* ist0; const float* ist1;       // ... their inverse channel scales
  RtbPtrs ri[MAX_IDENT];
  const float* bt;                       // tail conv bias
  int n;
};

// Shape of a stage: RTB (C0 [+ C1 concatenated] -> CM channels, 1x1 residual conv), N_IDENT identity RTBs (the skip tensor is
// the output of RTB number MID_AFTER), an optional strided / transposed tail conv; L = length of the level.  A workgroup owns
// SPB = 4 samples in every stage.  XSTR / XSS: row and sample stride (floats) of the one row-form fp32 slab left, the stage's
// input as the previous stage's tail hands it over (downs.1 -> downs.2): [sample][2 + position][channel], even row stride (two
// channels per 8-byte access), sample stride padded to 16 (mod 32) floats.
template <int C0_, int C1_, int CM_, int L_, int RES0_, int N_IDENT_, int MID_AFTER_, int TAIL_>
struct ChainCfg {
  static constexpr int C0 = C0_, C1 = C1_, CM = CM_, L = L_, RES0 = RES0_, N_IDENT = N_IDENT_;
  static constexpr int MID_AFTER = MID_AFTER_, TAIL = TAIL_, SPB = 4;
  static constexpr int C0P = (C0 + 7) / 8 * 8, XSTR = C0P + 2, SROWS = L + 4;
  static constexpr int XSS = SROWS * XSTR + (16 - (SROWS * XSTR) % 32 + 32) % 32;
  static_assert(CM % 32 == 0 && L >= 16 && N_IDENT <= MAX_IDENT, "stage shape");
};

// sum over the CPG adjacent lanes (channels) of a GroupNorm group, same value in all of them
template <int CPG>
__device__ __forceinline__ float group_colsum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  if constexpr (CPG >= 8) v = dpp_add<0x141>(v);
  if constexpr (CPG >= 16) v = dpp_add<0x140>(v);
  return v;
}


//                  C0   C1   CM   L   RES0      N_IDENT MID_AFTER TAIL
using CH_D0 = ChainCfg<4, 0, 32, 64, RES_CONV, 1, -1, TAIL_DOWN>;     // downs.0: RTB, RTB, Downsample1d
using CH_D1 = ChainCfg<32, 0, 64, 32, RES_CONV, 1, 1, TAIL_DOWN>;     // downs.1 (skip1 = output of its 2nd RTB)
using CH_D2 = ChainCfg<64, 0, 128, 16, RES_CONV, 3, 1, TAIL_NONE>;    // downs.2 + mid_block1/2 (skip2 after downs.2)
using CH_U0 = ChainCfg<128, 128, 64, 16, RES_CONV, 1, -1, TAIL_UP>;   // ups.0: cat(x, skip2) RTB, RTB, Upsample1d
using CH_U1 = ChainCfg<64, 64, 32, 32, RES_CONV, 1, -1, TAIL_UP>;     // ups.1: cat(x, skip1) RTB, RTB, Upsample1d

constexpr int cmax(int a, int b) { return a > b ? a : b; }
// LDS of a workgroup: the largest stage is downs.2 / ups.0 -- the row-form fp32 x slab of downs.2's input + the 128-channel Rd
// slab (2 x 21504 B) behind it; downs.1 (input slab + 64-channel slab) and the four private slabs of the wave-private stages
// (downs.0, ups.1 + final block) fit below it (static_asserts in the stage bodies).
constexpr int MX_OFF = ((CH_D2::SPB * CH_D2::XSS * 4 + 255) / 256 * 256 + 43008) / 4 + 8;
// + the per-sample maxima of the dynamic input scales + the second part of downs.2's lane-private residual parking area (the
// first part is the stage's dead x slab: 5 + 3 float4 per thread)
constexpr int PARK2_OFF = MX_OFF + MX_FLOATS;
constexpr int UNET_LDS_FLOATS = PARK2_OFF + 3 * 256 * 4;

// ----------------------------------------------------------------------------------------------------------------
// DIRECT f16x2 convolutions on a row-form slab (downs.1, downs.2 + mid blocks, ups.0; the wave-private stages use the same
// GEMM loop on per-wave slabs).  Every weight fragment is re-used on 64 GEMM rows (a wave's unit is 1-2 n-tiles x the FOUR M
// tiles = samples of the workgroup), a conv is ONE slab store and one barrier pair, 8 accumulator streams (32-64 registers).
// (HISTORY.md section 3.1 has the history: the transform-domain forms of rounds 1-2 were bound by their weight stream.)
// GEMM: M tile s = sample s, row i = position; the taps of a k = 5 conv are row-shifted views of the slab
//     Rd[piece][lane group j][chunk kc][row = 20 s + 2 + position][8 channels]   (fp16, 2-row zero halo per sample)
// (channel block kc + KC j, KC = C / 32: the four blocks of a K = 32 chunk lie G = a multiple of 256 B apart, so a b128 A
// read is conflict free; the blocks of one lane group BX = 1280 + 32 B, so the epilogue's dword stores -- lanes = 4
// channel pairs x 4 blocks x 4 position groups -- are 2-way at worst, which is free).  C/D layout: a lane holds positions
// 4 g .. 4 g + 3 (g = lane >> 4) of ALL four samples for its 1-2 channels; a GroupNorm group (8 lanes x 16 positions) is
// reduced by three DPP steps and two cross-row shuffles.  Weights: per n-tile [tap][chunk kc][piece][lane] x 16 B.
// ----------------------------------------------------------------------------------------------------------------
template <int C> struct RdGeo {
  static constexpr int KC = C / 32, RPS = 20, BX = 4 * RPS * 16 + 32, G = (KC * BX + 255) / 256 * 256, PS = 4 * G, BYTES = 2 * PS;
  static constexpr int FRAGS5 = 5 * KC * 2;                  // weight fragments per n-tile of a k = 5 conv
  static constexpr int tile_row(int m) { return m * RPS; }   // first slab row (halo included) of M tile m = sample m
};
// The same slab for a stage of length 32 (downs.1): a sample is TWO M tiles (positions 0 .. 15, 16 .. 31) between its 2-row
// halos, 36 rows per sample; a wave's four M tiles are the two samples of its sample pair.
template <int C> struct RlGeo {
  static constexpr int KC = C / 32, RPS = 36, BX = 4 * RPS * 16 + 32, G = (KC * BX + 255) / 256 * 256, PS = 4 * G, BYTES = 2 * PS;
  static constexpr int FRAGS5 = 5 * KC * 2, FRAGS3 = 3 * KC * 2;
  static constexpr int tile_row(int m) { return (m >> 1) * RPS + (m & 1) * 16; }
};
template <class GEO>
__device__ __forceinline__ void rd_zero_halo(char* slab) {
  constexpr int TOT = 2 * 4 * GEO::KC * 4 * 4;               // pieces x lane groups x chunks x samples x halo rows, 16 B each
  for (int idx = opaque_tid(); idx < TOT; idx += 256) {
    const int hr = idx & 3, sm = (idx >> 2) & 3, blk = (idx >> 4) % (4 * GEO::KC), q = idx / (64 * GEO::KC);
    *reinterpret_cast<uint4*>(slab + q * GEO::PS + (blk / GEO::KC) * GEO::G + (blk % GEO::KC) * GEO::BX +
                              (sm * GEO::RPS + (hr < 2 ? hr : GEO::RPS - 4 + hr)) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
}
template <class GEO, int NT>
__device__ __forceinline__ void rd_load_b(u32x4 (&b)[NT][2], const u32x4* const (&w)[NT], int step) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 2; ++q) b[t][q] = w[t][(step * 2 + q) * 64];
}
// A fragments of one sample PAIR (M tiles 2 hp, 2 hp + 1; SM = 1: of the single M tile) at slab-row offset rowoff, chunk kc
template <class GEO, int SM = 2>
__device__ __forceinline__ void rd_load_a(u32x4 (&a)[SM][2], const char* va, int rowoff, int kc, int hp) {
#pragma unroll
  for (int sm = 0; sm < SM; ++sm)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      a[sm][q] = *reinterpret_cast<const u32x4*>(va + q * GEO::PS + kc * GEO::BX + (GEO::tile_row(SM * hp + sm) + rowoff) * 16);
}
constexpr int RD_RD = 2;                    // weight ring depth in steps
#ifndef MMD_D2_RD
#define MMD_D2_RD 3
#endif
#ifndef MMD_D1_RD
#define MMD_D1_RD 3
#endif
#ifndef MMD_U0C_RD
#define MMD_U0C_RD 4
#endif
#ifndef MMD_U0_RD
#define MMD_U0_RD 3
#endif
template <class GEO, int NT, int RD = RD_RD>
__device__ __forceinline__ void rd_ring_load(u32x4 (&b)[RD][NT][2], const u32x4* const (&w)[NT]) {
#pragma unroll
  for (int i = 0; i < RD; ++i) rd_load_b<GEO, NT>(b[i], w, i);
  MMD_PIN_LOADS();
}
// acc[sample][tile] (+)= conv over TAPS taps (slab rows TAP0 .. TAP0 + TAPS - 1 relative to the output position) x the C
// channels of the slab; va = slab + the lane's A offset (lane group lane >> 4, row lane & 15); w[tile] = the tile's pack +
// lane; b = ring pre-loaded with the first RD_RD steps.  RES: the stage's 1x1 residual conv rides on the centre tap's A
// fragments (res[sample][tile] (+)=, weights wr[tile] = [chunk kc][piece] + lane).  FRESH: start from zero.
template <class GEO, int NT, int TAP0, int TAPS, bool FRESH, bool RES, int MT = 4, int RD = RD_RD>
__device__ __forceinline__ void rd_taps(f32x4 (&acc)[MT][NT], f32x4 (&res)[MT][NT], const char* va, const u32x4* const (&w)[NT],
                                        const u32x4* const (&wr)[NT], u32x4 (&b)[RD][NT][2]) {
  // M tiles are processed in pairs (SM = 2), or a single one (MT = 1: the half-sample waves of unet_kernel<2> at L = 32)
  constexpr int KC = GEO::KC, STEPS = TAPS * KC, SM = MT >= 2 ? 2 : 1, HP = MT / SM;
  static_assert(MT == 1 || MT % 2 == 0, "M tiles are processed in pairs");
  // A fragments are double-buffered by M-tile pair (half a step = 2 M tiles x NT n-tiles x 3 MFMAs): 32 registers
  u32x4 a[2][SM][2];
  rd_load_a<GEO, SM>(a[0], va, TAP0, 0, 0);
  // (the residual conv's weights are requested RES_LOOK steps before the centre tap's step that uses them: loaded there,
  // every one of its steps would wait for an L2 round trip; all up front, they would cost 32 registers for two taps)
  constexpr int C0 = (2 - TAP0) * KC;                        // the centre tap's first step
  constexpr int RES_LOOK = C0 < 3 ? C0 : 3;
  u32x4 brp[RES ? KC : 1][NT][2];
  auto load_br = [&](int kc) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) brp[kc][t][q] = wr[t][(kc * 2 + q) * 64];
  };
  // Every step (tap, chunk kc) is unrolled: the ring slot step % RD and the A buffer parity are static for any depth, the
  // loop has no branches, and the scheduler sees the whole conv (rolled over the taps, downs.2's convs ran 5 % slower at <= 512
  // trajectories).
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int st = tap * KC + kc, ri = st % RD;
      const bool zero = FRESH && st == 0, last_kc = kc + 1 == KC, with_res = RES && TAP0 + tap == 2;
      if constexpr (RES) {
        if (st + RES_LOOK >= C0 && st + RES_LOOK < C0 + KC) load_br(st + RES_LOOK - C0);
      }
#pragma unroll
      for (int hp = 0; hp < HP; ++hp) {
        // the next half step's A fragments (the next M-tile pair; then the next chunk, or chunk 0 of the next tap; past the
        // last step: a valid, unused read)
        const int cur = (st * HP + hp) & 1;
        if (hp + 1 < HP) rd_load_a<GEO, SM>(a[cur ^ 1], va, TAP0 + tap, kc, hp + 1);
        else rd_load_a<GEO, SM>(a[cur ^ 1], va, last_kc ? TAP0 + tap + 1 : TAP0 + tap, last_kc ? 0 : kc + 1, 0);
        MMD_PIN_LOADS();
        const u32x4(&ac)[SM][2] = a[cur];
        const u32x4(&bc)[NT][2] = b[ri];
#pragma unroll
        for (int sm = 0; sm < SM; ++sm)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (zero) vb_three<true>(acc[SM * hp + sm][t], ac[sm], bc[t]);
            else vb_three<false>(acc[SM * hp + sm][t], ac[sm], bc[t]);
          }
        if constexpr (RES) {
          if (with_res) {
#pragma unroll
            for (int sm = 0; sm < SM; ++sm)
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                const u32x4(&bw)[2] = brp[kc][t];
                if (FRESH && kc == 0) vb_three<true>(res[SM * hp + sm][t], ac[sm], bw);
                else vb_three<false>(res[SM * hp + sm][t], ac[sm], bw);
              }
          }
        }
      }
      if (st + RD < STEPS) rd_load_b<GEO, NT>(b[ri], w, st + RD);
      MMD_PIN_LOADS();
    }
}
// GroupNorm + Mish of the direct-layout tile acc[sample][tile] (raw f16x2 conv output: true value = acc * isc[tile] *
// inv[sample]) + add(sample, tile, r); NG = values per group (16 channels x 16 positions for two interleaved n-tiles at C =
// 128, 8 x 16 for one n-tile at C = 64); the lane's NT channels all belong to one group, which is 8 lanes x the wave's four
// 16-lane rows (position groups).
template <int NT, int NG, bool ACT, class ADD, int NS>
__device__ __forceinline__ void rd_gn_mish(f32x4 (&acc)[NS][NT], const float (&bias)[NT], const float (&gamma)[NT],
                                           const float (&beta)[NT], const float (&isc)[NT], const float (&inv)[NS],
                                           const ActScale& as, ADD add) {
  constexpr float inv_n = 1.f / (float)NG;
  float bsum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) bsum += bias[t];
  const float bmean = group_colsum<8>(bsum) * 16.f * inv_n;
  float k[NS][NT], sum[NS], dm[NS][NT], sq[NS];
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) {
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      k[sm][t] = isc[t] * inv[sm];
      v = fmaf((acc[sm][t][0] + acc[sm][t][1]) + (acc[sm][t][2] + acc[sm][t][3]), k[sm][t], v);
    }
    sum[sm] = group_colsum<8>(v);
  }
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) sum[sm] = add_xor16(sum[sm]);
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) sum[sm] = add_xor32(sum[sm]);
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) {
    const float mean = fmaf(sum[sm], inv_n, bmean);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      dm[sm][t] = mean - bias[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = fmaf(acc[sm][t][r], k[sm][t], -dm[sm][t]);
        v = fmaf(d, d, v);
      }
    }
    sq[sm] = group_colsum<8>(v);
  }
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) sq[sm] = add_xor16(sq[sm]);
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) sq[sm] = add_xor32(sq[sm]);
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) {
    const float rstd = __builtin_amdgcn_rsqf(fmaf(sq[sm], inv_n, 1e-5f));   // (argument >= 1e-5: no denormal handling needed)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      GnCoef cf = gn_coef(dm[sm][t], rstd, gamma[t], beta[t]);
      cf.sa *= k[sm][t];
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const f32x2_t o = gn_mish2<ACT>(f32x2_t{acc[sm][t][r], acc[sm][t][r + 1]}, cf, f32x2_t{add(sm, t, r), add(sm, t, r + 1)}, as);
        acc[sm][t][r] = o.x;
        acc[sm][t][r + 1] = o.y;
      }
    }
  }
}
// per-sample |x| maxima of a direct-layout tile -> mx region 0 (and region2 if > 0): row_max16, one cross-row step,
// lanes 0 / 32 write the wave's two partials: slots 2 wave + {0, 1} of MX_SLOTS = 8
template <int NT, int NS>
__device__ __forceinline__ void rd_dyn_out(const f32x4 (&acc)[NS][NT], float* mx, int wave, int lane, int region2) {
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(acc[sm][t][r]));
    m = row_max16(m);
    m = max_xor16(m);
    if ((lane & 31) == 0) {
      mx[sm * MX_SLOTS + 2 * wave + (lane >> 5)] = m;
      if (region2) mx[region2 * MX_REGION + sm * MX_SLOTS + 2 * wave + (lane >> 5)] = m;
    }
  }
}
__device__ __forceinline__ float mx_read(const float* mx, int sm) {
  const float4 p = *reinterpret_cast<const float4*>(mx + sm * MX_SLOTS), q = *reinterpret_cast<const float4*>(mx + sm * MX_SLOTS + 4);
  return fmaxf(fmaxf(fmaxf(p.x, p.y), fmaxf(p.z, p.w)), fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
}
// two-interleaved-n-tile tile (lane: channels c0, c0 + 1 = block 4 wave + (n >> 2), dword n & 3; positions 4 g + r) -> slab
template <class GEO, int NS>
__device__ __forceinline__ void rd_store2(char* vs, const f32x4 (&acc)[NS][2]) {   // vs = slab + lane's (block, row 2 + 4 g, dword)
#pragma unroll
  for (int sm = 0; sm < NS; ++sm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const F16Pair f = f16_split2(acc[sm][0][r], acc[sm][1][r]);
      *reinterpret_cast<unsigned*>(vs + (sm * GEO::RPS + r) * 16) = f.hi;
      *reinterpret_cast<unsigned*>(vs + GEO::PS + (sm * GEO::RPS + r) * 16) = f.lo;
    }
}
// one-n-tile tile (lane: channel c, positions 4 g + r of the NS samples): the lanes of a pair (n, n ^ 1) swap half the samples,
// the even lane stores the first NS / 2 samples of channels (c, c + 1), the odd lane the other half of (c - 1, c).
// vs = slab + the lane's (block of c, row 2 + 4 g, dword (c & 7) >> 1) offset
template <class GEO, int NS>
__device__ __forceinline__ void rd_store1(char* vs, const f32x4 (&acc)[NS][1], int lane) {
  constexpr int HS = NS / 2;
  const bool odd = lane & 1;
  if constexpr (NS == 1) {
    // one sample: BOTH lanes of a pair store the same dword (c, c + 1) -- no divergent branch around the store: with the store under
    // `if (even lane)` the results were wrong on the hardware (the cross-lane read apparently ends up inside the branch, where the odd
    // lanes are disabled and read as 0); the duplicate same-value LDS write is free
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float own = acc[0][0][r];
      const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0xB1, 0xf, 0xf, true));
      const F16Pair f = f16_split2(odd ? recv : own, odd ? own : recv);        // (low channel, high channel)
      *reinterpret_cast<unsigned*>(vs + r * 16) = f.hi;
      *reinterpret_cast<unsigned*>(vs + GEO::PS + r * 16) = f.lo;
    }
  }
#pragma unroll
  for (int h = 0; h < HS; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float send = odd ? acc[h][0][r] : acc[HS + h][0][r];
      const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, true));
      const float own = odd ? acc[HS + h][0][r] : acc[h][0][r];
      const F16Pair f = f16_split2(odd ? recv : own, odd ? own : recv);        // (low channel, high channel)
      char* p = vs + ((odd ? HS + h : h) * GEO::RPS + r) * 16;
      *reinterpret_cast<unsigned*>(p) = f.hi;
      *reinterpret_cast<unsigned*>(p + GEO::PS) = f.lo;
    }
}
// row-form fp32 slab [sample][20][XSTR] (2-row halo) of C channels -> the Rd slab, times the sample's dynamic scale
template <int C, int XSS, int XSTR, int NS>
__device__ __forceinline__ void rowform_to_rd(const float* xslab, char* slab, const float* mx) {
  using GEO = RdGeo<C>;
  constexpr int CP2 = C / 2, ITEMS = 16 * NS * CP2;          // (sample, position) x channel pairs
  static_assert(ITEMS % 256 == 0 && XSTR % 2 == 0, "items per thread; 8-byte aligned channel pairs");
  const int tid = opaque_tid();
#pragma unroll
  for (int it = 0; it < ITEMS / 256; ++it) {
    const int idx = it * 256 + tid;
    const int cp = idx % CP2, sp = idx / CP2, sm = sp >> 4, pos = sp & 15;
    const float sc = dyn_scale(mx_read(mx, sm)).s;
    const float2 t = *reinterpret_cast<const float2*>(xslab + sm * XSS + (2 + pos) * XSTR + 2 * cp);
    const F16Pair f = f16_split2(t.x * sc, t.y * sc);
    const int blk = cp >> 2;
    char* dst = slab + (blk / GEO::KC) * GEO::G + (blk % GEO::KC) * GEO::BX + (sm * GEO::RPS + 2 + pos) * 16 + (cp & 3) * 4;
    *reinterpret_cast<unsigned*>(dst) = f.hi;
    *reinterpret_cast<unsigned*>(dst + GEO::PS) = f.lo;
  }
}

// ----------------------------------------------------------------------------------------------------------------
// WAVE-PRIVATE direct stages (downs.0; one sample per wave): at L = 64 a sample is four M tiles of its own, and with 32
// channels a wave holds a whole sample (4 M tiles x 2 interleaved n-tiles = 8 accumulators) -- so every conv of the stage
// reads only what the same wave wrote: no workgroup barrier anywhere inside the stage (LDS operations of one wave execute in
// order), GroupNorm statistics and the dynamic input scales are wave reductions, and the weights (20 KB per conv) are
// streamed by each wave.  Slab of one sample: Rw[piece][lane group j][chunk kc][row = 2 + position][8 channels].
// ----------------------------------------------------------------------------------------------------------------
template <int C, int L> struct RwGeo {
  static constexpr int KC = C / 32, RPS = 16, ROWS = L + 4, BX = ROWS * 16 + 32, G = (KC * BX + 255) / 256 * 256, PS = 4 * G;
  static constexpr int BYTES = 2 * PS, FRAGS5 = 5 * KC * 2, FRAGS3 = 3 * KC * 2;
  static constexpr int tile_row(int m) { return m * RPS; }   // M tile m = positions 16 m .. 16 m + 15 of the wave's sample
};
// A slab read at STRIDE 2 (Downsample1d = a k3 conv at the even positions only): M tile m = outputs 16 m .. 16 m + 15 of a
// sample = slab rows 2 (16 m + n) + tap of it, so the lane offset in `va` is n x 32 B (the caller adds n x 16 to the stride-1
// va) and the 16 lanes of a row group span 512 B: the b128 reads are 2-way bank conflicts, half as many of them and half the
// MFMAs and weight loads of the conv evaluated at every position.  SROWS = slab rows from one sample to the next.
template <class GEO, int SROWS, int TILES_PER_SAMPLE> struct Stride2 : GEO {
  static constexpr int tile_row(int m) { return (m / TILES_PER_SAMPLE) * SROWS + (m % TILES_PER_SAMPLE) * 32; }
};
__device__ __forceinline__ float wave_sum_rows(float v) {   // v + the same lane of the other three 16-lane rows
  v = add_xor16(v);
  v = add_xor32(v);
  return v;
}
// GroupNorm + Mish of ONE sample's tile acc[M tile][tile] (positions 16 mt + 4 g + r; true value = acc * isc[tile] * inv);
// GL = lanes per group (the lane's NT channels belong to one group), NG = values per group
template <int MT, int NT, int GL, int NG, bool ACT, class ADD>
__device__ __forceinline__ void rw_gn_mish(f32x4 (&acc)[MT][NT], const float (&bias)[NT], const float (&gamma)[NT],
                                           const float (&beta)[NT], const float (&isc)[NT], float inv, const ActScale& as, ADD add) {
  constexpr float inv_n = 1.f / (float)NG;
  auto gsum = [](float v) {
    v = dpp_add<0xB1>(v);
    if constexpr (GL >= 4) v = dpp_add<0x4E>(v);
    if constexpr (GL >= 8) v = dpp_add<0x141>(v);
    return wave_sum_rows(v);
  };
  float k[NT], bsum = 0.f, v = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    k[t] = isc[t] * inv;
    bsum += bias[t];
    float st = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) st += (acc[mt][t][0] + acc[mt][t][1]) + (acc[mt][t][2] + acc[mt][t][3]);
    v = fmaf(st, k[t], v);
  }
  // mean over the group of (x + bias): every channel's bias counts at the sample's 16 MT positions, 4 per lane row
  const float mean = (gsum(v) + gsum(bsum) * (float)(4 * MT)) * inv_n;
  float dm[NT], q = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    dm[t] = mean - bias[t];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = fmaf(acc[mt][t][r], k[t], -dm[t]);
        q = fmaf(d, d, q);
      }
  }
  const float rstd = __builtin_amdgcn_rsqf(fmaf(gsum(q), inv_n, 1e-5f));
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    GnCoef cf = gn_coef(dm[t], rstd, gamma[t], beta[t]);
    cf.sa *= k[t];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const f32x2_t o = gn_mish2<ACT>(f32x2_t{acc[mt][t][r], acc[mt][t][r + 1]}, cf, f32x2_t{add(mt, t, r), add(mt, t, r + 1)}, as);
        acc[mt][t][r] = o.x;
        acc[mt][t][r + 1] = o.y;
      }
  }
}
// The same for the stages whose waves are whole samples in unet_kernel<4> and HALF samples in unet_kernel<2> (downs.0, ups.1 +
// final block): the statistics of a sample are DEFINED through its two halves (positions [0, L / 2) and [L / 2, L): HT = 1 or 2
// M tiles each) -- per half the mean of x = acc k + bias over the group and the sum of squared deviations from THAT mean, combined
// by the pairwise update mean = (m0 + m1) / 2, M2 = (M2_0 + M2_1) + (m1 - m0)^2 N / 4.  A whole-sample wave evaluates both halves
// itself; two half-sample waves evaluate one each and swap (mean, M2) through LDS -- the same arithmetic, the same bits.
struct HalfStat { float mean, m2; };
template <int GL>
__device__ __forceinline__ float rw_gsum(float v) {          // sum over the GL lanes of a group and the wave's four 16-lane rows
  v = dpp_add<0xB1>(v);
  if constexpr (GL >= 4) v = dpp_add<0x4E>(v);
  if constexpr (GL >= 8) v = dpp_add<0x141>(v);
  return wave_sum_rows(v);
}
// bsum4 = rw_gsum(sum of the lane's biases) (every channel's bias counts once per position: 4 positions per lane row and M tile)
template <int HT, int NT, int GL, int NG>
__device__ __forceinline__ HalfStat rw_half_stat(const f32x4 (&acc)[HT][NT], const float (&bias)[NT], const float (&k)[NT], float bsum4) {
  float v = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float st = (acc[0][t][0] + acc[0][t][1]) + (acc[0][t][2] + acc[0][t][3]);
    if constexpr (HT == 2) st += (acc[1][t][0] + acc[1][t][1]) + (acc[1][t][2] + acc[1][t][3]);
    v = fmaf(st, k[t], v);
  }
  HalfStat h;
  h.mean = (rw_gsum<GL>(v) + bsum4 * (float)(4 * HT)) * (2.f / (float)NG);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float dm = h.mean - bias[t];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = fmaf(acc[mt][t][r], k[t], -dm);
        q = fmaf(d, d, q);
      }
  }
  h.m2 = rw_gsum<GL>(q);
  return h;
}
struct GnStat { float mean, rstd; };
template <int NG>
__device__ __forceinline__ GnStat gn_combine(const HalfStat& h0, const HalfStat& h1) {
  const float dlt = h1.mean - h0.mean;
  const float m2 = fmaf(dlt * dlt, 0.25f * (float)NG, h0.m2 + h1.m2);
  return GnStat{0.5f * (h0.mean + h1.mean), __builtin_amdgcn_rsqf(fmaf(m2, 1.f / (float)NG, 1e-5f))};
}
// GroupNorm affine + Mish + add(mt, t, r) on MT tiles with the sample's statistics st
template <int MT, int NT, bool ACT, class ADD>
__device__ __forceinline__ void rw_gn_apply(f32x4 (&acc)[MT][NT], const float (&bias)[NT], const float (&gamma)[NT],
                                            const float (&beta)[NT], const float (&k)[NT], const GnStat& st, const ActScale& as, ADD add) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    GnCoef cf = gn_coef(st.mean - bias[t], st.rstd, gamma[t], beta[t]);
    cf.sa *= k[t];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const f32x2_t o = gn_mish2<ACT>(f32x2_t{acc[mt][t][r], acc[mt][t][r + 1]}, cf, f32x2_t{add(mt, t, r), add(mt, t, r + 1)}, as);
        acc[mt][t][r] = o.x;
        acc[mt][t][r + 1] = o.y;
      }
  }
}
// ... of a WHOLE sample held by one wave (MT = 2 HT M tiles)
template <int MT, int NT, int GL, int NG, bool ACT, class ADD>
__device__ __forceinline__ void rw_gn_mish_whole(f32x4 (&acc)[MT][NT], const float (&bias)[NT], const float (&gamma)[NT],
                                                 const float (&beta)[NT], const float (&isc)[NT], float inv, const ActScale& as, ADD add) {
  constexpr int HT = MT / 2;
  static_assert(MT == 2 || MT == 4, "two halves of one or two M tiles");
  float k[NT], bsum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    k[t] = isc[t] * inv;
    bsum += bias[t];
  }
  const float bsum4 = rw_gsum<GL>(bsum);
  const HalfStat h0 = rw_half_stat<HT, NT, GL, NG>(reinterpret_cast<const f32x4(&)[HT][NT]>(acc[0]), bias, k, bsum4);
  const HalfStat h1 = rw_half_stat<HT, NT, GL, NG>(reinterpret_cast<const f32x4(&)[HT][NT]>(acc[HT]), bias, k, bsum4);
  rw_gn_apply<MT, NT, ACT>(acc, bias, gamma, beta, k, gn_combine<NG>(h0, h1), as, add);
}
// ... of the HALF sample this wave holds (HT M tiles; half index hf); the partner wave's statistics arrive through xch = the
// sample's exchange area [2 halves][64 lanes] of HalfStat (one workgroup barrier; the caller guarantees another barrier between
// this read and the next write of the area)
template <int HT, int NT, int GL, int NG, bool ACT, class ADD>
__device__ __forceinline__ void rw_gn_mish_half(f32x4 (&acc)[HT][NT], const float (&bias)[NT], const float (&gamma)[NT],
                                                const float (&beta)[NT], const float (&isc)[NT], float inv, const ActScale& as, ADD add,
                                                HalfStat* xch, int hf, int lane) {
  float k[NT], bsum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    k[t] = isc[t] * inv;
    bsum += bias[t];
  }
  const float bsum4 = rw_gsum<GL>(bsum);
  const HalfStat own = rw_half_stat<HT, NT, GL, NG>(acc, bias, k, bsum4);
  xch[hf * 64 + lane] = own;
  __syncthreads();
  const HalfStat other = xch[(hf ^ 1) * 64 + lane];
  const GnStat st = hf ? gn_combine<NG>(other, own) : gn_combine<NG>(own, other);
  rw_gn_apply<HT, NT, ACT>(acc, bias, gamma, beta, k, st, as, add);
}
template <int MT, int NT>
__device__ __forceinline__ float rw_absmax(const f32x4 (&acc)[MT][NT]) {   // the sample's |x| maximum, in every lane
  float m = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(acc[mt][t][r]));
  m = row_max16(m);
  m = max_xor16(m);
  return max_xor32(m);
}
// two-interleaved-n-tile tile of one sample (lane: channels 2 n, 2 n + 1 (+ 32 per further pair); positions 16 mt + 4 g + r)
// -> the wave's slab; vs = slab + the lane's (block n >> 2, row 2 + 4 g, dword n & 3) offset
template <class GEO, int MT>
__device__ __forceinline__ void rw_store2(char* vs, const f32x4 (&acc)[MT][2]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const F16Pair f = f16_split2(acc[mt][0][r], acc[mt][1][r]);
      *reinterpret_cast<unsigned*>(vs + (mt * 16 + r) * 16) = f.hi;
      *reinterpret_cast<unsigned*>(vs + GEO::PS + (mt * 16 + r) * 16) = f.lo;
    }
}
__device__ __forceinline__ void wave_lds_fence() {           // a wave's own LDS writes before its own later reads
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// downs.0 (4 -> 32 -> 32 channels at L = 64, Downsample1d): wave = sample.  The first conv's K is 5 taps x 4 channels = 20
// of the 32 slots of ONE MFMA chunk (im2col: lane group j holds taps 2 j, 2 j + 1 -- two consecutive 8-byte rows of the
// [row][4 channel] input slab), its 1x1 residual conv a second chunk with only the centre tap's slots non-zero.  The raw
// network input has no bounded range: dynamic scale from the sample's own maximum.  The stride-2 tail reads its slab at stride 2
// (Stride2: 3 taps on two M tiles, 36 MFMAs; the A reads are 2-way bank conflicted, half as many as at every position).
// The stage's output goes straight into the next stage's input slab (RlGeo<32>) as f16 pieces under the sample's own dynamic
// scale, behind a workgroup barrier (it aliases the waves' slabs); the sample's maximum goes to mx.
template <class CF, int NS>
__device__ __forceinline__ void chain_body_d0w(const ChainArgs& a, float* lds, int n0, int lane, int wave, int trb, int tb_off = 0) {
  static_assert(CF::L == 64 && CF::CM == 32 && CF::C0 == 4 && CF::C1 == 0 && CF::RES0 == RES_CONV && CF::N_IDENT == 1 &&
                    CF::TAIL == TAIL_DOWN, "downs.0");
  using GW = RwGeo<32, 64>;
  constexpr int XIN = 72 * 8;                                // bytes per piece of the [row][4 channel] input slab (rows -2 .. 69)
  constexpr int W_BYTES = GW::BYTES + 2 * XIN + 128;
  static_assert(4 * W_BYTES <= MX_OFF * 4, "four private slabs");
  char* const slab = reinterpret_cast<char*>(lds) + wave * W_BYTES;
  char* const xin = slab + GW::BYTES;
  const int n = lane & 15, g = lane >> 4, c0 = 2 * n;
  const char* const va = slab + g * GW::G + n * 16;          // A fragment: row lane & 15, lane group lane >> 4 (KC = 1: block j)
  char* const vs = slab + (n >> 2) * GW::G + (2 + 4 * g) * 16 + (n & 3) * 4;
  auto wptr = [&](const uint4* w, int frags, int t) { return reinterpret_cast<const u32x4*>(w) + (size_t)t * frags * 64 + lane; };
  TR(trb + 0);
  // ---- stage the sample: lane = position; [row = 2 + position][4 channels] x two pieces, zero rows around it
  float inv_in;
  {
    const bool valid = wave < NS && n0 + wave < a.n;           // (NS < 4: the other waves run on zeros, see unet_kernel)
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) v = *reinterpret_cast<const float4*>(a.in0 + ((size_t)(n0 + wave) * 64 + lane) * 4);
    float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    m = row_max16(m);
    m = max_xor16(m);
    m = max_xor32(m);
    const DynScale ds = dyn_scale(m);
    inv_in = ds.inv;
    const F16Pair p0 = f16_split2(v.x * ds.s, v.y * ds.s), p1 = f16_split2(v.z * ds.s, v.w * ds.s);
    *reinterpret_cast<uint2*>(xin + (2 + lane) * 8) = make_uint2(p0.hi, p1.hi);
    *reinterpret_cast<uint2*>(xin + XIN + (2 + lane) * 8) = make_uint2(p0.lo, p1.lo);
    if (lane < 16) {                                         // rows 0, 1, 66 .. 71 of both pieces
      const int row = (lane & 7) < 2 ? (lane & 7) : 64 + (lane & 7);
      *reinterpret_cast<uint2*>(xin + (lane >> 3) * XIN + row * 8) = make_uint2(0u, 0u);
    }
    // zero halo rows of the conv slab (rows 0, 1, 66, 67 of the 4 blocks x 2 pieces)
    if (lane < 32) *reinterpret_cast<uint4*>(slab + (lane >> 4) * GW::PS + ((lane >> 2) & 3) * GW::G + ((lane & 3) < 2 ? (lane & 3) : 64 + (lane & 3)) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  wave_lds_fence();
  f32x4 acc[4][2], res[4][2];
  const Epi<2> e0a = epi_load<2>(a.r0.ba, a.r0.ga, a.r0.bea, a.r0.tb + tb_off, a.r0.isa, c0);
  const float br0[2] = {a.br[c0], a.br[c0 + 1]}, isr0[2] = {a.isr[c0], a.isr[c0 + 1]};
  // ---- RTB 0 conv A (im2col chunk) + the 1x1 residual conv
  {
    u32x4 b[2][2], br[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        b[t][q] = wptr(a.r0.wa_bf, 2, t)[q * 64];
        br[t][q] = wptr(a.wres_bf, 2, t)[q * 64];
      }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      u32x4 af[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint2* p = reinterpret_cast<const uint2*>(xin + q * XIN + (mt * 16 + n + 2 * g) * 8);
        const uint2 lo = p[0], hi = p[1];
        af[q] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        vb_three<true>(acc[mt][t], af, b[t]);
        vb_three<true>(res[mt][t], af, br[t]);
      }
    }
  }
  const float one = 1.f;
  auto epi = [&](const float* bs, const float* gm, const float* be, const float* tb, const float* isc) {
    return epi_load<2>(bs, gm, be, tb, isc, c0);
  };
  auto gn = [&](auto conv_a, const Epi<2>& e, float inv, float act_s) {
    if constexpr (decltype(conv_a)::value) {
      const float t0 = e.tb[0] * act_s, t1 = e.tb[1] * act_s;
      rw_gn_mish_whole<4, 2, 2, 256, true>(acc, e.b, e.g, e.be, e.is, inv, act_scale(act_s), [&](int, int t, int) { return t ? t1 : t0; });
    } else {
      rw_gn_mish_whole<4, 2, 2, 256, false>(acc, e.b, e.g, e.be, e.is, inv, ActScale{}, [&](int mt, int t, int r) { return res[mt][t][r]; });
    }
  };
  // The whole weight set of a conv (5 taps x 2 n-tiles x 2 pieces = 20 KB per wave) is requested BEFORE the epilogue that
  // produces the conv's input (preload), so the L2 latency hides behind GroupNorm + Mish instead of in front of the MFMAs.
  u32x4 ring[5][2][2];
  auto preload = [&](const uint4* w) {
    const u32x4* wp[2] = {wptr(w, GW::FRAGS5, 0), wptr(w, GW::FRAGS5, 1)};
    rd_ring_load<GW, 2, 5>(ring, wp);
  };
  auto conv = [&](const uint4* w) {                          // one 32 -> 32 conv over the tile in acc (already scaled)
    const u32x4* wp[2] = {wptr(w, GW::FRAGS5, 0), wptr(w, GW::FRAGS5, 1)};
    rw_store2<GW, 4>(vs, acc);
    wave_lds_fence();
    rd_taps<GW, 2, 0, 5, true, false, 4, 5>(acc, res, va, wp, wp, ring);
    wave_lds_fence();                                        // (the next store must not overtake these reads)
  };
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int t = 0; t < 2; ++t) res[mt][t] = res[mt][t] * (isr0[t] * inv_in) + br0[t];
  preload(a.r0.wb_bf);
  gn(std::true_type{}, e0a, inv_in, a.r0.act_a);
  TR(trb + 1);
  {
    const Epi<2> e = epi(a.r0.bb, a.r0.gb, a.r0.beb, nullptr, a.r0.isb);
    conv(a.r0.wb_bf);
    preload(a.ri[0].wa_bf);
    gn(std::false_type{}, e, one, 1.f);
  }
  TR(trb + 2);
  // ---- identity RTB
  {
    const RtbPtrs& R = a.ri[0];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) res[mt][t] = acc[mt][t];
    const DynScale ds = dyn_scale(rw_absmax<4, 2>(acc));
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[mt][t] *= ds.s;
    const Epi<2> ea = epi(R.ba, R.ga, R.bea, R.tb + tb_off, R.isa);
    conv(R.wa_bf);
    preload(R.wb_bf);
    gn(std::true_type{}, ea, ds.inv, R.act_a);
    TR(trb + 3);
    const Epi<2> eb = epi(R.bb, R.gb, R.beb, nullptr, R.isb);
    conv(R.wb_bf);
    gn(std::false_type{}, eb, one, 1.f);
    TR(trb + 4);
  }
  // ---- tail: Downsample1d = Conv1d(k3, s2, p1): y[p] = sum_t x[p + t - 1] W_t at the even p
  {
    const DynScale ds = dyn_scale(rw_absmax<4, 2>(acc));
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[mt][t] *= ds.s;
    const u32x4* wt[2] = {wptr(a.wt_bf0, GW::FRAGS3, 0), wptr(a.wt_bf0, GW::FRAGS3, 1)};
    u32x4 ring3[3][2][2];
    const float bt[2] = {a.bt[c0], a.bt[c0 + 1]}, ist[2] = {a.ist0[c0] * ds.inv, a.ist0[c0 + 1] * ds.inv};
    rd_ring_load<GW, 2, 3>(ring3, wt);
    rw_store2<GW, 4>(vs, acc);
    wave_lds_fence();
    // (outputs q = 16 mt + 4 g + r = the even positions 2 q: two M tiles read at stride 2)
    f32x4 y[2][2];
    rd_taps<Stride2<GW, 0, 2>, 2, 1, 3, true, false, 2, 3>(y, y, va + n * 16, wt, wt, ring3);
    float mo = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          y[mt][t][r] = fmaf(y[mt][t][r], ist[t], bt[t]);
          mo = fmaxf(mo, fabsf(y[mt][t][r]));
        }
    mo = row_max16(mo);
    mo = max_xor16(mo);
    mo = max_xor32(mo);
    __syncthreads();                                         // every wave is done with its slab: the next stage's slab aliases them
    if (lane < MX_SLOTS) (lds + MX_OFF)[wave * MX_SLOTS + lane] = mo;
    // -> the next stage's input slab (RlGeo<32>: rows 36 sample + 2 + q), channels 2 n, 2 n + 1 = block n >> 2, dword n & 3, as
    //    f16 pieces under the sample's own dynamic scale (the next stage reads it from mx)
    using GN = RlGeo<32>;
    const float so = dyn_scale(mo).s;
    char* const lb = reinterpret_cast<char*>(lds);
    char* xb = lb + (n >> 2) * GN::G + (wave * GN::RPS + 2 + 4 * g) * 16 + (n & 3) * 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const F16Pair f = f16_split2(y[mt][0][r] * so, y[mt][1][r] * so);
        *reinterpret_cast<unsigned*>(xb + (16 * mt + r) * 16) = f.hi;
        *reinterpret_cast<unsigned*>(xb + GN::PS + (16 * mt + r) * 16) = f.lo;
      }
    if (lane < 32)                                           // halo rows 0, 1, 34, 35 of the sample's 4 blocks x 2 pieces
      *reinterpret_cast<uint4*>(lb + (lane >> 4) * GN::PS + ((lane >> 2) & 3) * GN::G +
                                (wave * GN::RPS + ((lane & 3) < 2 ? (lane & 3) : 32 + (lane & 3))) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  TR(trb + 5);
}

// Cooperative weight staging of the half-sample stages (unet_kernel<2>): there every wave of the workgroup needs ALL B fragments of
// a 32 -> 32 conv, and four waves fetching the same 20 KB through the CU's 64 B/clk vector-memory path took ~0.8 us per conv
// (the issue of the loads itself blocks: profiles/r04_trace_512_half_sample.txt).  Instead the conv's NFRAG fragments (1 KiB
// each: [lane] x 16 B, contiguous in the pack) go global -> LDS ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4: no
// registers; wave w moves fragments w, w + 4, ...), one conv ahead into the other of two buffers, and the GEMM loop reads its B
// fragments from LDS through a two-step ring.  The issuing wave waits for its own pieces (vmcnt) before the barrier that
// publishes the slab the conv reads.
template <int NFRAG>
__device__ __forceinline__ void stage_weights(const uint4* w, char* dst, int wave, int lane) {
  static_assert(NFRAG % 4 == 0, "fragments are dealt to the four waves");
  const uint4* src = w + lane;
#pragma unroll
  for (int i = 0; i < NFRAG / 4; ++i) {
    const int f = wave + 4 * i;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 64),
                                     (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ void staged_weights_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
constexpr int WBUF_BYTES = 20 * 1024;                        // the largest staged conv: 32 -> 32, k = 5, two pieces

// downs.0 for unet_kernel<2> (two trajectories per workgroup): a sample is split between two waves by POSITION -- wave = (sample sp =
// wave >> 1, half hf = wave & 1: positions 32 hf .. 32 hf + 31 = M tiles 2 hf, 2 hf + 1 of the sample's four), so all four waves
// work on real data with half the MFMAs and half the epilogue values each (the whole-sample form left two waves on zeros).  The
// two waves share the sample's slab (a conv's taps reach two rows into the other half: one workgroup barrier between the slab
// store and the taps), GroupNorm statistics are the two-half combination of rw_half_stat (one exchange through LDS per conv,
// whose barrier also orders the next slab store behind the partner's taps), dynamic scales take the sample's maximum from the
// two waves' partials in mx.  Per-sample arithmetic is that of chain_body_d0w: bitwise equal results.
template <class CF, int NV = 2>     // NV: the workgroup's REAL samples (unet_kernel<1>: sample 1 is fed zeros and never stored)
__device__ __forceinline__ void chain_body_d0s(const ChainArgs& a, float* lds, int n0, int lane, int wave, int trb, int tb_off = 0) {
  static_assert(CF::L == 64 && CF::CM == 32 && CF::C0 == 4 && CF::C1 == 0 && CF::RES0 == RES_CONV && CF::N_IDENT == 1 &&
                    CF::TAIL == TAIL_DOWN, "downs.0");
  using GW = RwGeo<32, 64>;
  constexpr int XIN = 72 * 8;
  constexpr int W_BYTES = GW::BYTES + 2 * XIN + 128;          // (per SAMPLE here)
  const int sp = wave >> 1, hf = wave & 1;
  char* const slab = reinterpret_cast<char*>(lds) + sp * W_BYTES;
  char* const xin = slab + GW::BYTES;
  float* const mx = lds + MX_OFF;
  HalfStat* const xch = reinterpret_cast<HalfStat*>(lds + PARK2_OFF) + sp * 128;   // (downs.2's parking area is idle in this stage)
  static_assert(2 * 128 * sizeof(HalfStat) <= 3 * 256 * 16, "exchange area inside the parking area");
  char* const wb0 = reinterpret_cast<char*>(lds) + 2 * W_BYTES;   // two weight buffers behind the two samples' slabs
  char* const wb1 = wb0 + WBUF_BYTES;
  static_assert(2 * W_BYTES + 2 * WBUF_BYTES <= MX_OFF * 4, "slabs + weight buffers below the maxima");
  static_assert(GW::FRAGS5 * 2 * 1024 <= WBUF_BYTES, "a staged conv fits its buffer");
  stage_weights<2 * GW::FRAGS5>(a.r0.wb_bf, wb0, wave, lane);     // RTB 0's conv B: lands while the sample is staged and conv A runs
  const int n = lane & 15, g = lane >> 4, c0 = 2 * n;
  const char* const va = slab + g * GW::G + (n + 32 * hf) * 16;
  char* const vs = slab + (n >> 2) * GW::G + (2 + 4 * g + 32 * hf) * 16 + (n & 3) * 4;
  auto wptr = [&](const uint4* w, int frags, int t) { return reinterpret_cast<const u32x4*>(w) + (size_t)t * frags * 64 + lane; };
  // the sample's maximum of a per-wave partial: both waves publish theirs in the sample's mx slots (4 hf .. 4 hf + 3), barrier
  auto sample_max = [&](float own) {
    if (lane < 4) mx[sp * MX_SLOTS + 4 * hf + lane] = own;
    __syncthreads();
    return mx_read(mx, sp);
  };
  TR(trb + 0);
  // ---- stage the sample: every wave loads all of it (lane = position: the exact maximum without an exchange) and writes its half
  float inv_in;
  {
    const bool valid = sp < NV && n0 + sp < a.n;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) v = *reinterpret_cast<const float4*>(a.in0 + ((size_t)(n0 + sp) * 64 + lane) * 4);
    float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    m = row_max16(m);
    m = max_xor16(m);
    m = max_xor32(m);
    const DynScale ds = dyn_scale(m);
    inv_in = ds.inv;
    const F16Pair p0 = f16_split2(v.x * ds.s, v.y * ds.s), p1 = f16_split2(v.z * ds.s, v.w * ds.s);
    if ((lane >> 5) == hf) {
      *reinterpret_cast<uint2*>(xin + (2 + lane) * 8) = make_uint2(p0.hi, p1.hi);
      *reinterpret_cast<uint2*>(xin + XIN + (2 + lane) * 8) = make_uint2(p0.lo, p1.lo);
    }
    if (hf == 0) {
      if (lane < 16) {                                         // rows 0, 1, 66 .. 71 of both pieces
        const int row = (lane & 7) < 2 ? (lane & 7) : 64 + (lane & 7);
        *reinterpret_cast<uint2*>(xin + (lane >> 3) * XIN + row * 8) = make_uint2(0u, 0u);
      }
      if (lane < 32) *reinterpret_cast<uint4*>(slab + (lane >> 4) * GW::PS + ((lane >> 2) & 3) * GW::G + ((lane & 3) < 2 ? (lane & 3) : 64 + (lane & 3)) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  f32x4 acc[2][2], res[2][2];
  const Epi<2> e0a = epi_load<2>(a.r0.ba, a.r0.ga, a.r0.bea, a.r0.tb + tb_off, a.r0.isa, c0);
  const float br0[2] = {a.br[c0], a.br[c0 + 1]}, isr0[2] = {a.isr[c0], a.isr[c0 + 1]};
  u32x4 b0[2][2], br[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      b0[t][q] = wptr(a.r0.wa_bf, 2, t)[q * 64];
      br[t][q] = wptr(a.wres_bf, 2, t)[q * 64];
    }
  __syncthreads();
  // ---- RTB 0 conv A (im2col chunk) + the 1x1 residual conv on the half's two M tiles
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    u32x4 af[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint2* p = reinterpret_cast<const uint2*>(xin + q * XIN + ((2 * hf + i) * 16 + n + 2 * g) * 8);
      const uint2 lo = p[0], hi = p[1];
      af[q] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      vb_three<true>(acc[i][t], af, b0[t]);
      vb_three<true>(res[i][t], af, br[t]);
    }
  }
  const float one = 1.f;
  auto epi = [&](const float* bs, const float* gm, const float* be, const float* tb, const float* isc) {
    return epi_load<2>(bs, gm, be, tb, isc, c0);
  };
  auto gn = [&](auto conv_a, const Epi<2>& e, float inv, float act_s) {
    if constexpr (decltype(conv_a)::value) {
      const float t0 = e.tb[0] * act_s, t1 = e.tb[1] * act_s;
      rw_gn_mish_half<2, 2, 2, 256, true>(acc, e.b, e.g, e.be, e.is, inv, act_scale(act_s), [&](int, int t, int) { return t ? t1 : t0; }, xch, hf, lane);
    } else {
      rw_gn_mish_half<2, 2, 2, 256, false>(acc, e.b, e.g, e.be, e.is, inv, ActScale{}, [&](int mt, int t, int r) { return res[mt][t][r]; }, xch, hf, lane);
    }
  };
  u32x4 ring[2][2][2];
  // one 32 -> 32 conv over the half tile in acc (already scaled), its weights staged in wb; behind the barrier the NEXT conv's
  // NEXT_FRAGS fragments start on their way into the other buffer (every wave is past the conv that read it)
  auto conv = [&](char* wb, auto next_frags, const uint4* w_next, char* wb_next, int tr = -1) {
    const u32x4* wp[2] = {reinterpret_cast<const u32x4*>(wb) + lane, reinterpret_cast<const u32x4*>(wb) + GW::FRAGS5 * 64 + lane};
    if (tr >= 0) TR(tr);
    rw_store2<GW, 2>(vs, acc);
    if (tr >= 0) TR(tr + 1);
    staged_weights_landed();
    __syncthreads();                                         // both halves of the sample and the conv's weights are in LDS
    if (tr >= 0) TR(tr + 2);
    if constexpr (decltype(next_frags)::value > 0) stage_weights<decltype(next_frags)::value>(w_next, wb_next, wave, lane);
    rd_ring_load<GW, 2, 2>(ring, wp);
    rd_taps<GW, 2, 0, 5, true, false, 2, 2>(acc, res, va, wp, wp, ring);
    if (tr >= 0) TR(tr + 3);
  };
  using F5 = std::integral_constant<int, 2 * GW::FRAGS5>;
  using F3 = std::integral_constant<int, 2 * GW::FRAGS3>;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < 2; ++t) res[mt][t] = res[mt][t] * (isr0[t] * inv_in) + br0[t];
  gn(std::true_type{}, e0a, inv_in, a.r0.act_a);
  TR(trb + 1);
  {
    const Epi<2> e = epi(a.r0.bb, a.r0.gb, a.r0.beb, nullptr, a.r0.isb);
    conv(wb0, F5{}, a.ri[0].wa_bf, wb1);
    gn(std::false_type{}, e, one, 1.f);
  }
  TR(trb + 2);
  // ---- identity RTB
  {
    const RtbPtrs& R = a.ri[0];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) res[mt][t] = acc[mt][t];
    const DynScale ds = dyn_scale(sample_max(rw_absmax<2, 2>(acc)));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[mt][t] *= ds.s;
    const Epi<2> ea = epi(R.ba, R.ga, R.bea, R.tb + tb_off, R.isa);
    conv(wb1, F5{}, R.wb_bf, wb0, trb + 6);
    TR(trb + 10);
    gn(std::true_type{}, ea, ds.inv, R.act_a);
    TR(trb + 3);
    const Epi<2> eb = epi(R.bb, R.gb, R.beb, nullptr, R.isb);
    conv(wb0, F3{}, a.wt_bf0, wb1);
    gn(std::false_type{}, eb, one, 1.f);
    TR(trb + 4);
  }
  // ---- tail: Downsample1d = Conv1d(k3, s2, p1) at the even positions: the half's 16 outputs = ONE M tile read at stride 2
  {
    const DynScale ds = dyn_scale(sample_max(rw_absmax<2, 2>(acc)));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[mt][t] *= ds.s;
    const u32x4* wt[2] = {reinterpret_cast<const u32x4*>(wb1) + lane, reinterpret_cast<const u32x4*>(wb1) + GW::FRAGS3 * 64 + lane};
    const float bt[2] = {a.bt[c0], a.bt[c0 + 1]}, ist[2] = {a.ist0[c0] * ds.inv, a.ist0[c0 + 1] * ds.inv};
    rw_store2<GW, 2>(vs, acc);
    staged_weights_landed();
    __syncthreads();
    f32x4 y[1][2];
    rd_ring_load<GW, 2, 2>(ring, wt);
    rd_taps<Stride2<GW, 0, 2>, 2, 1, 3, true, false, 1, 2>(y, y, va + n * 16, wt, wt, ring);
    float mo = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y[0][t][r] = fmaf(y[0][t][r], ist[t], bt[t]);
        mo = fmaxf(mo, fabsf(y[0][t][r]));
      }
    mo = row_max16(mo);
    mo = max_xor16(mo);
    mo = max_xor32(mo);
    // (the barrier inside: every wave is done with its slab -- the next stage's slab aliases them -- and the sample's maxima
    // are published for the next stage's dynamic scale)
    const float so = dyn_scale(sample_max(mo)).s;
    using GN = RlGeo<32>;
    char* const lb = reinterpret_cast<char*>(lds);
    char* xb = lb + (n >> 2) * GN::G + (sp * GN::RPS + 2 + 16 * hf + 4 * g) * 16 + (n & 3) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const F16Pair f = f16_split2(y[0][0][r] * so, y[0][1][r] * so);
      *reinterpret_cast<unsigned*>(xb + r * 16) = f.hi;
      *reinterpret_cast<unsigned*>(xb + GN::PS + r * 16) = f.lo;
    }
    if (hf == 0 && lane < 32)                                // halo rows 0, 1, 34, 35 of the sample's 4 blocks x 2 pieces
      *reinterpret_cast<uint4*>(lb + (lane >> 4) * GN::PS + ((lane >> 2) & 3) * GN::G +
                                (sp * GN::RPS + ((lane & 3) < 2 ? (lane & 3) : 32 + (lane & 3))) * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  TR(trb + 5);
}

// downs.1 (32 -> 64 -> 64 channels at L = 32, Downsample1d) in the direct form on RlGeo slabs.  Wave w = (n-tile pair np = w &
// 1: channels 32 np + 2 n + t, interleaved columns; sample pair sp = w >> 1: samples 2 sp, 2 sp + 1), so a weight fragment is
// used on four M tiles and fetched by two waves; acc[m][t]: M tile m = sample 2 sp + (m >> 1), positions 16 (m & 1) + 4 g + r.
// A GroupNorm group (8 channels x 32 positions of a sample) is 4 lanes x 2 tiles x the sample's 2 M tiles: wave-internal.
// The input slab arrives from downs.0's tail (f16 pieces, per-sample scales in mx); the strided tail reads its slab at stride 2
// (Stride2) and writes downs.2's row-form fp32 x slab (CFN geometry) + the per-sample maxima to mx.
// skip: the stage's skip tensor (output of its second RTB) in the acc layout.
template <class CF, class CFN, int NS>
__device__ __forceinline__ void chain_body_d1d(const ChainArgs& a, float* lds, int lane, int wave, f32x4 (&skip)[NS][2], int trb, int tb_off = 0) {
  static_assert(CF::L == 32 && CF::CM == 64 && CF::C0 == 32 && CF::C1 == 0 && CF::RES0 == RES_CONV && CF::N_IDENT == 1 &&
                    CF::MID_AFTER == 1 && CF::TAIL == TAIL_DOWN, "downs.1");
  using GI = RlGeo<32>;
  using GH = RlGeo<64>;
  constexpr int H_OFF = GI::BYTES;                           // the 64-channel slab lies behind the input slab
  static_assert(H_OFF + GH::BYTES <= MX_OFF * 4, "input slab + 64-channel slab");
  char* const lb = reinterpret_cast<char*>(lds);
  char* const slabH = lb + H_OFF;
  float* const mx = lds + MX_OFF;
  // SW samples per wave: the pair 2 sp, 2 sp + 1 of a four-sample workgroup, or sample sp of a two-sample one (unet_kernel<2>: all
  // four waves on real samples, half the M tiles each); a sample = 2 M tiles, so a wave has NS of them
  constexpr int SW = NS / 2;
  const int n = lane & 15, g = lane >> 4, np = wave & 1, sp = wave >> 1, s0 = SW * sp;
  const int c0 = 32 * np + 2 * n;
  const char* const vaI = lb + g * GI::G + (s0 * GI::RPS + n) * 16;
  const char* const vaH = slabH + g * GH::G + (s0 * GH::RPS + n) * 16;
  // the lane's channel pair (c0, c0 + 1) = block 4 np + (n >> 2) = (chunk (n >> 2) & 1, lane group 2 np + (n >> 3)), dword n & 3
  char* const vsH = slabH + (2 * np + (n >> 3)) * GH::G + ((n >> 2) & 1) * GH::BX + (s0 * GH::RPS + 2 + 4 * g) * 16 + (n & 3) * 4;
  auto wptr = [&](const uint4* w, int frags, int h) { return reinterpret_cast<const u32x4*>(w) + (size_t)(2 * np + h) * frags * 64 + lane; };
  auto epi = [&](const float* b, const float* gm, const float* be, const float* tb, const float* isc) {
    return epi_load<2>(b, gm, be, tb, isc, c0);
  };
  f32x4 acc[NS][2], res[NS][2];
  constexpr int RD1 = MMD_D1_RD;                               // weight ring depth of the 64 -> 64 convs
  u32x4 ring[RD1][2][2];
  auto store_tile = [&]() {
#pragma unroll
    for (int m = 0; m < NS; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const F16Pair f = f16_split2(acc[m][0][r], acc[m][1][r]);
        *reinterpret_cast<unsigned*>(vsH + (GH::tile_row(m) + r) * 16) = f.hi;
        *reinterpret_cast<unsigned*>(vsH + GH::PS + (GH::tile_row(m) + r) * 16) = f.lo;
      }
  };
  // GroupNorm + Mish of acc, sample by sample (M tiles 2 s, 2 s + 1)
  auto gn = [&](auto conv_a, const Epi<2>& e, const float (&inv)[SW], float act_s) {
#pragma unroll
    for (int sl = 0; sl < SW; ++sl) {
      f32x4(&t)[2][2] = reinterpret_cast<f32x4(&)[2][2]>(acc[2 * sl]);
      if constexpr (decltype(conv_a)::value) {
        const float t0 = e.tb[0] * act_s, t1 = e.tb[1] * act_s;
        rw_gn_mish<2, 2, 4, 256, true>(t, e.b, e.g, e.be, e.is, inv[sl], act_scale(act_s), [&](int, int tt, int) { return tt ? t1 : t0; });
      } else {
        rw_gn_mish<2, 2, 4, 256, false>(t, e.b, e.g, e.be, e.is, inv[sl], ActScale{}, [&](int mt, int tt, int r) { return res[2 * sl + mt][tt][r]; });
      }
    }
  };
  // per-sample |x| maxima of the tile in v -> all eight slots of the two samples (the two waves of a sample pair fill them)
  auto maxima_out = [&](const f32x4 (&v)[NS][2]) {
    float m2[2] = {0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < SW; ++sl) {
      float m = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(v[2 * sl + mt][t][r]));
      m = row_max16(m);
      m = max_xor16(m);
      m2[sl] = max_xor32(m);
    }
    if (lane < 4 * SW) mx[(s0 + (lane >> 2)) * MX_SLOTS + np + 2 * (lane & 3)] = (lane >> 2) ? m2[1] : m2[0];
  };
  // dynamic input scale of the tile in acc from the maxima in mx: scale in place, the inverse scales per sample of the pair
  auto scale_in = [&](float (&inv)[SW]) {
#pragma unroll
    for (int sl = 0; sl < SW; ++sl) {
      const DynScale ds = dyn_scale(mx_read(mx, s0 + sl));
      inv[sl] = ds.inv;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[2 * sl + mt][t] *= ds.s;
    }
  };
  // one 64 -> 64 conv over the tile in acc (already scaled); on entry every wave is past its reads of the slab.  PF (one sample per
  // wave = unet_kernel<2>, the latency-bound launches): the NEXT conv's first ring steps are requested right behind this conv's taps
  // and travel during the epilogue (chain_body_d2d has the measurement); `frags_next`: the next pack's fragments per n-tile.
#ifdef MMD_NO_PF
  constexpr bool PF = false;
#else
  constexpr bool PF = SW == 1;
#endif
  auto prefetch = [&](const uint4* w, int frags) {
    if constexpr (PF) {
      const u32x4* wp[2] = {wptr(w, frags, 0), wptr(w, frags, 1)};
      rd_ring_load<GH, 2, RD1>(ring, wp);
    }
  };
  auto conv = [&](const uint4* w, const uint4* w_next, int frags_next) {
    const u32x4* wp[2] = {wptr(w, GH::FRAGS5, 0), wptr(w, GH::FRAGS5, 1)};
    if constexpr (!PF) rd_ring_load<GH, 2, RD1>(ring, wp);
    store_tile();
    __syncthreads();
    rd_taps<GH, 2, 0, 5, true, false, NS, RD1>(acc, acc, vaH, wp, wp, ring);
    prefetch(w_next, frags_next);
  };
  float one2[SW];
#pragma unroll
  for (int sl = 0; sl < SW; ++sl) one2[sl] = 1.f;

  // =================== RTB 0 (32 -> 64): conv A + the 1x1 residual conv on the centre tap ===================
  {
    u32x4 ring5[5][2][2];
    const u32x4* wpa[2] = {wptr(a.r0.wa_bf, GI::FRAGS5, 0), wptr(a.r0.wa_bf, GI::FRAGS5, 1)};
    const u32x4* wpr[2] = {wptr(a.wres_bf, 2 * GI::KC, 0), wptr(a.wres_bf, 2 * GI::KC, 1)};
    rd_ring_load<GI, 2, 5>(ring5, wpa);
    const Epi<2> e0a = epi(a.r0.ba, a.r0.ga, a.r0.bea, a.r0.tb + tb_off, a.r0.isa);
    const float br[2] = {a.br[c0], a.br[c0 + 1]}, isr[2] = {a.isr[c0], a.isr[c0 + 1]};
    __syncthreads();                                         // downs.0's tail has written the input slab and its maxima
    TR(trb + 0);
    float inv_in[SW];
#pragma unroll
    for (int sl = 0; sl < SW; ++sl) inv_in[sl] = dyn_scale(mx_read(mx, s0 + sl)).inv;
    rd_zero_halo<GH>(slabH);
    rd_taps<GI, 2, 0, 5, true, true, NS, 5>(acc, res, vaI, wpa, wpr, ring5);
#pragma unroll
    for (int m = 0; m < NS; ++m)
#pragma unroll
      for (int t = 0; t < 2; ++t) res[m][t] = res[m][t] * (isr[t] * inv_in[m >> 1]) + br[t];
    prefetch(a.r0.wb_bf, GH::FRAGS5);
    gn(std::true_type{}, e0a, inv_in, a.r0.act_a);
  }
  TR(trb + 1);
  {
    const Epi<2> e = epi(a.r0.bb, a.r0.gb, a.r0.beb, nullptr, a.r0.isb);
    conv(a.r0.wb_bf, a.ri[0].wa_bf, GH::FRAGS5);             // (its slab is not the one conv A reads: no barrier before the store)
    gn(std::false_type{}, e, one2, 1.f);
  }
  TR(trb + 2);
  // =================== identity RTB ===================
  {
    const RtbPtrs& R = a.ri[0];
#pragma unroll
    for (int m = 0; m < NS; ++m)
#pragma unroll
      for (int t = 0; t < 2; ++t) res[m][t] = acc[m][t];
    maxima_out(acc);
    __syncthreads();                                         // the previous conv is done reading the slab
    float inv[SW];
    scale_in(inv);
    const Epi<2> ea = epi(R.ba, R.ga, R.bea, R.tb + tb_off, R.isa);
    conv(R.wa_bf, R.wb_bf, GH::FRAGS5);
    gn(std::true_type{}, ea, inv, R.act_a);
    TR(trb + 3);
    __syncthreads();
    const Epi<2> eb = epi(R.bb, R.gb, R.beb, nullptr, R.isb);
    conv(R.wb_bf, a.wt_bf0, GH::FRAGS3);                     // (next: the strided tail's 3-tap pack)
    gn(std::false_type{}, eb, one2, 1.f);
    TR(trb + 4);
#pragma unroll
    for (int m = 0; m < NS; ++m)
#pragma unroll
      for (int t = 0; t < 2; ++t) skip[m][t] = acc[m][t];
  }
  // =================== tail: Downsample1d = Conv1d(k3, s2, p1): y[p] = sum_t x[p + t - 1] W_t at the even p ===================
  {
    maxima_out(acc);
    __syncthreads();
    float inv[SW];
    scale_in(inv);
    const u32x4* wt[2] = {wptr(a.wt_bf0, GH::FRAGS3, 0), wptr(a.wt_bf0, GH::FRAGS3, 1)};
    const float bt[2] = {a.bt[c0], a.bt[c0 + 1]}, ist[2] = {a.ist0[c0], a.ist0[c0 + 1]};
    if constexpr (!PF) rd_ring_load<GH, 2, RD1>(ring, wt);
    store_tile();
    __syncthreads();
    // (outputs q = 4 g + r = the even positions 2 q of the wave's samples: one M tile each, read at stride 2; the GEMM loop takes
    // M tiles in pairs: a one-sample wave computes its tile twice)
    f32x4 y[2][2];
    rd_taps<Stride2<GH, SW == 2 ? GH::RPS : 0, 1>, 2, 1, 3, true, false, 2, RD1>(y, y, vaH + n * 16, wt, wt, ring);
    float m2[2] = {0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < SW; ++sl) {
      float m = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          y[sl][t][r] = fmaf(y[sl][t][r], ist[t] * inv[sl], bt[t]);
          m = fmaxf(m, fabsf(y[sl][t][r]));
        }
      m = row_max16(m);
      m = max_xor16(m);
      m2[sl] = max_xor32(m);
    }
    __syncthreads();                                         // every wave is done reading the slab the next stage's x slab aliases
    if (lane < 4 * SW) mx[(s0 + (lane >> 2)) * MX_SLOTS + np + 2 * (lane & 3)] = (lane >> 2) ? m2[1] : m2[0];
    // -> the next stage's row-form fp32 x slab [sample][2 + q][CFN::XSTR]
    float* xb = lds + s0 * CFN::XSS + (2 + 4 * g) * CFN::XSTR + c0;
#pragma unroll
    for (int sl = 0; sl < SW; ++sl)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float2*>(xb + sl * CFN::XSS + r * CFN::XSTR) = make_float2(y[sl][0][r], y[sl][1][r]);
  }
  TR(trb + 5);
}

// downs.2 + mid blocks in the direct form.  Wave w owns the n-tiles 2 w, 2 w + 1 with INTERLEAVED columns (column n of tile h =
// channel 32 w + 2 n + h: adjacent channels per lane, dword slab stores) x all four samples; acc[sample][h][r] = position
// 4 (lane >> 4) + r.  acc: the stage's output; mid: the skip tensor (after RTB MID_AFTER).
// Lane-private parking of a 32-register tile in LDS ([i][thread] x 16 B: conflict-free b128, no synchronisation -- a thread
// reads back only what it wrote): the residual tile of an RTB waits there instead of in 32 VGPRs while the block's two convs
// run (the kernel sits at the 256-register limit of two waves per SIMD; a compiler spill to scratch costs a vmcnt(0) wait
// behind every weight load in flight).  Parts 0 .. 4 in the stage's dead x slab, 5 .. 7 behind the maxima.
__device__ __forceinline__ float* park_slot(float* lds, int i) {
  return (i < 5 ? lds + i * 1024 : lds + PARK2_OFF + (i - 5) * 1024) + opaque_tid() * 4;
}
template <int NS>
__device__ __forceinline__ void park_tile(float* lds, const f32x4 (&t)[NS][2]) {
#pragma unroll
  for (int i = 0; i < 2 * NS; ++i) *reinterpret_cast<f32x4*>(park_slot(lds, i)) = t[i >> 1][i & 1];
}
template <int NS>
__device__ __forceinline__ void unpark_tile(float* lds, f32x4 (&t)[NS][2]) {
#pragma unroll
  for (int i = 0; i < 2 * NS; ++i) t[i >> 1][i & 1] = *reinterpret_cast<const f32x4*>(park_slot(lds, i));
}

template <class CF, int NS>
__device__ __forceinline__ void chain_body_d2d(const ChainArgs& a, float* lds, int lane, int wave, f32x4 (&acc)[NS][2],
                                               f32x4 (&mid)[NS][2], int trb, int tb_off = 0) {
  static_assert(CF::L == 16 && CF::CM == 128 && CF::C0 == 64 && CF::C1 == 0 && CF::RES0 == RES_CONV && CF::TAIL == TAIL_NONE &&
                    CF::MID_AFTER >= 1, "downs.2 + mid blocks");
  using G128 = RdGeo<128>;
  using G64 = RdGeo<64>;
  const int n = lane & 15, g = lane >> 4;
  const int c0 = 32 * wave + 2 * n;                          // the lane's channels c0 (tile 0), c0 + 1 (tile 1)
  // LDS: the row-form fp32 x slab (previous stage's tail tile) at the start, the Rd slab behind it (conv A's 64-channel
  // input uses its first bytes in the 64-channel geometry)
  constexpr int S_OFF = (CF::SPB * CF::XSS * 4 + 255) / 256 * 256;
  static_assert(S_OFF + G128::BYTES <= MX_OFF * 4, "x slab + Rd slab must fit below the maxima");
  static_assert(S_OFF >= 5 * 1024 * 4, "the dead x slab holds 5 of the 8 parked float4 per thread");
  char* const slab = reinterpret_cast<char*>(lds) + S_OFF;
  float* const mx = lds + MX_OFF;
  const char* const va128 = slab + g * G128::G + n * 16;     // A fragment: row lane & 15 = position, lane group lane >> 4
  const char* const va64 = slab + g * G64::G + n * 16;
  char* const vs = slab + wave * G128::G + (n >> 2) * G128::BX + (2 + 4 * g) * 16 + (n & 3) * 4;
  auto wptr = [&](const uint4* w, int frags, int h) { return reinterpret_cast<const u32x4*>(w) + (size_t)(2 * wave + h) * frags * 64 + lane; };
  constexpr int RDD = MMD_D2_RD;                               // weight ring depth of the 128 -> 128 convs
  u32x4 ring[RDD][2][2];
  const u32x4* wpa[2] = {wptr(a.r0.wa_bf, G64::FRAGS5, 0), wptr(a.r0.wa_bf, G64::FRAGS5, 1)};
  const u32x4* wpr[2] = {wptr(a.wres_bf, 2 * G64::KC, 0), wptr(a.wres_bf, 2 * G64::KC, 1)};
  rd_ring_load<G64, 2, 2>(reinterpret_cast<u32x4(&)[2][2][2]>(ring), wpa);   // (conv A + residual streams: depth 2, or it spills)
  __syncthreads();                                           // the x slab (previous stage's tail tile) and its maxima are staged
  TR(trb + 0);

  float one4[NS];
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) one4[sm] = 1.f;
  // GroupNorm + Mish of acc.  Conv A (tb != nullptr): + the time bias, output carried times act_s (conv B's static f16x2
  // input scale); conv B: + the residual tile, which comes back from its parking area.  isc: the conv's inverse weight scales,
  // inv: inverse dynamic input scales
  auto epi = [&](const float* b, const float* gm, const float* be, const float* tb, const float* isc) {
    return epi_load<2>(b, gm, be, tb, isc, c0);
  };
  auto gn = [&](auto conv_a, const Epi<2>& e, const float (&inv)[NS], float act_s) {
    if constexpr (decltype(conv_a)::value) {
      const float t0 = e.tb[0] * act_s, t1 = e.tb[1] * act_s;
      rd_gn_mish<2, 256, true>(acc, e.b, e.g, e.be, e.is, inv, act_scale(act_s), [&](int, int t, int) { return t ? t1 : t0; });
    } else {
      f32x4 res[NS][2];
      unpark_tile(lds, res);
      rd_gn_mish<2, 256, false>(acc, e.b, e.g, e.be, e.is, inv, ActScale{}, [&](int sm, int t, int r) { return res[sm][t][r]; });
    }
  };
  // one 128 -> 128 conv over the tile in acc (already scaled for f16x2); on entry every wave is past its reads of the slab.
  // PF (two trajectories per workgroup: the latency-bound launches of <= 512 trajectories, where nothing else on the CU covers an L2
  // round trip): the first RDD weight steps of the NEXT conv are requested right behind this conv's taps, so they travel during the
  // GroupNorm + Mish epilogue and the slab store instead of in front of the first MFMA (tools/ubench/pair_split.hip, arm basePF: 3 - 10 %
  // of a conv); with four trajectories per workgroup the 48 ring registers would have to live through the epilogue of a 32-register tile.
#ifdef MMD_NO_PF
  constexpr bool PF = false;                                  // (A/B side build: profiles/r06_prefetch_ab.txt)
#else
  constexpr bool PF = NS <= 2;
#endif
  auto prefetch = [&](const uint4* w) {
    if constexpr (PF) {
      const u32x4* wp[2] = {wptr(w, G128::FRAGS5, 0), wptr(w, G128::FRAGS5, 1)};
      rd_ring_load<G128, 2, RDD>(ring, wp);
    }
  };
  auto conv = [&](const uint4* w, const uint4* w_next) {
    const u32x4* wp[2] = {wptr(w, G128::FRAGS5, 0), wptr(w, G128::FRAGS5, 1)};
    TR(trb + 10);
    if constexpr (!PF) rd_ring_load<G128, 2, RDD>(ring, wp);
    rd_store2<G128>(vs, acc);
    TR(trb + 11);
    __syncthreads();
    TR(trb + 12);
    rd_taps<G128, 2, 0, 5, true, false, NS, RDD>(acc, acc, va128, wp, wp, ring);
    if (w_next) prefetch(w_next);
    TR(trb + 13);
  };

  // =================== RTB 0 (64 -> 128): conv A + the 1x1 residual conv from the row-form x slab ===================
  float inv_in[NS];
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) inv_in[sm] = dyn_scale(mx_read(mx, sm)).inv;
  rd_zero_halo<G64>(slab);
  const Epi<2> e0a = epi(a.r0.ba, a.r0.ga, a.r0.bea, a.r0.tb + tb_off, a.r0.isa);
  const float br[2] = {a.br[c0], a.br[c0 + 1]}, isr[2] = {a.isr[c0], a.isr[c0 + 1]};
  rowform_to_rd<CF::C0P, CF::XSS, CF::XSTR, NS>(lds, slab, mx);
  __syncthreads();
  {
    f32x4 res[NS][2];
    rd_taps<G64, 2, 0, 5, true, true, NS, 2>(acc, res, va64, wpa, wpr, reinterpret_cast<u32x4(&)[2][2][2]>(ring));
#pragma unroll
    for (int sm = 0; sm < NS; ++sm)
#pragma unroll
      for (int t = 0; t < 2; ++t) res[sm][t] = res[sm][t] * (isr[t] * inv_in[sm]) + br[t];
    park_tile(lds, res);                                     // (every wave is past the barrier behind the x slab's last read)
  }
  prefetch(a.r0.wb_bf);                                      // (the 64 -> 128 conv's depth-2 ring is consumed: RTB 0's conv B travels now)
  gn(std::true_type{}, e0a, inv_in, a.r0.act_a);
  TR(trb + 1);
  __syncthreads();                                           // conv A is done reading the 64-channel slab
  rd_zero_halo<G128>(slab);
  {
    const Epi<2> e = epi(a.r0.bb, a.r0.gb, a.r0.beb, nullptr, a.r0.isb);
    conv(a.r0.wb_bf, CF::N_IDENT > 0 ? a.ri[0].wa_bf : nullptr);
    gn(std::false_type{}, e, one4, 1.f);
  }

  // =================== identity RTBs ===================
#pragma unroll 1
  for (int k = 0; k < CF::N_IDENT; ++k) {
    const RtbPtrs& R = a.ri[k];
    park_tile(lds, acc);                                     // the block's input = its residual
    rd_dyn_out<2>(acc, mx, wave, lane, k == CF::MID_AFTER ? 1 : 0);   // (the input of the RTB after MID_AFTER is the skip tensor)
    __syncthreads();                                         // the previous conv is done reading the slab
    float inv[NS];
#pragma unroll
    for (int sm = 0; sm < NS; ++sm) {
      const DynScale ds = dyn_scale(mx_read(mx, sm));
      inv[sm] = ds.inv;
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[sm][t] *= ds.s;
    }
    const Epi<2> ea = epi(R.ba, R.ga, R.bea, R.tb + tb_off, R.isa);
    conv(R.wa_bf, R.wb_bf);
    gn(std::true_type{}, ea, inv, R.act_a);
    __syncthreads();
    const Epi<2> eb = epi(R.bb, R.gb, R.beb, nullptr, R.isb);
    conv(R.wb_bf, k + 1 < CF::N_IDENT ? a.ri[k + 1].wa_bf : nullptr);
    gn(std::false_type{}, eb, one4, 1.f);
    TR(trb + 18);
    if (CF::MID_AFTER == k + 1) {
#pragma unroll
      for (int sm = 0; sm < NS; ++sm)
#pragma unroll
        for (int t = 0; t < 2; ++t) mid[sm][t] = acc[sm][t];
    }
  }
  rd_dyn_out<2>(acc, mx, wave, lane, 2);                     // the stage's output: ups.0's conv A takes its maximum from region 2
}

// ups.0 in the direct form: cat(mid output, skip2) -> RTB (256 -> 64, with its 1x1 residual conv) -> RTB (64 -> 64) ->
// Upsample1d = ConvTranspose1d(k4, s2, p1) as two 2-tap parity passes, all f16x2 on Rd slabs.  Wave w owns the n-tile of
// channels 16 w + (lane & 15) x all four samples.  x0 / x1: the two 128-channel chunks of the input (downs.2's tiles, in
// ITS layout: store2(tile) writes one into the 128-channel slab); xe / xo: the stage's output (even / odd positions).
template <class CF, int NS, class STORE2>
__device__ __forceinline__ void chain_body_u0d(const ChainArgs& a, float* lds, int lane, int wave, f32x4 (&x0)[NS][2],
                                               f32x4 (&x1)[NS][2], STORE2 store2, char* slab128, f32x4 (&xe)[NS][1],
                                               f32x4 (&xo)[NS][1], int trb, int tb_off = 0) {
  static_assert(CF::L == 16 && CF::CM == 64 && CF::C0 == 128 && CF::C1 == 128 && CF::RES0 == RES_CONV && CF::TAIL == TAIL_UP &&
                    CF::N_IDENT == 1, "ups.0");
  using G128 = RdGeo<128>;
  using G64 = RdGeo<64>;
  const int n = lane & 15, g = lane >> 4, col = 16 * wave + n;
  char* const slab64 = reinterpret_cast<char*>(lds);
  static_assert(G64::BYTES <= MX_OFF * 4, "64-channel Rd slab");
  float* const mx = lds + MX_OFF;
  const char* const va128 = slab128 + g * G128::G + n * 16;
  const char* const va64 = slab64 + g * G64::G + n * 16;
  char* const vs64 = slab64 + wave * G64::G + (n >> 3) * G64::BX + (2 + 4 * g) * 16 + ((n & 7) >> 1) * 4;
  auto wptr = [&](const uint4* w, int frags) { return reinterpret_cast<const u32x4*>(w) + (size_t)wave * frags * 64 + lane; };
  constexpr int RDU = MMD_U0_RD;                               // weight ring depth of conv A's two 128-channel chunks
  u32x4 ringa[RDU][1][2];
  constexpr int RDC = MMD_U0C_RD;                              // ... of the 64 -> 64 convs and the tail's parity passes
  u32x4 ring[RDC][1][2];
  const u32x4* wp0[1] = {wptr(a.r0.wa_bf, G128::FRAGS5)};
  const u32x4* wp1[1] = {wptr(a.wa0_c1_bf, G128::FRAGS5)};
  const u32x4* wr0[1] = {wptr(a.wres_bf, 2 * G128::KC)};
  const u32x4* wr1[1] = {wptr(a.wres_c1_bf, 2 * G128::KC)};
  rd_ring_load<G128, 1, RDU>(ringa, wp0);
  __syncthreads();                                           // the previous stage is done with the slab; its maxima are in mx
  TR(trb + 0);

  f32x4 acc[NS][1], res[NS][1];
  float one4[NS];
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) one4[sm] = 1.f;
  auto epi = [&](const float* b, const float* gm, const float* be, const float* tb, const float* isc) {
    return epi_load<1>(b, gm, be, tb, isc, col);
  };
  auto gn = [&](auto conv_a, const Epi<1>& e, const float (&inv)[NS], float act_s) {
    if constexpr (decltype(conv_a)::value) {
      const float t0 = e.tb[0] * act_s;
      rd_gn_mish<1, 128, true>(acc, e.b, e.g, e.be, e.is, inv, act_scale(act_s), [&](int, int, int) { return t0; });
    } else {
      rd_gn_mish<1, 128, false>(acc, e.b, e.g, e.be, e.is, inv, ActScale{}, [&](int sm, int, int r) { return res[sm][0][r]; });
    }
  };
  // dynamic input scale of a conv on the tile in acc: (maxima -> mx, barrier, then) scale in place; the inverse scales
  auto dyn_scale_acc = [&](float (&inv)[NS]) {
#pragma unroll
    for (int sm = 0; sm < NS; ++sm) {
      const DynScale ds = dyn_scale(mx_read(mx, sm));
      inv[sm] = ds.inv;
      acc[sm][0] *= ds.s;
    }
  };
  // one 64 -> 64 conv over the tile in acc (already scaled); on entry every wave is past its reads of the slab.  PF (two trajectories
  // per workgroup = unet_kernel<2>): the NEXT pack's first ring steps are requested right behind this conv's taps and travel during the
  // epilogue (chain_body_d2d has the measurement); `frags_next`: the next pack's fragments per n-tile.
#ifdef MMD_NO_PF
  constexpr bool PF = false;
#else
  constexpr bool PF = NS <= 2;
#endif
  auto prefetch64 = [&](const uint4* w, int frags) {
    if constexpr (PF) {
      const u32x4* wp[1] = {wptr(w, frags)};
      rd_ring_load<G64, 1, RDC>(ring, wp);
    }
  };
  auto conv64 = [&](const uint4* w, const uint4* w_next, int frags_next) {
    const u32x4* wp[1] = {wptr(w, G64::FRAGS5)};
    if constexpr (!PF) rd_ring_load<G64, 1, RDC>(ring, wp);
    rd_store1<G64>(vs64, acc, lane);
    __syncthreads();
    rd_taps<G64, 1, 0, 5, true, false, NS, RDC>(acc, res, va64, wp, wp, ring);
    prefetch64(w_next, frags_next);
  };

  // =================== RTB 0: cat(x0, x1) -> 64 channels; the 1x1 residual conv rides on the centre tap ===================
  float inv_in[NS];
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) {
    // residual-stream input: dynamic scale from the maxima downs.2 left in regions 1 (skip2) and 2 (mid output) of mx
    const DynScale ds = dyn_scale(fmaxf(mx_read(mx + MX_REGION, sm), mx_read(mx + 2 * MX_REGION, sm)));
    inv_in[sm] = ds.inv;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      x0[sm][t] *= ds.s;
      x1[sm][t] *= ds.s;
    }
  }
  const Epi<1> e0a = epi(a.r0.ba, a.r0.ga, a.r0.bea, a.r0.tb + tb_off, a.r0.isa);
  const float br = a.br[col], isr = a.isr[col];
  store2(x0);
  TR(160);
  __syncthreads();
  TR(161);
  rd_taps<G128, 1, 0, 5, true, true, NS, RDU>(acc, res, va128, wp0, wr0, ringa);
  rd_ring_load<G128, 1, RDU>(ringa, wp1);
  TR(162);
  __syncthreads();                                           // every wave is done reading chunk 0
  store2(x1);
  TR(163);
  __syncthreads();
  TR(164);
  rd_taps<G128, 1, 0, 5, false, true, NS, RDU>(acc, res, va128, wp1, wr1, ringa);
  TR(165);
#pragma unroll
  for (int sm = 0; sm < NS; ++sm) res[sm][0] = res[sm][0] * (isr * inv_in[sm]) + br;
  prefetch64(a.r0.wb_bf, G64::FRAGS5);
  gn(std::true_type{}, e0a, inv_in, a.r0.act_a);
  TR(trb + 1);
  __syncthreads();                                           // chunk 1 is consumed
  rd_zero_halo<G64>(slab64);
  {
    const Epi<1> e = epi(a.r0.bb, a.r0.gb, a.r0.beb, nullptr, a.r0.isb);
    conv64(a.r0.wb_bf, a.ri[0].wa_bf, G64::FRAGS5);
    gn(std::false_type{}, e, one4, 1.f);
  }
  TR(trb + 4);
  // =================== identity RTB ===================
  {
    const RtbPtrs& R = a.ri[0];
#pragma unroll
    for (int sm = 0; sm < NS; ++sm) res[sm][0] = acc[sm][0];
    rd_dyn_out<1>(acc, mx, wave, lane, 0);
    __syncthreads();                                         // the previous conv is done reading the slab
    float inv[NS];
    dyn_scale_acc(inv);
    const Epi<1> ea = epi(R.ba, R.ga, R.bea, R.tb + tb_off, R.isa);
    conv64(R.wa_bf, R.wb_bf, G64::FRAGS5);
    gn(std::true_type{}, ea, inv, R.act_a);
    TR(trb + 5);
    __syncthreads();
    const Epi<1> eb = epi(R.bb, R.gb, R.beb, nullptr, R.isb);
    conv64(R.wb_bf, a.wt_bf0, 2 * G64::KC * 2);              // (next: the transposed tail's first parity pack)
    gn(std::false_type{}, eb, one4, 1.f);
    TR(trb + 6);
  }
  // =================== tail: out[2 m] = in[m - 1] W3 + in[m] W1, out[2 m + 1] = in[m] W2 + in[m + 1] W0 ===================
  {
    rd_dyn_out<1>(acc, mx, wave, lane, 0);
    __syncthreads();
    float inv[NS];
    dyn_scale_acc(inv);
    const u32x4* wt0[1] = {wptr(a.wt_bf0, 2 * G64::KC * 2)};
    const u32x4* wt1[1] = {wptr(a.wt_bf1, 2 * G64::KC * 2)};
    const float bt = a.bt[col], is0 = a.ist0[col], is1 = a.ist1[col];
    if constexpr (!PF) rd_ring_load<G64, 1, RDC>(ring, wt0);
    rd_store1<G64>(vs64, acc, lane);
    __syncthreads();
    TR(trb + 7);
    rd_taps<G64, 1, 1, 2, true, false, NS, RDC>(xe, res, va64, wt0, wt0, ring);
    rd_ring_load<G64, 1, RDC>(ring, wt1);
    rd_taps<G64, 1, 2, 2, true, false, NS, RDC>(xo, res, va64, wt1, wt1, ring);
    // the stage's output stays in registers: xe / xo[sample][0][r] = positions 2 m, 2 m + 1 (m = 4 g + r) of channel col; the
    // per-sample maxima of the wave's 16 channels go to slot `wave` of mx region 0 (the caller's barrier publishes them)
#pragma unroll
    for (int sm = 0; sm < NS; ++sm) {
      float m = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xe[sm][0][r] = fmaf(xe[sm][0][r], is0 * inv[sm], bt);
        xo[sm][0][r] = fmaf(xo[sm][0][r], is1 * inv[sm], bt);
        m = fmaxf(m, fmaxf(fabsf(xe[sm][0][r]), fabsf(xo[sm][0][r])));
      }
      m = row_max16(m);
      m = max_xor16(m);
      m = max_xor32(m);
      if (lane == 0) mx[sm * MX_SLOTS + wave] = m;
    }
  }
}

// ups.1 (cat(x, skip1): 128 -> 32 -> 32 channels at L = 32, Upsample1d) + the final block (Conv1dBlock 32 -> 32 at L = 64, 1x1
// conv 32 -> 4), wave = sample, all convs direct f16x2 (layers.py:346-358, temporal_unet.py:104-110, 166-172).  Only the
// first conv needs the other waves: its input arrives distributed by CHANNEL (ups.0's output xe / xo and the skip tensor kept
// from downs.1: wave w holds channels 16 w + (lane & 15) of all four samples), so the two 64-channel chunks are written into
// the four samples' slabs across waves, one after the other through the same 10 KB slab (4 workgroup barriers); everything
// after it -- 3 convs, the transposed tail as two parity passes, the final block and the output store -- reads only what the
// same wave wrote (wave_lds_fence).  Slabs: RwGeo<64, 32> (conv A chunks), RwGeo<32, 32>, RwGeo<32, 64> (final block).
template <class CF, int NS>
__device__ __forceinline__ void chain_body_u1w(const ChainArgs& a, const FinalArgs& f, const FusedStep& fs, float* lds, int n0, int lane_in, int wave,
                                               const f32x4 (&xe)[NS][1], const f32x4 (&xo)[NS][1], const f32x4 (&skip)[NS][2],
                                               int trb, int tb_off = 0) {
  // (an opaque copy of the lane index: the stage's lane-derived offsets are recomputed here -- a handful of VALU ops -- instead
  // of being kept alive, i.e. spilled, since the stages that happen to use the same products)
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  static_assert(CF::L == 32 && CF::CM == 32 && CF::C0 == 64 && CF::C1 == 64 && CF::RES0 == RES_CONV && CF::N_IDENT == 1 &&
                    CF::TAIL == TAIL_UP, "ups.1");
  using GA = RwGeo<64, 32>;
  using GB = RwGeo<32, 32>;
  using GF = RwGeo<32, 64>;
  constexpr int W_BYTES = cmax(GA::BYTES, cmax(GB::BYTES, GF::BYTES)) + 128;
  static_assert(4 * W_BYTES <= MX_OFF * 4, "four private slabs");
  char* const lb = reinterpret_cast<char*>(lds);
  char* const slab = lb + wave * W_BYTES;
  float* const mx = lds + MX_OFF;
  const int n = lane & 15, g = lane >> 4, c0 = 2 * n;
  const bool odd = n & 1;
  auto wptr = [&](const uint4* w, int frags, int t) { return reinterpret_cast<const u32x4*>(w) + (size_t)t * frags * 64 + lane; };
  auto swap1 = [](float v) {                                  // the value of the partner lane (n ^ 1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  };
  const u32x4* wp0[2] = {wptr(a.r0.wa_bf, GA::FRAGS5, 0), wptr(a.r0.wa_bf, GA::FRAGS5, 1)};
  const u32x4* wp1[2] = {wptr(a.wa0_c1_bf, GA::FRAGS5, 0), wptr(a.wa0_c1_bf, GA::FRAGS5, 1)};
  const u32x4* wr0[2] = {wptr(a.wres_bf, 2 * GA::KC, 0), wptr(a.wres_bf, 2 * GA::KC, 1)};
  const u32x4* wr1[2] = {wptr(a.wres_c1_bf, 2 * GA::KC, 0), wptr(a.wres_c1_bf, 2 * GA::KC, 1)};
  constexpr int RDA = 5;                                      // conv A's weight ring: half a chunk ahead
  u32x4 ring[RDA][2][2];
  rd_ring_load<GA, 2, RDA>(ring, wp0);
  // ---- the skip tensor's per-sample maxima (downs.1's layout: wave = (channel half np, sample pair sp); skip[m][t][r]: sample 2 sp
  //      + (m >> 1), channel 32 np + 2 n + t, position 16 (m & 1) + 4 g + r) -> slots 4 .. 7 of mx region 0 (the two waves of a
  //      sample pair fill them); ups.0 left its output's maxima in slots 0 .. 3
  //      (two trajectories per workgroup: sp = the wave's one sample, NS = 2 M tiles)
  constexpr int SW = NS / 2;
  const int np = wave & 1, sp = wave >> 1, s0 = SW * sp;
  {
    float m2[2] = {0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < SW; ++sl) {
      float m = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(skip[2 * sl + mt][t][r]));
      m = row_max16(m);
      m = max_xor16(m);
      m2[sl] = max_xor32(m);
    }
    if (lane < 2 * SW) mx[(s0 + (lane >> 1)) * MX_SLOTS + 4 + np + 2 * (lane & 1)] = (lane >> 1) ? m2[1] : m2[0];
  }
  __syncthreads();                                           // ups.0 is done with its slabs; the maxima are in mx
  TR(trb + 0);
  float sc[4];
#pragma unroll
  for (int sm = 0; sm < 4; ++sm) sc[sm] = dyn_scale(mx_read(mx, sm)).s;
  const float inv_in = dyn_scale(mx_read(mx, wave)).inv;
  const float sc_lo = dyn_scale(mx_read(mx, s0)).s, sc_hi = dyn_scale(mx_read(mx, s0 + SW - 1)).s;
  // channel col = 16 wave + n of a 64-channel chunk: block 2 wave + (n >> 3) = (lane group wave, chunk n >> 3), the pair (n & ~1,
  // n | 1) one dword; the lanes of a pair swap halves so that each stores whole dwords
  char* const cdst = lb + wave * GA::G + (n >> 3) * GA::BX + ((n & 7) >> 1) * 4 + 2 * 16;
  {
    // zero halo rows 0, 1, 34, 35 of the own slab's 8 blocks x 2 pieces
    const int hr = lane & 3;
    *reinterpret_cast<uint4*>(slab + (lane >> 5) * GA::PS + ((lane >> 3) & 3) * GA::G + ((lane >> 2) & 1) * GA::BX +
                              (hr < 2 ? hr : 32 + hr) * 16) = make_uint4(0u, 0u, 0u, 0u);
    // chunk 0 = ups.0's output: the even lane stores the even positions 2 (4 g + r) of channels (col, col + 1), the odd lane
    // the odd positions of (col - 1, col)
    char* const d0 = cdst + (8 * g + (odd ? 1 : 0)) * 16;
#pragma unroll
    for (int sm = 0; sm < NS; ++sm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float own = (odd ? xo[sm][0][r] : xe[sm][0][r]) * sc[sm];
        const float recv = swap1((odd ? xe[sm][0][r] : xo[sm][0][r]) * sc[sm]);
        const F16Pair p = f16_split2(odd ? recv : own, odd ? own : recv);
        char* d = d0 + sm * W_BYTES + 2 * r * 16;
        *reinterpret_cast<unsigned*>(d) = p.hi;
        *reinterpret_cast<unsigned*>(d + GA::PS) = p.lo;
      }
  }
  TR(trb + 6);
  __syncthreads();
  TR(trb + 7);
  const char* const vaA = slab + g * GA::G + n * 16;
  f32x4 acc[2][2], res[2][2];
  rd_taps<GA, 2, 0, 5, true, true, 2, RDA>(acc, res, vaA, wp0, wr0, ring);
  rd_ring_load<GA, 2, RDA>(ring, wp1);
  TR(trb + 8);
  __syncthreads();                                           // every wave has consumed chunk 0
  {
    // chunk 1 = skip: the lane's channel pair (32 np + 2 n, + 1) of the 64-channel chunk = block 4 np + (n >> 2) = (chunk (n >> 2) &
    // 1, lane group 2 np + (n >> 3)), dword n & 3, in the slabs of samples 2 sp, 2 sp + 1
    char* const sdst = lb + (2 * np + (n >> 3)) * GA::G + ((n >> 2) & 1) * GA::BX + (n & 3) * 4 + (2 + 4 * g) * 16;
#pragma unroll
    for (int m = 0; m < NS; ++m) {
      const float sm_s = (m >> 1) ? sc_hi : sc_lo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const F16Pair p = f16_split2(skip[m][0][r] * sm_s, skip[m][1][r] * sm_s);
        char* d = sdst + (s0 + (m >> 1)) * W_BYTES + (16 * (m & 1) + r) * 16;
        *reinterpret_cast<unsigned*>(d) = p.hi;
        *reinterpret_cast<unsigned*>(d + GA::PS) = p.lo;
      }
    }
  }
  __syncthreads();
  TR(trb + 10);
  const Epi<2> e0a = epi_load<2>(a.r0.ba, a.r0.ga, a.r0.bea, a.r0.tb + tb_off, a.r0.isa, c0);
  const float br[2] = {a.br[c0], a.br[c0 + 1]}, isr[2] = {a.isr[c0], a.isr[c0 + 1]};
  rd_taps<GA, 2, 0, 5, false, true, 2, RDA>(acc, res, vaA, wp1, wr1, ring);
  TR(trb + 11);
  // ---- from here on the wave is on its own: 32-channel slab
  const char* const vaB = slab + g * GB::G + n * 16;
  char* const vsB = slab + (n >> 2) * GB::G + (2 + 4 * g) * 16 + (n & 3) * 4;
  u32x4 ring5[5][2][2];
  auto preload = [&](const uint4* w) {
    const u32x4* wp[2] = {wptr(w, GB::FRAGS5, 0), wptr(w, GB::FRAGS5, 1)};
    rd_ring_load<GB, 2, 5>(ring5, wp);
  };
  preload(a.r0.wb_bf);
  const float one = 1.f;
  auto epi = [&](const float* bs, const float* gm, const float* be, const float* tb, const float* isc) {
    return epi_load<2>(bs, gm, be, tb, isc, c0);
  };
  auto gn = [&](auto conv_a, const Epi<2>& e, float inv, float act_s) {
    if constexpr (decltype(conv_a)::value) {
      const float t0 = e.tb[0] * act_s, t1 = e.tb[1] * act_s;
      rw_gn_mish_whole<2, 2, 2, 128, true>(acc, e.b, e.g, e.be, e.is, inv, act_scale(act_s), [&](int, int t, int) { return t ? t1 : t0; });
    } else {
      rw_gn_mish_whole<2, 2, 2, 128, false>(acc, e.b, e.g, e.be, e.is, inv, ActScale{}, [&](int mt, int t, int r) { return res[mt][t][r]; });
    }
  };
  auto conv = [&](const uint4* w) {                          // one 32 -> 32 conv over the tile in acc (already scaled)
    const u32x4* wp[2] = {wptr(w, GB::FRAGS5, 0), wptr(w, GB::FRAGS5, 1)};
    rw_store2<GB, 2>(vsB, acc);
    wave_lds_fence();
    rd_taps<GB, 2, 0, 5, true, false, 2, 5>(acc, res, vaB, wp, wp, ring5);
    wave_lds_fence();                                        // (the next store must not overtake these reads)
  };
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < 2; ++t) res[mt][t] = res[mt][t] * (isr[t] * inv_in) + br[t];
  gn(std::true_type{}, e0a, inv_in, a.r0.act_a);
  TR(trb + 1);
  wave_lds_fence();                                          // conv A's reads are done: the slab changes its geometry
  if (lane < 32) *reinterpret_cast<uint4*>(slab + (lane >> 4) * GB::PS + ((lane >> 2) & 3) * GB::G + ((lane & 3) < 2 ? (lane & 3) : 32 + (lane & 3)) * 16) = make_uint4(0u, 0u, 0u, 0u);
  {
    const Epi<2> e = epi(a.r0.bb, a.r0.gb, a.r0.beb, nullptr, a.r0.isb);
    conv(a.r0.wb_bf);
    preload(a.ri[0].wa_bf);
    gn(std::false_type{}, e, one, 1.f);
  }
  TR(trb + 2);
  // ---- identity RTB
  {
    const RtbPtrs& R = a.ri[0];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) res[mt][t] = acc[mt][t];
    const DynScale ds = dyn_scale(rw_absmax<2, 2>(acc));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[mt][t] *= ds.s;
    const Epi<2> ea = epi(R.ba, R.ga, R.bea, R.tb + tb_off, R.isa);
    conv(R.wa_bf);
    preload(R.wb_bf);
    gn(std::true_type{}, ea, ds.inv, R.act_a);
    TR(trb + 3);
    const Epi<2> eb = epi(R.bb, R.gb, R.beb, nullptr, R.isb);
    conv(R.wb_bf);
    gn(std::false_type{}, eb, one, 1.f);
    TR(trb + 4);
  }
  // ---- tail: Upsample1d = ConvTranspose1d(k4, s2, p1): out[2 m] = in[m - 1] W3 + in[m] W1, out[2 m + 1] = in[m] W2 + in[m + 1] W0
  //      -> the final block's input (L = 64) in the 64-row slab
  const char* const vaF = slab + g * GF::G + n * 16;
  char* const vsF = slab + (n >> 2) * GF::G + (n & 3) * 4;
  f32x4 y[4][2];
  float inv_f;
  {
    const DynScale ds = dyn_scale(rw_absmax<2, 2>(acc));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[mt][t] *= ds.s;
    const u32x4* wt0[2] = {wptr(a.wt_bf0, 2 * GB::KC * 2, 0), wptr(a.wt_bf0, 2 * GB::KC * 2, 1)};
    const u32x4* wt1[2] = {wptr(a.wt_bf1, 2 * GB::KC * 2, 0), wptr(a.wt_bf1, 2 * GB::KC * 2, 1)};
    u32x4 ring2[2][2][2];
    const float bt[2] = {a.bt[c0], a.bt[c0 + 1]};
    const float is0[2] = {a.ist0[c0] * ds.inv, a.ist0[c0 + 1] * ds.inv}, is1[2] = {a.ist1[c0] * ds.inv, a.ist1[c0 + 1] * ds.inv};
    rd_ring_load<GB, 2, 2>(ring2, wt0);
    rw_store2<GB, 2>(vsB, acc);
    wave_lds_fence();
    f32x4 e[2][2], o[2][2];
    rd_taps<GB, 2, 1, 2, true, false, 2, 2>(e, res, vaB, wt0, wt0, ring2);
    rd_ring_load<GB, 2, 2>(ring2, wt1);
    rd_taps<GB, 2, 2, 2, true, false, 2, 2>(o, res, vaB, wt1, wt1, ring2);
    float m = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          e[mt][t][r] = fmaf(e[mt][t][r], is0[t], bt[t]);
          o[mt][t][r] = fmaf(o[mt][t][r], is1[t], bt[t]);
          m = fmaxf(m, fmaxf(fabsf(e[mt][t][r]), fabsf(o[mt][t][r])));
        }
    m = row_max16(m);
    m = max_xor16(m);
    m = max_xor32(m);
    const DynScale df = dyn_scale(m);
    inv_f = df.inv;
    wave_lds_fence();                                        // the tail's reads are done: 64-row geometry
    if (lane < 32) *reinterpret_cast<uint4*>(slab + (lane >> 4) * GF::PS + ((lane >> 2) & 3) * GF::G + ((lane & 3) < 2 ? (lane & 3) : 64 + (lane & 3)) * 16) = make_uint4(0u, 0u, 0u, 0u);
    // positions 2 m + parity, m = 16 mt + 4 g + r: rows 2 + 32 mt + 8 g + 2 r + parity
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const F16Pair pe = f16_split2(e[mt][0][r] * df.s, e[mt][1][r] * df.s), po = f16_split2(o[mt][0][r] * df.s, o[mt][1][r] * df.s);
        char* d = vsF + (2 + 32 * mt + 8 * g + 2 * r) * 16;
        *reinterpret_cast<unsigned*>(d) = pe.hi;
        *reinterpret_cast<unsigned*>(d + GF::PS) = pe.lo;
        *reinterpret_cast<unsigned*>(d + 16) = po.hi;
        *reinterpret_cast<unsigned*>(d + 16 + GF::PS) = po.lo;
      }
  }
  TR(trb + 5);
  // ---- final block: Conv1dBlock(32 -> 32, k5) + GroupNorm + Mish, then the 1x1 conv 32 -> 4 (N padded to one n-tile)
  {
    const u32x4* wf[2] = {wptr(f.w5, GF::FRAGS5, 0), wptr(f.w5, GF::FRAGS5, 1)};
    rd_ring_load<GF, 2, 5>(ring5, wf);
    const u32x4* w1[1] = {reinterpret_cast<const u32x4*>(f.w1_bf) + lane};
    u32x4 ring1[1][1][2];
    rd_ring_load<GF, 1, 1>(ring1, w1);
    const Epi<2> ef = epi_load<2>(f.bias, f.gamma, f.beta, nullptr, f.isc, c0);
    const float b1 = f.w1_bias[n & 3], s1 = f.is1[n & 3];
    wave_lds_fence();
    rd_taps<GF, 2, 0, 5, true, false, 4, 5>(y, y, vaF, wf, wf, ring5);
    rw_gn_mish_whole<4, 2, 2, 256, true>(y, ef.b, ef.g, ef.be, ef.is, inv_f, act_scale(f.act), [](int, int, int) { return 0.f; });
    wave_lds_fence();
    rw_store2<GF, 4>(vsF + (2 + 4 * g) * 16, y);
    wave_lds_fence();
    f32x4 out[4][1];
    rd_taps<GF, 1, 2, 1, true, false, 4, 1>(out, out, vaF, w1, w1, ring1);
    if (fs.enabled) {
      // eps[64][4] -> the wave's slab as float4 rows, lane = support point: the unguided ddpm_sample_fn step (sample_functions.py:
      // 40-86; ddpm_guide_kernel's arithmetic, guide_dev.h) on the wave's trajectory, in place
      float* const et = reinterpret_cast<float*>(slab);
      wave_lds_fence();                                      // (the 1x1 conv's reads of the slab are done)
      if (n < 4) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) et[(16 * mt + 4 * g + r) * 4 + n] = fmaf(out[mt][0][r], s1, b1);
      }
      wave_lds_fence();
      if (wave < NS && n0 + wave < a.n) {
        const float4 e = *reinterpret_cast<const float4*>(et + lane * 4);
        const int traj = fs.traj0 + n0 + wave, robot = traj / fs.spr;
        const size_t idx = (size_t)traj * H + lane;
        float4 v = ddpm_posterior_mean(fs.x[idx], e, fs.a_t, fs.b_t, fs.c1, fs.c2);
        if (fs.do_noise)
          v = add_step_noise(v, fs.noise ? fs.noise[idx] : traj_normal4(fs.seed, fs.robot_seeds, fs.draw, fs.traj_base, idx, robot, fs.spr), fs.sigma,
                             fs.noise_std_extra);
        float4 hv;
        if (hard_row(fs.hard_rows, fs.n_hard, fs.hard, robot, lane, hv)) v = hv;
        fs.x[idx] = v;
        if (fs.chain) fs.chain[idx] = v;
      }
    } else if (n < 4 && wave < NS && n0 + wave < a.n) {
      float* dst = f.out + ((size_t)(n0 + wave) * 64 + 4 * g) * 4 + n;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(16 * mt + r) * 4] = fmaf(out[mt][0][r], s1, b1);
    }
  }
}

// ups.1 + final block for unet_kernel<2>: like chain_body_d0s a sample is split between two waves by position (wave = (sample sp =
// wave >> 1, half hf = wave & 1): ONE M tile of the L = 32 convs, two of the final block's L = 64), the sample's slab is shared,
// GroupNorm statistics / dynamic scales are exchanged through LDS.  The two input chunks arrive across waves as in
// chain_body_u1w.  Bitwise equal results.
template <class CF, int NV = 2>
__device__ __forceinline__ void chain_body_u1s(const ChainArgs& a, const FinalArgs& f, const FusedStep& fs, float* lds, int n0, int lane_in, int wave,
                                               const f32x4 (&xe)[2][1], const f32x4 (&xo)[2][1], const f32x4 (&skip)[2][2], int trb, int tb_off = 0) {
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  static_assert(CF::L == 32 && CF::CM == 32 && CF::C0 == 64 && CF::C1 == 64 && CF::RES0 == RES_CONV && CF::N_IDENT == 1 &&
                    CF::TAIL == TAIL_UP, "ups.1");
  using GA = RwGeo<64, 32>;
  using GB = RwGeo<32, 32>;
  using GF = RwGeo<32, 64>;
  constexpr int W_BYTES = cmax(GA::BYTES, cmax(GB::BYTES, GF::BYTES)) + 128;   // (the geometry of chain_body_u1w: per SAMPLE here)
  char* const lb = reinterpret_cast<char*>(lds);
  const int sp = wave >> 1, hf = wave & 1;                   // (downs.1's skip layout has the same sample index: np = wave & 1 there)
  char* const slab = lb + sp * W_BYTES;
  float* const mx = lds + MX_OFF;
  HalfStat* const xch = reinterpret_cast<HalfStat*>(lds + PARK2_OFF) + sp * 128;
  const int n = lane & 15, g = lane >> 4, c0 = 2 * n;
  const bool odd = n & 1;
  auto wptr = [&](const uint4* w, int frags, int t) { return reinterpret_cast<const u32x4*>(w) + (size_t)t * frags * 64 + lane; };
  auto swap1 = [](float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  };
  auto sample_max = [&](float own) {
    if (lane < 4) mx[sp * MX_SLOTS + 4 * hf + lane] = own;
    __syncthreads();
    return mx_read(mx, sp);
  };
  const u32x4* wp0[2] = {wptr(a.r0.wa_bf, GA::FRAGS5, 0), wptr(a.r0.wa_bf, GA::FRAGS5, 1)};
  const u32x4* wp1[2] = {wptr(a.wa0_c1_bf, GA::FRAGS5, 0), wptr(a.wa0_c1_bf, GA::FRAGS5, 1)};
  const u32x4* wr0[2] = {wptr(a.wres_bf, 2 * GA::KC, 0), wptr(a.wres_bf, 2 * GA::KC, 1)};
  const u32x4* wr1[2] = {wptr(a.wres_c1_bf, 2 * GA::KC, 0), wptr(a.wres_c1_bf, 2 * GA::KC, 1)};
  constexpr int RDA = 5;
  u32x4 ring[RDA][2][2];
  rd_ring_load<GA, 2, RDA>(ring, wp0);
  // ---- the skip tensor's per-sample maxima (downs.1's layout for two trajectories: wave = (channel half np = wave & 1, sample wave >> 1),
  //      skip[m][t][r]: M tile m of the sample) -> slots 4 .. 7 of mx region 0; ups.0 left its output's maxima in slots 0 .. 3
  {
    float m = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(skip[mt][t][r]));
    m = row_max16(m);
    m = max_xor16(m);
    m = max_xor32(m);
    if (lane < 2) mx[sp * MX_SLOTS + 4 + hf + 2 * lane] = m;
  }
  __syncthreads();                                           // ups.0 is done with its slabs; the maxima are in mx
  TR(trb + 0);
  // staged weights (stage_weights above): two buffers behind the two samples' slabs; RTB 0's conv B lands while conv A runs
  char* const wb0 = lb + 2 * W_BYTES;
  char* const wb1 = wb0 + WBUF_BYTES;
  static_assert(2 * W_BYTES + 2 * WBUF_BYTES <= MX_OFF * 4, "slabs + weight buffers below the maxima");
  static_assert(GB::FRAGS5 * 2 * 1024 <= WBUF_BYTES && GF::FRAGS5 * 2 * 1024 <= WBUF_BYTES, "a staged conv fits its buffer");
  stage_weights<2 * GB::FRAGS5>(a.r0.wb_bf, wb0, wave, lane);
  float sc[2];
#pragma unroll
  for (int sm = 0; sm < 2; ++sm) sc[sm] = dyn_scale(mx_read(mx, sm)).s;
  const float inv_in = dyn_scale(mx_read(mx, sp)).inv;
  const float sc_own = dyn_scale(mx_read(mx, sp)).s;
  char* const cdst = lb + wave * GA::G + (n >> 3) * GA::BX + ((n & 7) >> 1) * 4 + 2 * 16;
  {
    // zero halo rows 0, 1, 34, 35 of the sample's slab (8 blocks x 2 pieces): one wave per sample
    if (hf == 0) {
      const int hr = lane & 3;
      *reinterpret_cast<uint4*>(slab + (lane >> 5) * GA::PS + ((lane >> 3) & 3) * GA::G + ((lane >> 2) & 1) * GA::BX +
                                (hr < 2 ? hr : 32 + hr) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    // chunk 0 = ups.0's output (wave w holds channels 16 w + n of both samples): as in chain_body_u1w
    char* const d0 = cdst + (8 * g + (odd ? 1 : 0)) * 16;
#pragma unroll
    for (int sm = 0; sm < 2; ++sm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float own = (odd ? xo[sm][0][r] : xe[sm][0][r]) * sc[sm];
        const float recv = swap1((odd ? xe[sm][0][r] : xo[sm][0][r]) * sc[sm]);
        const F16Pair p = f16_split2(odd ? recv : own, odd ? own : recv);
        char* d = d0 + sm * W_BYTES + 2 * r * 16;
        *reinterpret_cast<unsigned*>(d) = p.hi;
        *reinterpret_cast<unsigned*>(d + GA::PS) = p.lo;
      }
  }
  TR(trb + 6);
  __syncthreads();
  TR(trb + 7);
  const char* const vaA = slab + g * GA::G + (n + 16 * hf) * 16;
  f32x4 acc[1][2], res[1][2];
  rd_taps<GA, 2, 0, 5, true, true, 1, RDA>(acc, res, vaA, wp0, wr0, ring);
  rd_ring_load<GA, 2, RDA>(ring, wp1);
  TR(trb + 8);
  __syncthreads();                                           // every wave has consumed chunk 0
  {
    // chunk 1 = skip (downs.1's layout: this wave holds channel pair (32 np + 2 n, + 1), np = wave & 1, of sample sp, both M tiles)
    const int np = hf;
    char* const sdst = lb + (2 * np + (n >> 3)) * GA::G + ((n >> 2) & 1) * GA::BX + (n & 3) * 4 + (2 + 4 * g) * 16 + sp * W_BYTES;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const F16Pair p = f16_split2(skip[m][0][r] * sc_own, skip[m][1][r] * sc_own);
        char* d = sdst + (16 * m + r) * 16;
        *reinterpret_cast<unsigned*>(d) = p.hi;
        *reinterpret_cast<unsigned*>(d + GA::PS) = p.lo;
      }
  }
  __syncthreads();
  TR(trb + 10);
  const Epi<2> e0a = epi_load<2>(a.r0.ba, a.r0.ga, a.r0.bea, a.r0.tb + tb_off, a.r0.isa, c0);
  const float br[2] = {a.br[c0], a.br[c0 + 1]}, isr[2] = {a.isr[c0], a.isr[c0 + 1]};
  rd_taps<GA, 2, 0, 5, false, true, 1, RDA>(acc, res, vaA, wp1, wr1, ring);
  TR(trb + 11);
  // ---- 32-channel slab of the sample, shared by its two waves
  const char* const vaB = slab + g * GB::G + (n + 16 * hf) * 16;
  char* const vsB = slab + (n >> 2) * GB::G + (2 + 4 * g + 16 * hf) * 16 + (n & 3) * 4;
  u32x4 ring2[2][2][2];
  const float one = 1.f;
  auto epi = [&](const float* bs, const float* gm, const float* be, const float* tb, const float* isc) {
    return epi_load<2>(bs, gm, be, tb, isc, c0);
  };
  auto gn = [&](auto conv_a, const Epi<2>& e, float inv, float act_s) {
    if constexpr (decltype(conv_a)::value) {
      const float t0 = e.tb[0] * act_s, t1 = e.tb[1] * act_s;
      rw_gn_mish_half<1, 2, 2, 128, true>(acc, e.b, e.g, e.be, e.is, inv, act_scale(act_s), [&](int, int t, int) { return t ? t1 : t0; }, xch, hf, lane);
    } else {
      rw_gn_mish_half<1, 2, 2, 128, false>(acc, e.b, e.g, e.be, e.is, inv, ActScale{}, [&](int mt, int t, int r) { return res[mt][t][r]; }, xch, hf, lane);
    }
  };
  // one 32 -> 32 conv over the half tile in acc (already scaled), its weights staged in wb; `stage_next` puts the next conv's on
  // their way behind the barrier (every wave is past the conv that read the other buffer)
  auto conv = [&](char* wb, auto stage_next) {
    const u32x4* wp[2] = {reinterpret_cast<const u32x4*>(wb) + lane, reinterpret_cast<const u32x4*>(wb) + GB::FRAGS5 * 64 + lane};
    rw_store2<GB, 1>(vsB, acc);
    staged_weights_landed();
    __syncthreads();
    stage_next();
    rd_ring_load<GB, 2, 2>(ring2, wp);
    rd_taps<GB, 2, 0, 5, true, false, 1, 2>(acc, res, vaB, wp, wp, ring2);
  };
  constexpr int TF = 2 * (2 * GB::KC * 2);                   // fragments of one parity pass of the transposed tail (both n-tiles)
#pragma unroll
  for (int t = 0; t < 2; ++t) res[0][t] = res[0][t] * (isr[t] * inv_in) + br[t];
  gn(std::true_type{}, e0a, inv_in, a.r0.act_a);             // (its barrier: conv A's reads are done, the slab changes its geometry)
  TR(trb + 1);
  if (hf == 0 && lane < 32) *reinterpret_cast<uint4*>(slab + (lane >> 4) * GB::PS + ((lane >> 2) & 3) * GB::G + ((lane & 3) < 2 ? (lane & 3) : 32 + (lane & 3)) * 16) = make_uint4(0u, 0u, 0u, 0u);
  {
    const Epi<2> e = epi(a.r0.bb, a.r0.gb, a.r0.beb, nullptr, a.r0.isb);
    conv(wb0, [&] { stage_weights<2 * GB::FRAGS5>(a.ri[0].wa_bf, wb1, wave, lane); });
    gn(std::false_type{}, e, one, 1.f);
  }
  TR(trb + 2);
  // ---- identity RTB
  {
    const RtbPtrs& R = a.ri[0];
#pragma unroll
    for (int t = 0; t < 2; ++t) res[0][t] = acc[0][t];
    const DynScale ds = dyn_scale(sample_max(rw_absmax<1, 2>(acc)));
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[0][t] *= ds.s;
    const Epi<2> ea = epi(R.ba, R.ga, R.bea, R.tb + tb_off, R.isa);
    conv(wb1, [&] { stage_weights<2 * GB::FRAGS5>(R.wb_bf, wb0, wave, lane); });
    gn(std::true_type{}, ea, ds.inv, R.act_a);
    TR(trb + 3);
    const Epi<2> eb = epi(R.bb, R.gb, R.beb, nullptr, R.isb);
    conv(wb0, [&] {                                          // the tail's two parity passes
      stage_weights<TF>(a.wt_bf0, wb1, wave, lane);
      stage_weights<TF>(a.wt_bf1, wb1 + TF * 1024, wave, lane);
    });
    gn(std::false_type{}, eb, one, 1.f);
    TR(trb + 4);
  }
  // ---- tail: Upsample1d as two parity passes -> the final block's input (L = 64), the half's rows of the 64-row slab
  const char* const vaF = slab + g * GF::G + (n + 32 * hf) * 16;
  char* const vsF = slab + (n >> 2) * GF::G + (n & 3) * 4;
  f32x4 y[2][2];
  float inv_f;
  {
    const DynScale ds = dyn_scale(sample_max(rw_absmax<1, 2>(acc)));
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[0][t] *= ds.s;
    const u32x4* const t0 = reinterpret_cast<const u32x4*>(wb1) + lane;
    const u32x4* const t1 = reinterpret_cast<const u32x4*>(wb1 + TF * 1024) + lane;
    const u32x4* wt0[2] = {t0, t0 + (TF / 2) * 64};
    const u32x4* wt1[2] = {t1, t1 + (TF / 2) * 64};
    const float bt[2] = {a.bt[c0], a.bt[c0 + 1]};
    const float is0[2] = {a.ist0[c0] * ds.inv, a.ist0[c0 + 1] * ds.inv}, is1[2] = {a.ist1[c0] * ds.inv, a.ist1[c0 + 1] * ds.inv};
    rw_store2<GB, 1>(vsB, acc);
    staged_weights_landed();
    __syncthreads();
    stage_weights<2 * GF::FRAGS5>(f.w5, wb0, wave, lane);    // the final block's k5 conv
    f32x4 e[1][2], o[1][2];
    rd_ring_load<GB, 2, 2>(ring2, wt0);
    rd_taps<GB, 2, 1, 2, true, false, 1, 2>(e, res, vaB, wt0, wt0, ring2);
    rd_ring_load<GB, 2, 2>(ring2, wt1);
    rd_taps<GB, 2, 2, 2, true, false, 1, 2>(o, res, vaB, wt1, wt1, ring2);
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e[0][t][r] = fmaf(e[0][t][r], is0[t], bt[t]);
        o[0][t][r] = fmaf(o[0][t][r], is1[t], bt[t]);
        m = fmaxf(m, fmaxf(fabsf(e[0][t][r]), fabsf(o[0][t][r])));
      }
    m = row_max16(m);
    m = max_xor16(m);
    m = max_xor32(m);
    const DynScale df = dyn_scale(sample_max(m));            // (its barrier: the tail's reads are done, 64-row geometry)
    inv_f = df.inv;
    if (hf == 0 && lane < 32) *reinterpret_cast<uint4*>(slab + (lane >> 4) * GF::PS + ((lane >> 2) & 3) * GF::G + ((lane & 3) < 2 ? (lane & 3) : 64 + (lane & 3)) * 16) = make_uint4(0u, 0u, 0u, 0u);
    // positions 2 m + parity, m = 16 hf + 4 g + r: rows 2 + 32 hf + 8 g + 2 r + parity
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const F16Pair pe = f16_split2(e[0][0][r] * df.s, e[0][1][r] * df.s), po = f16_split2(o[0][0][r] * df.s, o[0][1][r] * df.s);
      char* d = vsF + (2 + 32 * hf + 8 * g + 2 * r) * 16;
      *reinterpret_cast<unsigned*>(d) = pe.hi;
      *reinterpret_cast<unsigned*>(d + GF::PS) = pe.lo;
      *reinterpret_cast<unsigned*>(d + 16) = po.hi;
      *reinterpret_cast<unsigned*>(d + 16 + GF::PS) = po.lo;
    }
  }
  TR(trb + 5);
  // ---- final block: Conv1dBlock(32 -> 32, k5) + GroupNorm + Mish, then the 1x1 conv 32 -> 4, on the half's two M tiles
  {
    const u32x4* wf[2] = {reinterpret_cast<const u32x4*>(wb0) + lane, reinterpret_cast<const u32x4*>(wb0) + GF::FRAGS5 * 64 + lane};
    const u32x4* w1[1] = {reinterpret_cast<const u32x4*>(f.w1_bf) + lane};
    u32x4 ring1[1][1][2];
    rd_ring_load<GF, 1, 1>(ring1, w1);
    const Epi<2> ef = epi_load<2>(f.bias, f.gamma, f.beta, nullptr, f.isc, c0);
    const float b1 = f.w1_bias[n & 3], s1 = f.is1[n & 3];
    staged_weights_landed();
    __syncthreads();                                         // the final block's input and weights are complete
    rd_ring_load<GF, 2, 2>(ring2, wf);
    rd_taps<GF, 2, 0, 5, true, false, 2, 2>(y, y, vaF, wf, wf, ring2);
    rw_gn_mish_half<2, 2, 2, 256, true>(y, ef.b, ef.g, ef.be, ef.is, inv_f, act_scale(f.act), [](int, int, int) { return 0.f; }, xch, hf, lane);
    // (the exchange's barrier: the partner is past its taps, the slab may be overwritten; the 1x1 conv reads only the centre
    // tap = the wave's own rows)
    rw_store2<GF, 2>(vsF + (2 + 4 * g + 32 * hf) * 16, y);
    wave_lds_fence();
    f32x4 out[2][1];
    rd_taps<GF, 1, 2, 1, true, false, 2, 1>(out, out, vaF, w1, w1, ring1);
    if (fs.enabled) {
      // eps[64][4] -> the sample's slab as float4 rows (both waves their halves), then the unguided ddpm_sample_fn step on the
      // trajectory by the sample's first wave, lane = support point (chain_body_u1w)
      float* const et = reinterpret_cast<float*>(slab);
      __syncthreads();                                       // (both waves' 1x1 reads of the slab are done)
      if (n < 4) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) et[(32 * hf + 16 * mt + 4 * g + r) * 4 + n] = fmaf(out[mt][0][r], s1, b1);
      }
      __syncthreads();
      if (hf == 0 && sp < NV && n0 + sp < a.n) {
        const float4 e = *reinterpret_cast<const float4*>(et + lane * 4);
        const int traj = fs.traj0 + n0 + sp, robot = traj / fs.spr;
        const size_t idx = (size_t)traj * H + lane;
        float4 v = ddpm_posterior_mean(fs.x[idx], e, fs.a_t, fs.b_t, fs.c1, fs.c2);
        if (fs.do_noise)
          v = add_step_noise(v, fs.noise ? fs.noise[idx] : traj_normal4(fs.seed, fs.robot_seeds, fs.draw, fs.traj_base, idx, robot, fs.spr), fs.sigma,
                             fs.noise_std_extra);
        float4 hv;
        if (hard_row(fs.hard_rows, fs.n_hard, fs.hard, robot, lane, hv)) v = hv;
        fs.x[idx] = v;
        if (fs.chain) fs.chain[idx] = v;
      }
    } else if (n < 4 && sp < NV && n0 + sp < a.n) {
      float* dst = f.out + ((size_t)(n0 + sp) * 64 + 32 * hf + 4 * g) * 4 + n;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(16 * mt + r) * 4] = fmaf(out[mt][0][r], s1, b1);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// The whole TemporalUnet forward for 4 samples in ONE workgroup / ONE launch: the five level chains and the final conv
// hand their activations to each other through LDS (tail tile -> next stage's x slab), the two skip connections wait in
// registers (32 VGPRs each) for the up path.  HBM traffic per trajectory and forward: 1 KiB in, 1 KiB out.
// ----------------------------------------------------------------------------------------------------------------
struct UnetArgs {
  ChainArgs c[5];
  FinalArgs fin;
  int n;
  FusedStep fs;           // enabled: the unguided DDPM step on the launch's trajectories follows in the same kernel
};

static_assert(CH_D0::SPB == 4 && CH_D1::SPB == 4 && CH_D2::SPB == 4 && CH_U0::SPB == 4 && CH_U1::SPB == 4,
              "every stage must own the same 4 samples");

// NS = trajectories per workgroup.  4: the form everything above is written for.  2 (launched for small batches, which leave
// most CUs without a workgroup otherwise: twice the workgroups): only samples 0, 1 exist -- the L = 16 stages (downs.2 + mid,
// ups.0: 3/4 of the matrix work, waves = channel slices x ALL samples) run over two samples, i.e. half the MFMAs, A reads,
// epilogue and parking per wave for the same weight stream, and downs.1's waves (n-tile pair x sample PAIR) take one sample each,
// while the stages whose waves ARE samples (downs.0, ups.1 + final block) keep their form with samples 2, 3 fed zeros and
// never stored.  Per-sample arithmetic is the same
// instruction sequence either way: the results are bitwise equal.
// tb_off: added to every RTB's time-bias pointer (floats) -- 0 in unet_kernel, whose host side bakes the step's row of the time
// table into the pointers; t * tb_total in the persistent kernel, whose pointers are those of row 0
// One trajectory per workgroup (unet_kernel<1>, launches of <= ns1_max = 256 trajectories -- one workgroup per CU at most; ONE planner
// call has 64): the L = 16 stages
// (downs.2 + mid, ups.0: their waves are channel slices x all samples) run ONE M tile per conv -- the conv's time there is the weight
// stream plus what the samples' MFMAs, A-fragment reads and epilogues add to it: 3.15 -> 2.7 us per 128 -> 128 conv at <= 64 workgroups
// (tools/ubench/pair_split.hip, arms basePF / base1PF) -- while the stages whose waves are sample halves or n-tile pairs x samples
// (downs.0, downs.1, ups.1 + final block) keep the two-trajectory form with sample 1 fed zeros and never stored.  A sample's
// arithmetic is the same instruction sequence: bitwise the results of unet_kernel<2> / <4>.
__device__ __forceinline__ void unet_forward_body1(const UnetArgs& a, const FusedStep& fs, int tb_off, float* lds, int n0, int lane, int wave) {
  f32x4 skip1[2][2], skip2[1][2], mid_out[1][2], xe[2][1], xo[2][1];
#pragma unroll
  for (int r = 0; r < 4; ++r) xe[1][0][r] = xo[1][0][r] = 0.f;
  chain_body_d0s<CH_D0, 1>(a.c[0], lds, n0, lane, wave, 0, tb_off);
  chain_body_d1d<CH_D1, CH_D2, 2>(a.c[1], lds, lane, wave, skip1, 40, tb_off);
  chain_body_d2d<CH_D2, 1>(a.c[2], lds, lane, wave, mid_out, skip2, 80, tb_off);
  TR(130);
  {
    using G128 = RdGeo<128>;
    constexpr int S_OFF = (CH_D2::SPB * CH_D2::XSS * 4 + 255) / 256 * 256;
    char* const slab128 = reinterpret_cast<char*>(lds) + S_OFF;
    char* const vs = slab128 + wave * G128::G + ((lane & 15) >> 2) * G128::BX + (2 + 4 * (lane >> 4)) * 16 + (lane & 3) * 4;
    chain_body_u0d<CH_U0, 1>(a.c[3], lds, lane, wave, mid_out, skip2, [&](const f32x4 (&t)[1][2]) { rd_store2<G128>(vs, t); }, slab128,
                             reinterpret_cast<f32x4(&)[1][1]>(xe), reinterpret_cast<f32x4(&)[1][1]>(xo), 136, tb_off);
  }
  TR(131);
  chain_body_u1s<CH_U1, 1>(a.c[4], a.fin, fs, lds, n0, lane, wave, xe, xo, skip1, 146, tb_off);
  TR(133);
}

template <int NS>
__device__ __forceinline__ void unet_forward_body(const UnetArgs& a, const FusedStep& fs, int tb_off, float* lds, int n0, int lane, int wave) {
  f32x4 skip1[NS][2], skip2[NS][2];
  // ---- downs.0 @ L=64 -> [4][32][32]: wave = sample, direct f16x2 convs on the wave's own slab (chain_body_d0w)
  if constexpr (NS == 2) chain_body_d0s<CH_D0>(a.c[0], lds, n0, lane, wave, 0, tb_off);
  else chain_body_d0w<CH_D0, NS>(a.c[0], lds, n0, lane, wave, 0, tb_off);
  // ---- downs.1 @ L=32 -> [4][16][64], skip1: direct f16x2 convs, wave = (n-tile pair, sample pair) (chain_body_d1d)
  chain_body_d1d<CH_D1, CH_D2, NS>(a.c[1], lds, lane, wave, skip1, 40, tb_off);
  // ---- downs.2 + mid blocks @ L=16 -> [NS][16][128], skip2: direct f16x2 convs (chain_body_d2d; lane = channels 32 wave + 2
  //      (lane & 15) + h, positions 4 (lane >> 4) + r of all NS samples)
  f32x4 mid_out[NS][2];
  chain_body_d2d<CH_D2, NS>(a.c[2], lds, lane, wave, mid_out, skip2, 80, tb_off);
  TR(130);
  // ---- ups.0 @ L=16: cat(x, skip2) -> [NS][32][64] (chain_body_u0d; the chunks are stored from downs.2's tiles); its output
  //      stays in registers (even / odd positions of channel 16 wave + (lane & 15))
  f32x4 xe[NS][1], xo[NS][1];
  {
    using G128 = RdGeo<128>;
    constexpr int S_OFF = (CH_D2::SPB * CH_D2::XSS * 4 + 255) / 256 * 256;
    char* const slab128 = reinterpret_cast<char*>(lds) + S_OFF;
    char* const vs = slab128 + wave * G128::G + ((lane & 15) >> 2) * G128::BX + (2 + 4 * (lane >> 4)) * 16 + (lane & 3) * 4;
    chain_body_u0d<CH_U0, NS>(a.c[3], lds, lane, wave, mid_out, skip2, [&](const f32x4 (&t)[NS][2]) { rd_store2<G128>(vs, t); },
                              slab128, xe, xo, 136, tb_off);
  }
  TR(131);
  // ---- ups.1 @ L=32: cat(x, skip1) -> [4][64][32], final_conv: Conv1dBlock(32->32) -> 1x1 conv (32->4) -> eps[n,64,4]:
  //      wave = sample (chain_body_u1w)
  if constexpr (NS == 2) chain_body_u1s<CH_U1>(a.c[4], a.fin, fs, lds, n0, lane, wave, xe, xo, skip1, 146, tb_off);
  else chain_body_u1w<CH_U1, NS>(a.c[4], a.fin, fs, lds, n0, lane, wave, xe, xo, skip1, 146, tb_off);
  TR(133);
}

template <int NS>
__device__ __forceinline__ void unet_forward_any(const UnetArgs& a, const FusedStep& fs, int tb_off, float* lds, int n0, int lane, int wave) {
  if constexpr (NS == 1) unet_forward_body1(a, fs, tb_off, lds, n0, lane, wave);
  else unet_forward_body<NS>(a, fs, tb_off, lds, n0, lane, wave);
}

template <int NS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void unet_kernel(UnetArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[UNET_LDS_FLOATS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unet_forward_any<NS>(a, a.fs, 0, lds, blockIdx.x * NS, lane, wave);
}

// A RUN of consecutive unguided DDPM steps in ONE launch (mmd_p_sample_loop: the steps before guidance starts, or every step of a
// prior-only call): a workgroup iterates the steps of its own NS trajectories -- forward, fused ddpm_sample_fn step (the wave that
// holds a trajectory's eps writes x in place), the next forward reads what the same workgroup wrote.  No launch boundary between
// the steps: no dispatch gap, and the workgroups of a CU never wait for the slowest workgroup of the chip.  sc[s]: the step's
// schedule coefficients and its row of the time table (a.c[*].*.tb point at row 0); chain / injected noise advance by one
// batch per step.  Same arithmetic as the launch-per-step form: bitwise-equal results.
// The argument block comes through a pointer into the constant address space, re-derived from an opaque integer every step: as
// by-value kernel arguments inside a loop the ~300 pointers were hoisted out of it, i.e. kept -- spilled -- across the whole forward
// (1100 VGPR spills); loaded where they are used (s_load from a uniform address) the loop body compiles like unet_kernel's.
static_assert(sizeof(UnetArgs) <= PERSIST_TABLE_BYTES - PERSIST_ARGS_OFF, "the argument block fits its workspace region");
template <int NS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void unet_persist_kernel(const UnetArgs* ap, const FusedStep* steps,
                                                                                                     int n_steps, int tb_total) {
  __shared__ __attribute__((aligned(16))) float lds[UNET_LDS_FLOATS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  typedef const __attribute__((address_space(4))) UnetArgs* ConstArgs;
  typedef const __attribute__((address_space(4))) FusedStep* ConstStep;
  for (int s = 0; s < n_steps; ++s) {
    unsigned long long pa = reinterpret_cast<unsigned long long>(ap), ps = reinterpret_cast<unsigned long long>(steps + s);
    asm volatile("" : "+s"(pa), "+s"(ps));
    const UnetArgs& a = *(const UnetArgs*)(ConstArgs)pa;
    const FusedStep& fs = *(const FusedStep*)(ConstStep)ps;     // (read where the fused step uses it: the tail of the forward)
    // (an opaque copy of the lane index per step: lane-derived slab offsets are loop invariant, and hoisted out of the loop they
    // would stay live -- spilled -- across the whole forward)
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    unet_forward_body<NS>(a, fs, fs.t_row * tb_total, lds, blockIdx.x * NS, lane_s, wave);
    // every wave is done with the LDS of this step, and the trajectories the workgroup wrote are visible to all of its waves
    // (unet_kernel<2>: a sample's second wave reads what its first wave stored)
    __threadfence_block();
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------------------------------
// time embedding table: TimeEncoder (layers.py:232-258) + every block's cond_mlp (layers.py:337-341) for all integer t
// ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mish_exact(float y) {
  float sp = y > 20.f ? y : log1pf(expf(y));
  return y * tanhf(sp);
}

__global__ void time_table_kernel(TimeArgs a) {
  __shared__ float emb[32], h1[128], m[32];
  const int t = blockIdx.x, tid = threadIdx.x;   // 128 threads
  if (tid < 16) {
    const float f = expf((float)tid * -(logf(10000.f) / 15.f));
    const float v = (float)t * f;
    emb[tid] = sinf(v);
    emb[tid + 16] = cosf(v);
  }
  __syncthreads();
  {
    float s = a.b1[tid];
    for (int k = 0; k < 32; ++k) s += a.w1[tid * 32 + k] * emb[k];
    h1[tid] = mish_exact(s);
  }
  __syncthreads();
  if (tid < 32) {
    float s = a.b3[tid];
    for (int k = 0; k < 128; ++k) s += a.w3[tid * 128 + k] * h1[k];
    m[tid] = mish_exact(s);          // cond_mlp starts with Mish
  }
  __syncthreads();
  for (int r = 0; r < a.n_rtb; ++r)
    for (int c = tid; c < a.cout[r]; c += 128) {
      float s = a.cb[r][c];
      for (int k = 0; k < 32; ++k) s += a.cw[r][c * 32 + k] * m[k];
      a.table[(size_t)t * a.total + a.off[r] + c] = s;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// host side: parameter spec, weight packing, forward orchestration
// ----------------------------------------------------------------------------------------------------------------

// the fused kernel's configuration: unet_input_dim 32, dim_mults (1, 2, 4) (every released checkpoint); anything else build_spec
// accepts runs layer by layer (unet_layers.hip)
static inline bool fused_config(int uid, int n_levels) { return uid == 32 && n_levels == 3; }

void launch_time_table(const TimeArgs& a, int T, hipStream_t st) { hipLaunchKernelGGL(time_table_kernel, dim3(T), dim3(128), 0, st, a); }

// downs.0's first conv (4 -> 32, k5) as ONE K = 32 chunk per n-tile: slot jj of lane (column n, lane group j) = tap 2 j + (jj >> 2),
// channel jj & 3 (taps >= 5: zero); is_res: the 1x1 residual conv [cout][4] on the centre tap's slots.  Interleaved column pairs.
static size_t pack_im2col4(std::vector<float>& blob, const float* w, int cout, bool is_res, const std::vector<float>& sc) {
  while (blob.size() % 4) blob.push_back(0.f);
  const size_t base = blob.size();
  const int tiles = cout / 16;
  blob.resize(base + ((size_t)tiles * 2 + 8) * 64 * 4, 0.f);
  uint16_t* out = reinterpret_cast<uint16_t*>(blob.data() + base);
  for (int t = 0; t < tiles; ++t)
    for (int lane = 0; lane < 64; ++lane)
      for (int jj = 0; jj < 8; ++jj) {
        const int n = (t / 2) * 32 + 2 * (lane & 15) + (t & 1), tap = 2 * (lane >> 4) + (jj >> 2), c = jj & 3;
        float v = 0.f;
        if (is_res) v = tap == 2 ? w[(size_t)n * 4 + c] : 0.f;
        else if (tap < 5) v = w[((size_t)n * 4 + c) * 5 + tap];
        uint16_t piece[2];
        f16_split_host(v, sc[n], piece);
        for (int q = 0; q < 2; ++q) out[(((size_t)t * 2 + q) * 64 + lane) * 8 + jj] = piece[q];
      }
  return base;
}
// ... of a 1x1 residual conv [cout][cin_full] chunk for the centre tap's extra streams: per n-tile [chunk kc][piece][lane]
static size_t pack_rd_res(std::vector<float>& blob, const float* wres, int cout, int cin_full, int c_lo, int cin_chunk,
                          bool pair_cols, const std::vector<float>& sc) {
  return pack_rd(blob, wres, cout, cin_full, c_lo, cin_chunk, 1, std::vector<int>{0}, false, pair_cols, sc);
}

// max over t of |time bias| per (RTB, channel): a host replica (double) of time_table_kernel, used only for the bound
// behind the static activation scales below (layers.py:232-258, 337-341)
static std::vector<std::vector<float>> time_bias_absmax(const float* const* tensors, const Spec& s, int T) {
  auto mish = [](double y) { return y * std::tanh(y > 20.0 ? y : std::log1p(std::exp(y))); };
  const float *w1 = tensors[s.t_time[0]], *b1 = tensors[s.t_time[1]], *w3 = tensors[s.t_time[2]], *b3 = tensors[s.t_time[3]];
  std::vector<std::vector<float>> mxv(s.rtb.size());
  for (size_t r = 0; r < s.rtb.size(); ++r) mxv[r].assign(s.rtb[r].cout, 0.f);
  for (int t = 0; t < T; ++t) {
    double emb[32], h1[128], m[32];
    for (int i = 0; i < 16; ++i) {
      const double f = std::exp((double)i * -(std::log(10000.0) / 15.0)), v = (double)t * f;
      emb[i] = std::sin(v); emb[i + 16] = std::cos(v);
    }
    for (int j = 0; j < 128; ++j) { double a = b1[j]; for (int k = 0; k < 32; ++k) a += (double)w1[j * 32 + k] * emb[k]; h1[j] = mish(a); }
    for (int j = 0; j < 32; ++j) { double a = b3[j]; for (int k = 0; k < 128; ++k) a += (double)w3[j * 128 + k] * h1[k]; m[j] = mish(a); }
    for (size_t r = 0; r < s.rtb.size(); ++r) {
      const float *cw = tensors[s.rtb[r].t_cw], *cb = tensors[s.rtb[r].t_cb];
      for (int c = 0; c < s.rtb[r].cout; ++c) {
        double a = cb[c];
        for (int k = 0; k < 32; ++k) a += (double)cw[c * 32 + k] * m[k];
        mxv[r][c] = fmaxf(mxv[r][c], (float)std::fabs(a));
      }
    }
  }
  return mxv;
}
// Static f16x2 input scale of a conv B: its input Mish(GroupNorm(.)) + time bias is bounded whatever the data --
// |x_hat| <= sqrt(N - 1) < 16 for a group of N = 256 values, Mish(y) in [-0.31, max(y, 0)] -- by B = max_c (max(0.31, 16
// |gamma_c| + |beta_c|) + max_t |tb_c(t)|); the power of two s = 2^(10 - floor(log2 B)) keeps every value below 2048 inside fp16
// and the typical ones (|x_hat| ~ 1) far above its denormals.
static float static_act_scale(const float* gamma, const float* beta, const std::vector<float>& tbmax, int c) {
  float B = 0.f;
  for (int i = 0; i < c; ++i) B = fmaxf(B, fmaxf(0.31f, 16.f * fabsf(gamma[i]) + fabsf(beta[i])) + tbmax[i] * 1.001f);
  if (!std::isfinite(B)) return 1.f;
  int ex;
  (void)frexpf(B, &ex);                      // B = f * 2^ex, f in [0.5, 1): floor(log2 B) = ex - 1
  return ldexpf(1.f, 10 - (ex - 1));
}

struct ConvW { size_t bias, gamma, beta, wbf, isc; };   // wbf / isc: f16x2 pack and its inverse channel scales
struct RtbW { ConvW a, b; size_t res_bias, res_isc, res_bf, res_c1_bf; int tb_off; size_t a_c1_bf; float act_a; };

}  // namespace mmd

using namespace mmd;

struct mmd_unet_s {
  mmd::LayeredUnet* layered = nullptr;   // set: a configuration other than the fused kernel's; everything below is unused
  int T = 0;
  int ns2_max = 512;         // unet_kernel<2> (two trajectories per workgroup) up to this batch size (mmd_unet_options.two_per_workgroup_max)
#ifndef MMD_NS1_MAX
#define MMD_NS1_MAX 256
#endif
  int ns1_max = MMD_NS1_MAX; // unet_kernel<1> (one trajectory per workgroup) up to this batch size (-DMMD_NS1_MAX=0: the A/B side build)
  size_t blob_bytes = 0;     // bytes of `blob` (mmd_unet_weight_bytes)
  float* blob = nullptr;     // packed weights / biases / affine params
  float* ttable = nullptr;   // [T][tb_total]
  int tb_total = 0;
  RtbW rtb[12];              // state_dict order: d00 d01 d10 d11 d20 d21 u00 u01 u10 u11 mid1 mid2
  ConvW down[2], up[2], fin;
  size_t up_bf[2][2] = {}, up_is[2][2] = {};     // the up stages' tails: f16x2 parity packs and their inverse scales
  size_t down_bf[2] = {0, 0}, down_is[2] = {0, 0};   // the down stages' tails: f16x2 pack and its inverse scales
  size_t fin_w1 = 0, fin_b1 = 0, fin_is1 = 0;    // final 1x1 conv: f16x2 pack (N padded to 16), bias, inverse scales / fin_act
  float fin_act = 1.f;                           // static scale of the final block's activations
};

// caller-owned event-pair pool (include/mmd_amd_debug.h); the unet handle itself is immutable after creation
struct mmd_profiler_s {
  int stride = 1, window = 1, period = 0;   // launch i of a period is bracketed iff (i / window) % stride == 0
  std::vector<hipEvent_t> ev;               // event pairs, in bracketing order
  std::vector<int> kind;                    // per pair: MMD_PROF_UNET / _UNET_FUSED / _STEP_GUIDED / _STEP_PLAIN
  hipEvent_t base = nullptr;                // recorded with the first bracket: origin of the interval clock
  size_t used = 0;                          // events handed out
  size_t seen[2] = {0, 0};                  // launches counted: UNet, step kernel (the same steps of both are bracketed)
};

namespace mmd {
// counter 0: UNet launches, 1: step-kernel launches.  Both are issued once per (step, stream chunk) in the same order, so
// the same steps of both are bracketed.
bool prof_begin(mmd_profiler_t prof, int counter, int kind, hipStream_t st) {
  if (!prof) return false;
  const size_t i = prof->period > 0 ? prof->seen[counter] % (size_t)prof->period : prof->seen[counter];
  ++prof->seen[counter];
  if (((i / (size_t)prof->window) % (size_t)prof->stride) != 0 || prof->used + 2 > prof->ev.size()) return false;
  if (prof->used == 0) (void)hipEventRecord(prof->base, st);
  (void)hipEventRecord(prof->ev[prof->used], st);
  prof->kind[prof->used / 2] = kind;
  return true;
}
void prof_end(mmd_profiler_t prof, hipStream_t st) {
  (void)hipEventRecord(prof->ev[prof->used + 1], st);
  prof->used += 2;
}
void prof_skip(mmd_profiler_t prof, int counter) {
  if (prof) ++prof->seen[counter];
}
}  // namespace mmd

namespace mmd {

static size_t push(std::vector<float>& blob, const float* p, int64_t n) {
  while (blob.size() % 4) blob.push_back(0.f);
  size_t off = blob.size();
  blob.insert(blob.end(), p, p + n);
  while (blob.size() % 4) blob.push_back(0.f);
  return off;
}

static RtbPtrs rtb_ptrs(const mmd_unet_s* u, const RtbW& w, int t) {
  RtbPtrs p{};
  p.ba = u->blob + w.a.bias; p.ga = u->blob + w.a.gamma; p.bea = u->blob + w.a.beta;
  p.tb = u->ttable + (size_t)t * u->tb_total + w.tb_off;
  p.bb = u->blob + w.b.bias; p.gb = u->blob + w.b.gamma; p.beb = u->blob + w.b.beta;
  p.wa_bf = w.a.wbf ? reinterpret_cast<const uint4*>(u->blob + w.a.wbf) : nullptr;
  p.wb_bf = w.b.wbf ? reinterpret_cast<const uint4*>(u->blob + w.b.wbf) : nullptr;
  p.isa = w.a.wbf ? u->blob + w.a.isc : nullptr;
  p.isb = w.b.wbf ? u->blob + w.b.isc : nullptr;
  p.act_a = w.act_a;
  return p;
}

// chain over RTBs rtb[0] (first) and rtb[1..n_ident]; tail = down/up conv weights or null
static ChainArgs args_chain(const mmd_unet_s* u, const RtbW* set, const int* rtb, int n_ident, const ConvW* tail,
                            const float* in0, int t, int n) {
  ChainArgs a{};
  a.in0 = in0; a.n = n;
  const RtbW& w0 = set[rtb[0]];
  a.r0 = rtb_ptrs(u, w0, t);
  a.wa0_c1_bf = w0.a_c1_bf ? reinterpret_cast<const uint4*>(u->blob + w0.a_c1_bf) : nullptr;
  a.br = u->blob + w0.res_bias;
  a.isr = w0.res_isc ? u->blob + w0.res_isc : nullptr;
  a.wres_bf = w0.res_bf ? reinterpret_cast<const uint4*>(u->blob + w0.res_bf) : nullptr;
  a.wres_c1_bf = w0.res_c1_bf ? reinterpret_cast<const uint4*>(u->blob + w0.res_c1_bf) : nullptr;
  for (int k = 0; k < n_ident; ++k) a.ri[k] = rtb_ptrs(u, set[rtb[1 + k]], t);
  if (tail) a.bt = u->blob + tail->bias;
  if (tail == &u->down[0] || tail == &u->down[1]) {
    const int i = tail == &u->down[1];
    a.wt_bf0 = reinterpret_cast<const uint4*>(u->blob + u->down_bf[i]);
    a.ist0 = u->blob + u->down_is[i];
  }
  if (tail == &u->up[0] || tail == &u->up[1]) {
    const int i = tail == &u->up[1];
    a.wt_bf0 = reinterpret_cast<const uint4*>(u->blob + u->up_bf[i][0]);
    a.wt_bf1 = reinterpret_cast<const uint4*>(u->blob + u->up_bf[i][1]);
    a.ist0 = u->blob + u->up_is[i][0];
    a.ist1 = u->blob + u->up_is[i][1];
  }
  return a;
}

}  // namespace mmd

extern "C" {

int mmd_unet_num_tensors(int unet_input_dim, int n_levels) {
  Spec s;
  if (!build_spec(unet_input_dim, n_levels, s)) {
    set_error("unsupported TemporalUnet configuration (unet_input_dim=%d, n_levels=%d)", unet_input_dim, n_levels);
    return -1;
  }
  return (int)s.numel.size();
}

int64_t mmd_unet_tensor_numel(int unet_input_dim, int n_levels, int index) {
  Spec s;
  if (!build_spec(unet_input_dim, n_levels, s) || index < 0 || index >= (int)s.numel.size()) return -1;
  return s.numel[index];
}

int mmd_unet_create(mmd_unet_t* out, int unet_input_dim, int n_levels, int n_diffusion_steps,
                    const float* const* tensors, const int64_t* numels, int n_tensors, const mmd_unet_options* options,
                    void* stream) {
  Spec s;
  MMD_REQUIRE(out != nullptr, "mmd_unet_create: out is NULL");
  MMD_REQUIRE(build_spec(unet_input_dim, n_levels, s),
              "unsupported TemporalUnet configuration (unet_input_dim=%d, n_levels=%d)", unet_input_dim, n_levels);
  MMD_REQUIRE(n_tensors == (int)s.numel.size(), "expected %d parameter tensors, got %d", (int)s.numel.size(), n_tensors);
  for (int i = 0; i < n_tensors; ++i)
    MMD_REQUIRE(numels[i] == s.numel[i] && tensors[i] != nullptr, "parameter tensor %d has %lld elements, expected %lld",
                i, (long long)numels[i], (long long)s.numel[i]);
  MMD_REQUIRE(n_diffusion_steps >= 1, "n_diffusion_steps must be >= 1");
  hipStream_t st = (hipStream_t)stream;

  auto* u = new mmd_unet_s();
  u->T = n_diffusion_steps;
  if (options && options->two_per_workgroup_max != 0) u->ns2_max = options->two_per_workgroup_max < 0 ? 0 : options->two_per_workgroup_max;
  // MMD_UNET_LAYERED: the layer-by-layer path for the fused kernel's own configuration too -- the two implementations share no
  // device code, tests/test_gpu_dim_mults.py holds one against the other
  if (!fused_config(unet_input_dim, n_levels) || (options && (options->flags & MMD_UNET_LAYERED))) {
    // e.g. UNET_DIM_MULTS[1] = (1, 2, 4, 8): layer by layer (unet_layers.hip)
    if (int rc = layered_create(&u->layered, s, n_diffusion_steps, tensors, options, st)) {
      delete u;
      return rc;
    }
    *out = u;
    return 0;
  }
  std::vector<float> blob;
  const std::vector<int> taps5 = {0, 1, 2, 3, 4}, taps3 = {0, 1, 2}, taps1 = {0};
  size_t raw_time[4];
  for (int i = 0; i < 4; ++i) raw_time[i] = push(blob, tensors[s.t_time[i]], s.numel[s.t_time[i]]);
  size_t raw_cw[12], raw_cb[12];
  int tb_off = 0;
  const std::vector<std::vector<float>> tbmax = time_bias_absmax(tensors, s, n_diffusion_steps);
  for (int r = 0; r < 12; ++r) {
    // state_dict order: d00 d01 d10 d11 d20 d21 u00 u01 u10 u11 mid1 mid2.  Every conv gets the one f16x2 pack its stage
    // body reads: direct packs (pack_rd; interleaved column pairs where a wave owns two n-tiles), with the 1x1 residual conv of
    // a stage's first RTB as a one-tap pack of its own.
    const Rtb& R = s.rtb[r];
    RtbW& W = u->rtb[r];
    W = RtbW{};
    W.act_a = 1.f;
    while (blob.size() % 4) blob.push_back(0.f);
    const bool d1s = r == 2 || r == 3, d2 = r == 4 || r == 5 || r == 10 || r == 11;   // d1s: downs.1 (chain_body_d1d)
    const float* wres = R.res ? tensors[R.t_rw] : nullptr;   // 1x1 residual conv [cout][cin]: fused into conv A's pack
    const std::vector<int> k5 = {0, 1, 2, 3, 4};
    const bool d0 = r == 0 || r == 1;                  // downs.0 (chain_body_d0w): interleaved column pairs like downs.2
    const bool u1 = r == 8 || r == 9;                  // ups.1 (chain_body_u1w): wave-private, interleaved column pairs
    const bool pairs = d2 || d0 || u1 || d1s;
    if (r == 0) {             // downs.0's first conv (4 -> 32): one im2col chunk + the 1x1 residual conv's chunk
      const std::vector<float> sc = rd_col_scales(tensors[R.t_w0], R.cout, R.cin, 5, k5, false);
      const std::vector<float> scr = rd_col_scales(wres, R.cout, R.cin, 1, std::vector<int>{0}, false);
      W.a.isc = push_inverse(blob, sc);
      W.res_isc = push_inverse(blob, scr);
      W.a.wbf = pack_im2col4(blob, tensors[R.t_w0], R.cout, false, sc);
      W.res_bf = pack_im2col4(blob, wres, R.cout, true, scr);
    } else if (r == 6 || r == 4 || r == 8 || r == 2) {   // conv A of a stage's first RTB: direct f16x2 (rd_taps), the 1x1 residual
                              // conv on the centre tap; the up stages' input is the two chunks of cat(x, skip)
      const std::vector<float> sc = rd_col_scales(tensors[R.t_w0], R.cout, R.cin, 5, k5, false);
      const std::vector<float> scr = rd_col_scales(wres, R.cout, R.cin, 1, std::vector<int>{0}, false);
      const bool whole = r == 4 || r == 2;
      const int chunk = whole ? R.cin : R.cin / 2;
      W.a.isc = push_inverse(blob, sc);
      W.res_isc = push_inverse(blob, scr);
      W.a.wbf = pack_rd(blob, tensors[R.t_w0], R.cout, R.cin, 0, chunk, 5, k5, false, pairs, sc);
      W.res_bf = pack_rd_res(blob, wres, R.cout, R.cin, 0, chunk, pairs, scr);
      if (!whole) {
        W.a_c1_bf = pack_rd(blob, tensors[R.t_w0], R.cout, R.cin, chunk, chunk, 5, k5, false, pairs, sc);
        W.res_c1_bf = pack_rd_res(blob, wres, R.cout, R.cin, chunk, chunk, pairs, scr);
      }
    } else if (R.cin == R.cout) {   // conv A of an identity RTB of the direct stages: dynamic input scale
      const std::vector<float> sc = rd_col_scales(tensors[R.t_w0], R.cout, R.cin, 5, k5, false);
      W.a.isc = push_inverse(blob, sc);
      W.a.wbf = pack_rd(blob, tensors[R.t_w0], R.cout, R.cin, 0, R.cin, 5, k5, false, pairs, sc);
    } else {
      set_error("mmd_unet_create: no pack for RTB %d", r);
      delete u;
      return 1;
    }
    W.a.bias = push(blob, tensors[R.t_b0], R.cout);
    W.a.gamma = push(blob, tensors[R.t_g0], R.cout);
    W.a.beta = push(blob, tensors[R.t_be0], R.cout);
    {                         // conv B: direct f16x2, its input scaled by the static act_a
      W.act_a = static_act_scale(tensors[R.t_g0], tensors[R.t_be0], tbmax[r], R.cout);
      const std::vector<float> sc = rd_col_scales(tensors[R.t_w1], R.cout, R.cout, 5, k5, false);
      W.b.isc = push_inverse(blob, sc, W.act_a);
      W.b.wbf = pack_rd(blob, tensors[R.t_w1], R.cout, R.cout, 0, R.cout, 5, k5, false, pairs, sc);
    }
    W.b.bias = push(blob, tensors[R.t_b1], R.cout);
    W.b.gamma = push(blob, tensors[R.t_g1], R.cout);
    W.b.beta = push(blob, tensors[R.t_be1], R.cout);
    raw_cw[r] = push(blob, tensors[R.t_cw], (int64_t)R.cout * 32);
    raw_cb[r] = push(blob, tensors[R.t_cb], R.cout);
    W.res_bias = R.res ? push(blob, tensors[R.t_rb], R.cout) : 0;
    W.tb_off = tb_off;
    tb_off += R.cout;
  }
  u->tb_total = tb_off;
  const int dims[4] = {4, unet_input_dim, unet_input_dim * 2, unet_input_dim * 4};
  for (int i = 0; i < 2; ++i) {
    const int c = dims[i + 1];
    u->down[i].bias = push(blob, tensors[s.t_down[i][1]], c);
    {                         // Downsample1d as a direct f16x2 conv (taps 0..2, interleaved column pairs) read at stride 2
      const std::vector<int> k3 = {0, 1, 2};
      const std::vector<float> sct = rd_col_scales(tensors[s.t_down[i][0]], c, c, 3, k3, false);
      u->down_is[i] = push_inverse(blob, sct);
      u->down_bf[i] = pack_rd(blob, tensors[s.t_down[i][0]], c, c, 0, c, 3, k3, false, true, sct);
    }
    const int cu = dims[2 - i];
    // ConvTranspose1d(k=4, s=2, p=1): out[2m] = in[m-1] W3 + in[m] W1 ; out[2m+1] = in[m] W2 + in[m+1] W0, as two direct f16x2
    // parity passes (chain_body_u0d; chain_body_u1w: interleaved column pairs)
    u->up[i].bias = push(blob, tensors[s.t_up[i][1]], cu);
    {
      const std::vector<int> ke = {3, 1}, ko = {2, 0};
      const std::vector<float> sce = rd_col_scales(tensors[s.t_up[i][0]], cu, cu, 4, ke, true);
      const std::vector<float> sco = rd_col_scales(tensors[s.t_up[i][0]], cu, cu, 4, ko, true);
      u->up_is[i][0] = push_inverse(blob, sce);
      u->up_is[i][1] = push_inverse(blob, sco);
      u->up_bf[i][0] = pack_rd(blob, tensors[s.t_up[i][0]], cu, cu, 0, cu, 4, ke, true, i == 1, sce);
      u->up_bf[i][1] = pack_rd(blob, tensors[s.t_up[i][0]], cu, cu, 0, cu, 4, ko, true, i == 1, sco);
    }
  }
  {                           // final block (chain_body_u1w): k5 conv with a dynamic input scale, 1x1 conv behind a static one
    const std::vector<float> sc = rd_col_scales(tensors[s.t_final[0]], 32, 32, 5, taps5, false);
    u->fin.isc = push_inverse(blob, sc);
    u->fin.wbf = pack_rd(blob, tensors[s.t_final[0]], 32, 32, 0, 32, 5, taps5, false, true, sc);
    u->fin.bias = push(blob, tensors[s.t_final[1]], 32);
    u->fin.gamma = push(blob, tensors[s.t_final[2]], 32);
    u->fin.beta = push(blob, tensors[s.t_final[3]], 32);
    u->fin_act = static_act_scale(tensors[s.t_final[2]], tensors[s.t_final[3]], std::vector<float>(32, 0.f), 32);
    std::vector<float> sc1 = rd_col_scales(tensors[s.t_final[4]], 4, 32, 1, taps1, false);
    u->fin_is1 = push_inverse(blob, sc1, u->fin_act);
    std::vector<float> w1(16 * 32, 0.f);      // N padded to one n-tile
    memcpy(w1.data(), tensors[s.t_final[4]], sizeof(float) * 4 * 32);
    sc1.resize(16, 1.f);
    u->fin_w1 = pack_rd(blob, w1.data(), 16, 32, 0, 32, 1, taps1, false, false, sc1);
    u->fin_b1 = push(blob, tensors[s.t_final[5]], 4);
  }

  if (hipMalloc(&u->blob, blob.size() * sizeof(float)) != hipSuccess ||
      hipMalloc(&u->ttable, (size_t)u->T * u->tb_total * sizeof(float)) != hipSuccess) {
    set_error("mmd_unet_create: hipMalloc failed");
    mmd_unet_destroy(u);
    return 1;
  }
  MMD_HIP_CHECK(hipMemcpyAsync(u->blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice, st));
  MMD_HIP_CHECK(hipStreamSynchronize(st));   // blob is a local vector
  u->blob_bytes = blob.size() * sizeof(float);

  TimeArgs ta{};
  ta.w1 = u->blob + raw_time[0]; ta.b1 = u->blob + raw_time[1];
  ta.w3 = u->blob + raw_time[2]; ta.b3 = u->blob + raw_time[3];
  for (int r = 0; r < 12; ++r) {
    ta.cw[r] = u->blob + raw_cw[r]; ta.cb[r] = u->blob + raw_cb[r];
    ta.cout[r] = s.rtb[r].cout; ta.off[r] = u->rtb[r].tb_off;
  }
  ta.n_rtb = 12; ta.total = u->tb_total; ta.table = u->ttable;
  hipLaunchKernelGGL(time_table_kernel, dim3(u->T), dim3(128), 0, st, ta);
  MMD_HIP_CHECK(hipGetLastError());
  MMD_HIP_CHECK(hipStreamSynchronize(st));
  *out = u;
  return 0;
}

int mmd_unet_destroy(mmd_unet_t u) {
  if (!u) return 0;
  layered_destroy(u->layered);
  if (u->blob) (void)hipFree(u->blob);
  if (u->ttable) (void)hipFree(u->ttable);
  delete u;
  return 0;
}

// The forward keeps every intermediate in LDS / registers; the workspace argument is kept in the ABI (callers pass the
// buffer they sized with this function) but only a token size is asked for.
size_t mmd_unet_workspace_bytes(mmd_unet_t u, int n_traj) {
  if (u && u->layered) return layered_workspace_bytes(u->layered, n_traj);   // (that path keeps its activations in the workspace)
  return n_traj > 0 ? PERSIST_TABLE_BYTES : 0;                                // (the step table of a persistent run of unguided steps)
}

// algorithmic FLOPs per trajectory of one forward: sum over its convs of 2 * C_out * taps * C_in * L_out
static constexpr double rtb_flops(double cin, double cout, double L) {
  return 2.0 * cout * 5 * cin * L + 2.0 * cout * 5 * cout * L + (cin != cout ? 2.0 * cout * cin * L : 0.0);
}
static const double kUnetFlops =
    rtb_flops(4, 32, 64) + rtb_flops(32, 32, 64) + 2.0 * 32 * 3 * 32 * 32 +
    rtb_flops(32, 64, 32) + rtb_flops(64, 64, 32) + 2.0 * 64 * 3 * 64 * 16 +
    rtb_flops(64, 128, 16) + 3 * rtb_flops(128, 128, 16) +
    rtb_flops(256, 64, 16) + rtb_flops(64, 64, 16) + 2.0 * 64 * 4 * 64 * 16 +
    rtb_flops(128, 32, 32) + rtb_flops(32, 32, 32) + 2.0 * 32 * 4 * 32 * 32 +
    2.0 * 32 * 5 * 32 * 64 + 2.0 * 4 * 32 * 64;

static constexpr double d5(double cin, double cout, double L) { return 2.0 * cout * 5 * cin * L; }
static const double kF16Flops =
    2 * (2.0 * 32 * 32 * 64) + 3 * d5(32, 32, 64) + 2.0 * 32 * 3 * 32 * 32 +                          // downs.0
    d5(32, 64, 32) + 2.0 * 64 * 32 * 32 + 3 * d5(64, 64, 32) + 2.0 * 64 * 3 * 64 * 16 +               // downs.1
    d5(64, 128, 16) + 2.0 * 128 * 64 * 16 + 7 * d5(128, 128, 16) +                                    // downs.2 + mid
    d5(256, 64, 16) + 2.0 * 64 * 256 * 16 + 3 * d5(64, 64, 16) + 2.0 * 64 * 4 * 64 * 16 +             // ups.0
    d5(128, 32, 32) + 2.0 * 32 * 128 * 32 + 3 * d5(32, 32, 32) + 2.0 * 32 * 4 * 32 * 32 +             // ups.1
    d5(32, 32, 64) + 2.0 * 16 * 32 * 64;                                                              // final block
static const double kFp32Flops = 0.0;                                                                 // (no fp32 MFMA left)
static const double kUnetMfmaFlops = kF16Flops + kFp32Flops;


static int unet_forward_impl(mmd_unet_t u, const float* x, int t, float* eps, int n, void* ws, size_t ws_bytes,
                             hipStream_t st, mmd_profiler_t prof, const FusedStep* fs = nullptr) {
  MMD_REQUIRE(u && x && eps && ws, "mmd_unet_forward: NULL argument");
  MMD_REQUIRE(n >= 1, "mmd_unet_forward: n_traj must be >= 1");
  MMD_REQUIRE(t >= 0 && t < u->T, "mmd_unet_forward: t=%d outside [0,%d)", t, u->T);
  MMD_REQUIRE(ws_bytes >= mmd_unet_workspace_bytes(u, n), "mmd_unet_forward: workspace too small");
  if (u->layered) {
    MMD_REQUIRE(!(fs && fs->enabled), "mmd_unet_forward: the fused unguided step exists in the fused kernel only");
    const bool bracket = prof_begin(prof, 0, MMD_PROF_UNET, st);
    const int rc = layered_forward(u->layered, x, t, eps, n, ws, ws_bytes, st);
    if (bracket) prof_end(prof, st);
    return rc;
  }
  // state_dict RTB indices: d00 d01 d10 d11 d20 d21 u00 u01 u10 u11 mid1 mid2 = 0..11
  static const int kD0[] = {0, 1}, kD1[] = {2, 3}, kD2[] = {4, 5, 10, 11}, kU0[] = {6, 7}, kU1[] = {8, 9};
  const RtbW* set = u->rtb;
  UnetArgs a{};
  a.n = n;
  if (fs) a.fs = *fs;
  a.c[0] = args_chain(u, set, kD0, 1, &u->down[0], x, t, n);
  a.c[1] = args_chain(u, set, kD1, 1, &u->down[1], nullptr, t, n);
  a.c[2] = args_chain(u, set, kD2, 3, nullptr, nullptr, t, n);
  a.c[3] = args_chain(u, set, kU0, 1, &u->up[0], nullptr, t, n);
  a.c[4] = args_chain(u, set, kU1, 1, &u->up[1], nullptr, t, n);
  a.fin.out = eps;
  a.fin.w5 = reinterpret_cast<const uint4*>(u->blob + u->fin.wbf);
  a.fin.isc = u->blob + u->fin.isc;
  a.fin.bias = u->blob + u->fin.bias; a.fin.gamma = u->blob + u->fin.gamma; a.fin.beta = u->blob + u->fin.beta;
  a.fin.act = u->fin_act;
  a.fin.w1_bf = reinterpret_cast<const uint4*>(u->blob + u->fin_w1);
  a.fin.is1 = u->blob + u->fin_is1;
  a.fin.w1_bias = u->blob + u->fin_b1;
  const bool bracket = prof_begin(prof, 0, fs && fs->enabled ? MMD_PROF_UNET_FUSED : MMD_PROF_UNET, st);
  // two trajectories per workgroup while that still leaves at most one workgroup per CU (256 CUs): see unet_kernel
  if (n <= u->ns1_max && n <= u->ns2_max) hipLaunchKernelGGL(unet_kernel<1>, dim3(n), dim3(256), 0, st, a);
  else if (n <= u->ns2_max) hipLaunchKernelGGL(unet_kernel<2>, dim3((n + 1) / 2), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(unet_kernel<4>, dim3((n + 3) / 4), dim3(256), 0, st, a);
  if (bracket) prof_end(prof, st);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
namespace mmd {
// A run of n_steps <= PERSIST_MAX_STEPS unguided steps of ALL n trajectories in one launch (unet_persist_kernel).  steps: host array of
// the steps' complete fused-step descriptors (x, hard, seed, traj0 = 0, the step's own coefficients / draw / noise and chain rows /
// t_row), copied into the workspace's table region in stream order (pageable source: staged by the runtime before the call returns).
int unet_persist_steps(mmd_unet_t u, int n, void* ws, size_t ws_bytes, hipStream_t st, const FusedStep* steps, int n_steps) {
  MMD_REQUIRE(u && !u->layered && ws && n >= 1, "unet_persist_steps: bad arguments");
  MMD_REQUIRE(n_steps >= 1 && n_steps <= PERSIST_MAX_STEPS && ws_bytes >= (size_t)PERSIST_TABLE_BYTES, "unet_persist_steps: step table");
  for (int s = 0; s < n_steps; ++s) MMD_REQUIRE(steps[s].t_row >= 0 && steps[s].t_row < u->T, "unet_persist_steps: t outside the time table");
  // (`steps` and the argument block below are PAGEABLE host memory: hipMemcpyAsync stages such a source into the runtime's own pinned
  // buffer before it returns -- the documented behaviour for pageable memory -- so the caller's stack array may be reused for the next
  // run; it also makes the copy host-synchronous, one reason this opt-in mode does not pay on the headline, and it is not legal inside a
  // stream capture, which mmd_p_sample_loop checks before it takes this path.  The first 12 KiB of the workspace are clobbered.)
  MMD_HIP_CHECK(hipMemcpyAsync(ws, steps, sizeof(FusedStep) * n_steps, hipMemcpyHostToDevice, st));
  char* const args_dev = reinterpret_cast<char*>(ws) + PERSIST_ARGS_OFF;
  static const int kD0[] = {0, 1}, kD1[] = {2, 3}, kD2[] = {4, 5, 10, 11}, kU0[] = {6, 7}, kU1[] = {8, 9};
  const RtbW* set = u->rtb;
  float* x = reinterpret_cast<float*>(steps[0].x);
  UnetArgs a{};
  a.n = n;
  a.c[0] = args_chain(u, set, kD0, 1, &u->down[0], x, 0, n);
  a.c[1] = args_chain(u, set, kD1, 1, &u->down[1], nullptr, 0, n);
  a.c[2] = args_chain(u, set, kD2, 3, nullptr, nullptr, 0, n);
  a.c[3] = args_chain(u, set, kU0, 1, &u->up[0], nullptr, 0, n);
  a.c[4] = args_chain(u, set, kU1, 1, &u->up[1], nullptr, 0, n);
  a.fin.out = nullptr;                                     // (every step is fused: eps never leaves the workgroup)
  a.fin.w5 = reinterpret_cast<const uint4*>(u->blob + u->fin.wbf);
  a.fin.isc = u->blob + u->fin.isc;
  a.fin.bias = u->blob + u->fin.bias; a.fin.gamma = u->blob + u->fin.gamma; a.fin.beta = u->blob + u->fin.beta;
  a.fin.act = u->fin_act;
  a.fin.w1_bf = reinterpret_cast<const uint4*>(u->blob + u->fin_w1);
  a.fin.is1 = u->blob + u->fin_is1;
  a.fin.w1_bias = u->blob + u->fin_b1;
  MMD_HIP_CHECK(hipMemcpyAsync(args_dev, &a, sizeof(a), hipMemcpyHostToDevice, st));   // (pageable source: staged before the call returns)
  const FusedStep* sd = reinterpret_cast<const FusedStep*>(ws);
  const UnetArgs* ap = reinterpret_cast<const UnetArgs*>(args_dev);
  if (n <= u->ns2_max) hipLaunchKernelGGL(unet_persist_kernel<2>, dim3((n + 1) / 2), dim3(256), 0, st, ap, sd, n_steps, u->tb_total);
  else hipLaunchKernelGGL(unet_persist_kernel<4>, dim3((n + 3) / 4), dim3(256), 0, st, ap, sd, n_steps, u->tb_total);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}
bool unet_fused_step_supported(mmd_unet_t u) { return u && !u->layered; }
int unet_forward_fused(mmd_unet_t u, const float* x, int t, float* eps, int n, void* ws, size_t ws_bytes, ::mmd_profiler_s* prof,
                       hipStream_t st, const FusedStep& fs) {
  return unet_forward_impl(u, x, t, eps, n, ws, ws_bytes, st, prof, &fs);
}
}  // namespace mmd
extern "C" {

int mmd_unet_forward(mmd_unet_t u, const float* x, int t, float* eps, int n, void* ws, size_t ws_bytes, void* stream) {
  return unet_forward_impl(u, x, t, eps, n, ws, ws_bytes, (hipStream_t)stream, nullptr);
}

int mmd_unet_forward_profiled(mmd_unet_t u, const float* x, int t, float* eps, int n, void* ws, size_t ws_bytes,
                              mmd_profiler_t prof, void* stream) {
  return unet_forward_impl(u, x, t, eps, n, ws, ws_bytes, (hipStream_t)stream, prof);
}

#ifdef MMD_TRACE
int mmd_debug_set_trace(void* dev_ptr) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dev_ptr, sizeof(dev_ptr)) == hipSuccess ? 0 : 1;
}
#endif

size_t mmd_unet_weight_bytes(mmd_unet_t u) {
  if (!u) return 0;
  return u->layered ? layered_weight_bytes(u->layered) : u->blob_bytes;
}

double mmd_unet_flops_per_trajectory(void) { return kUnetFlops; }
double mmd_unet_mfma_flops_per_trajectory(void) { return kUnetMfmaFlops; }
double mmd_unet_f16x2_flops_per_trajectory(void) { return kF16Flops; }

int mmd_profiler_create(mmd_profiler_t* out, int max_launches, int stride) {
  return mmd_profiler_create_windowed(out, max_launches, stride, 1, 0);
}

int mmd_profiler_create_windowed(mmd_profiler_t* out, int max_launches, int stride, int window, int period) {
  MMD_REQUIRE(out && max_launches > 0, "mmd_profiler_create: bad arguments");
  auto* p = new mmd_profiler_s();
  p->stride = stride > 0 ? stride : 1;
  p->window = window > 0 ? window : 1;
  p->period = period > 0 ? period : 0;
  if (hipEventCreate(&p->base) != hipSuccess) {
    set_error("mmd_profiler_create: hipEventCreate failed");
    delete p;
    return 1;
  }
  p->ev.resize((size_t)2 * max_launches);
  p->kind.assign((size_t)max_launches, 0);
  for (auto& e : p->ev)
    if (hipEventCreate(&e) != hipSuccess) {
      set_error("mmd_profiler_create: hipEventCreate failed");
      e = nullptr;
      mmd_profiler_destroy(p);
      return 1;
    }
  *out = p;
  return 0;
}

int mmd_profiler_destroy(mmd_profiler_t p) {
  if (!p) return 0;
  for (auto& e : p->ev)
    if (e) (void)hipEventDestroy(e);
  if (p->base) (void)hipEventDestroy(p->base);
  delete p;
  return 0;
}

int mmd_profiler_intervals(mmd_profiler_t p, int kind, double* start_ms, double* end_ms, int cap, int* n_out) {
  MMD_REQUIRE(p && start_ms && end_ms && n_out, "mmd_profiler_intervals: NULL argument");
  int cnt = 0;
  for (size_t i = 0; i + 1 < p->used && cnt < cap; i += 2) {
    if (p->kind[i / 2] != kind) continue;
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, p->base, p->ev[i]) == hipSuccess && hipEventElapsedTime(&b, p->base, p->ev[i + 1]) == hipSuccess) {
      start_ms[cnt] = a;
      end_ms[cnt] = b;
      ++cnt;
    }
  }
  *n_out = cnt;
  return 0;
}

int mmd_profiler_read(mmd_profiler_t p, double* mean_ms, int* n_launches) {
  MMD_REQUIRE(p && mean_ms && n_launches, "mmd_profiler_read: NULL argument");
  double tot = 0.0;
  int cnt = 0;
  for (size_t i = 0; i + 1 < p->used; i += 2) {
    float ms = 0.f;
    if ((p->kind[i / 2] == MMD_PROF_UNET || p->kind[i / 2] == MMD_PROF_UNET_FUSED) && hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]) == hipSuccess) { tot += ms; ++cnt; }
  }
  *mean_ms = cnt ? tot / cnt : 0.0;
  *n_launches = cnt;
  p->used = 0;
  p->seen[0] = p->seen[1] = 0;
  return 0;
}

}  // extern "C"
