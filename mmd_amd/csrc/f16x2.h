// fp32 GEMMs on the fp16 matrix pipe ("f16x2") -- the building blocks shared by the fused TemporalUnet kernel (unet.hip) and the
// layer-by-layer path (unet_layers.hip): the two-piece fp16 split of an fp32 operand, the MFMA triple of one K = 32 chunk, the
// dynamic per-sample input scale, and the host-side packing of conv weights into MFMA B-fragment order.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace mmd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Dynamic f16x2 input scale of a conv whose input is NOT bounded by a GroupNorm (the input of a ResidualTemporalBlock: the
// residual stream, which follows the magnitude of the network input): per sample, from the exact maximum M of the conv's
// input tile, s = 2^(10 - floor(log2 M)), so that every value |x| s < 2048 fits fp16 whatever
// the input's magnitude, and the values that matter (within 2^-13 of the maximum) keep both pieces normal.  inv = 1 / s.
struct DynScale { float s, inv; };
__device__ __forceinline__ DynScale dyn_scale(float M) {
  unsigned eb = (__float_as_uint(M) >> 23) & 0xffu;        // M >= 0: biased exponent (inf / NaN: 255 -> NaN out, as in fp32)
  eb = eb < 64u ? 64u : eb;                                  // M < 2^-63 (all zero): s = 2^73
  return DynScale{__uint_as_float((264u - eb) << 23), __uint_as_float((eb - 10u) << 23)};
}
// ----------------------------------------------------------------------------------------------------------------
// fp32 GEMM on the fp16 matrix pipe ("f16x2"): every conv of the network.  Every fp32 operand is split into TWO fp16 pieces
// by rounding to nearest, x0 = RN16(x), x1 = RN16(x - x0) (x - x0 is exact in fp32; each rounding is good to 2^-11 of what it rounds: |x - x0 - x1| <= 2^-22 |x|, measured worst case
// 2^-23, rms 4.2e-8 -- fp32 rounding itself: 3.4e-8 -- as long as x1 stays above fp16's denormal step 2^-24), and a product a * w is accumulated as a1 w0 + a0 w1 + a0 w0 (low order first) on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation; the dropped a1 w1 is <= 2^-22 |a w|.  Measured against fp64 the result is
// more accurate than the fp32 MFMA chain it replaces and than the three-piece bf16 split of round 2
// (tools/ubench/f16x2_emul.hip: rms error 1.5e-7 / 2.9e-7 / 4.0e-7 of rms(D) at K = 128 / 640 / 1280 against 2.1e-7 / 4.4e-7 /
// 6.4e-7 for the fp32 chain and 1.7e-7 / 3.9e-7 / 5.5e-7 for bf16x3) at HALF the matrix-pipe time of bf16x3 (3 instead of 6
// MFMAs per K = 32 chunk, 1/5 of the fp32 MFMA's) and 4 instead of 6 bytes per weight.  fp16 has five exponent bits:
// gfx950's MFMA honours fp16 denormal inputs (probed in the same ubench), so a low piece below 2^-14 keeps an ABSOLUTE
// precision of 2^-25; the weights of every output channel are scaled on the host by a power of two that puts the channel's
// largest |w| into [2^14, 2^15) (exact; undone for free inside the GroupNorm epilogue, `isc`), and conv inputs carry a static
// (conv B) or dynamic per-sample (conv A) power-of-two scale that keeps them below 2048 and their low pieces normal or within
// 2^-25 of it (ActScale / dyn_scale below).
// Three MFMAs on ONE accumulator issue back to back without a bubble, while an MFMA that depends on the one two
// before it waits (ubench: 8 accumulators x 3 in a row 17.2 cycles per MFMA, two alternating accumulators 30): a step
// runs its accumulator streams one after the other (vb_three keeps a triple together).
// ----------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the two pieces of a pair of values (v0: the even channel) as dwords {v1 piece, v0 piece}: hi = RN16(v), lo = RN16(v - hi)
struct F16Pair { unsigned hi, lo; };
__device__ __forceinline__ F16Pair f16_split2(float v0, float v1) {
  const f16x2 h = __builtin_convertvector(f32x2{v0, v1}, f16x2);                  // v_cvt_pk_f16_f32 (round to nearest even)
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  // v - h (exact in fp32) as ONE v_fma_mix_f32 each -- h * (-1) + v with the fp16 half read in place -- instead of a
  // conversion and a subtraction
  float d0, d1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hb), "v"(v0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hb), "v"(v1));
  const f16x2 l = __builtin_convertvector(f32x2{d0, d1}, f16x2);
  return F16Pair{hb, __builtin_bit_cast(unsigned, l)};
}
__device__ __forceinline__ f32x4 mfma_h(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// one K = 32 chunk of one accumulator stream: a1 b0 + a0 b1 + a0 b0, back to back on the same accumulator
template <bool ZERO>
__device__ __forceinline__ void vb_three(f32x4& x, const u32x4 (&a)[2], const u32x4 (&b)[2]) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 c = ZERO ? z : x;
  c = mfma_h(a[1], b[0], c);
  c = mfma_h(a[0], b[1], c);
  c = mfma_h(a[0], b[0], c);
  x = c;
#ifndef MMD_VB3_LOOSE
  // keep the triple together: left alone, the scheduler interleaves the streams of a half step round robin, and with only two
  // streams (ups.0: one n-tile per wave) every MFMA then waits for the one two before it (30 instead of 17 cycles each)
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// ----------------------------------------------------------------------------------------------------------------
// host side: per-channel weight scales and B-fragment packs
// ----------------------------------------------------------------------------------------------------------------
// power of two that puts m into [2^14, 2^15) (fp16's largest binade but one); 1 for m = 0 or non-finite
static inline float f16_scale_for(float m) {
  if (!(m > 0.f) || !std::isfinite(m)) return 1.f;
  int ex;
  (void)frexpf(m, &ex);                    // m = f * 2^ex, f in [0.5, 1)
  return ldexpf(1.f, 15 - ex);
}
// u * scale -> hi = RN16, lo = RN16(u * scale - hi) (the device-side split of f16_split2)
static inline void f16_split_host(float u, float scale, uint16_t (&piece)[2]) {
  const float us = u * scale;              // exact (power of two), |us| < 2^15
  const _Float16 h = (_Float16)us;
  const _Float16 l = (_Float16)(us - (float)h);
  memcpy(&piece[0], &h, 2);
  memcpy(&piece[1], &l, 2);
}
static size_t push_inverse(std::vector<float>& blob, const std::vector<float>& sc, float in_scale = 1.f) {
  while (blob.size() % 4) blob.push_back(0.f);
  const size_t off = blob.size();
  for (float v : sc) blob.push_back(1.f / (v * in_scale));   // powers of two: exact
  while (blob.size() % 4) blob.push_back(0.f);
  return off;
}

// ---- direct f16x2 packs (rd_taps): per n-tile [tap][chunk kc][piece q][lane] x 16 B; lane = (column n, the 8 channels of
// block kc + KC (lane >> 4) of the chunk [c_lo, c_lo + cin_chunk)); kidx[tap] = the kernel index of slab-row tap `tap`; conv
// weight layout [cout][cin_full][ks], or ConvTranspose1d [cin_full][cout][ks] (transposed); pair_cols: the interleaved
// n-tile pairs of chain_body_d2d.  sc = the per-output-channel scales (rd_col_scales over ALL chunks and taps).
static inline float rd_w(const float* w, int cout, int cin_full, int ks, bool transposed, int n, int ci, int k) {
  return transposed ? w[((size_t)ci * cout + n) * ks + k] : w[((size_t)n * cin_full + ci) * ks + k];
}
static std::vector<float> rd_col_scales(const float* w, int cout, int cin_full, int ks, const std::vector<int>& kidx, bool transposed) {
  std::vector<float> sc(cout);
  for (int n = 0; n < cout; ++n) {
    float m = 0.f;
    for (int ci = 0; ci < cin_full; ++ci)
      for (int k : kidx) m = fmaxf(m, fabsf(rd_w(w, cout, cin_full, ks, transposed, n, ci, k)));
    sc[n] = f16_scale_for(m);
  }
  return sc;
}
static size_t pack_rd(std::vector<float>& blob, const float* w, int cout, int cin_full, int c_lo, int cin_chunk, int ks,
                      const std::vector<int>& kidx, bool transposed, bool pair_cols, const std::vector<float>& sc) {
  while (blob.size() % 4) blob.push_back(0.f);
  const size_t base = blob.size();
  const int tiles = cout / 16, KC = cin_chunk / 32, T = (int)kidx.size();
  const size_t frags = (size_t)tiles * T * KC * 2;
  blob.resize(base + (frags + 8) * 64 * 4, 0.f);             // + slack for the ring's over-read past the last tile
  uint16_t* out = reinterpret_cast<uint16_t*>(blob.data() + base);
  for (int t = 0; t < tiles; ++t)
    for (int tap = 0; tap < T; ++tap)
      for (int kc = 0; kc < KC; ++kc)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int n = pair_cols ? (t / 2) * 32 + 2 * (lane & 15) + (t & 1) : 16 * t + (lane & 15);
            const int ci = c_lo + 8 * (KC * (lane >> 4) + kc) + j;
            uint16_t piece[2];
            f16_split_host(rd_w(w, cout, cin_full, ks, transposed, n, ci, kidx[tap]), sc[n], piece);
            for (int q = 0; q < 2; ++q) out[(((((size_t)t * T + tap) * KC + kc) * 2 + q) * 64 + lane) * 8 + j] = piece[q];
          }
  return base;
}

}  // namespace mmd
