// GroupNorm affine + Mish as the fused TemporalUnet kernel (unet.hip) and the layer-by-layer kernels (unet_layers.hip) evaluate it.
#pragma once
#include <hip/hip_runtime.h>

namespace mmd {

// Mish(y) = y * tanh(softplus(y)) = y * n / (n + 2), n = e^y (e^y + 2)   (torch.nn.Mish; softplus threshold 20).
// GroupNorm affine + Mish + the add that follows it (time bias after conv A, residual after conv B) in 9 VALU ops per
// activation (every VALU op costs the SIMD ~5 cycles of MFMA issue).  Everything is carried in units of log2(e):
// yl = y * log2 e = x * sa + sb with sa = rstd * gamma * log2 e, sb = (beta - mean * rstd * gamma) * log2 e folded per
// (sample, channel); e^y = exp2(yl); the 1 / log2 e is folded into the denominator: q = n / ((n + 2) log2 e), so
// y * n / (n + 2) = yl * q, and the trailing add rides in the last fma.  Clamping the exponent at 20 makes n / (n + 2)
// round to 1 for y > 20 (n ~ 2.4e17), i.e. mish(y) = y up to an ulp or two: the softplus threshold branch of
// torch.nn.Mish without a select.
struct GnCoef { float sa, sb; };
__device__ __forceinline__ GnCoef gn_coef(float mean, float rstd, float gamma, float beta) {
  constexpr float LOG2E = 1.44269504088896341f;
  GnCoef c;
  const float s = rstd * gamma;
  c.sa = s * LOG2E;
  c.sb = fmaf(-mean, s, beta) * LOG2E;
  return c;
}
// ActScale: the activations this epilogue produces are carried times a power of two `s` (the static f16x2 input scale of the
// conv that consumes them, chosen on the host from the GroupNorm affine / time-bias bounds): mish(y) * s = yl * (n / ((n + 2)
// log2 e / s)) -- the scale rides in the denominator's two constants, which become registers (l2e = log2 e / s, l2e2 = 2 l2e);
// the addend arrives already scaled.  ACT = false: the literal constants (s = 1).
struct ActScale { float l2e, l2e2; };
__device__ __forceinline__ ActScale act_scale(float s) {
  constexpr float LOG2E = 1.44269504088896341f;
  const float inv = __builtin_amdgcn_rcpf(s);            // exact: s is a power of two
  return ActScale{LOG2E * inv, 2.f * LOG2E * inv};
}

// The same for two values with packed fp32 arithmetic (v_pk_fma / v_pk_mul / v_pk_add: 2 values per ~1.25 issue slots): six
// packed + two v_min + four transcendental ops per pair instead of 2 x 9.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <bool ACT = false>
__device__ __forceinline__ f32x2_t gn_mish2(f32x2_t x, const GnCoef& c, f32x2_t addend, const ActScale& as = ActScale{}) {
  constexpr float LOG2E = 1.44269504088896341f;
#ifdef MMD_GN_SCALAR   // (A/B build: the same arithmetic as plain fp32 VALU ops -- next to another wave's MFMA stream a v_pk_* issues
                       // once per MFMA, a plain VALU op every ~10 cycles: tools/ubench/mfma_valu_overlap.hip)
  f32x2_t out;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float yl = fmaf(x[i], c.sa, c.sb);
    const float e = __builtin_amdgcn_exp2f(fminf(yl, 20.f * LOG2E));
    const float n = e * (e + 2.f);
    const float den = fmaf(n, ACT ? as.l2e : LOG2E, ACT ? as.l2e2 : 2.f * LOG2E);
    out[i] = fmaf(yl, n * __builtin_amdgcn_rcpf(den), addend[i]);
  }
  return out;
#endif
  const f32x2_t sa = {c.sa, c.sa}, sb = {c.sb, c.sb}, two = {2.f, 2.f};
  const f32x2_t yl = __builtin_elementwise_fma(x, sa, sb);
  const f32x2_t e = {__builtin_amdgcn_exp2f(fminf(yl.x, 20.f * LOG2E)), __builtin_amdgcn_exp2f(fminf(yl.y, 20.f * LOG2E))};
  const f32x2_t n = e * (e + two);
  const f32x2_t k1 = ACT ? f32x2_t{as.l2e, as.l2e} : f32x2_t{LOG2E, LOG2E}, k2 = ACT ? f32x2_t{as.l2e2, as.l2e2} : f32x2_t{2.f * LOG2E, 2.f * LOG2E};
  const f32x2_t den = __builtin_elementwise_fma(n, k1, k2);
  const f32x2_t q = n * f32x2_t{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  return __builtin_elementwise_fma(yl, q, addend);
}

}  // namespace mmd
