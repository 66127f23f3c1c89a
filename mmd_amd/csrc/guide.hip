// Cost-guidance gradient + DDPM step algebra for gfx950.
//
// One wavefront per trajectory, lane = support point t (H = 64 = wave width): a trajectory's 1 KiB is one fully
// coalesced 16-byte-per-lane load, the GP-prior stencil talks to lanes t-1 / t+1 through wave shuffles, and all
// n_guide_steps inner iterations of guide_gradient_steps (sample_functions.py:89-107) run in registers.  The SDF
// grid is an L2-resident float4 texture (sdf, dsdf/dx, dsdf/dy, 0) so a point costs one 16-byte gather.  A
// workgroup is 4 waves = 4 trajectories of the same robot, so the robot's constraint groups are block-uniform and
// the time-bucketed (ELL) constraint table [slot][t] is read coalesced across lanes.
//
// Closed-form gradients (no autograd), term by term as GuideManagerTrajectoriesWithVelocity.forward does
// (guides.py:180-226): per cost -> clip by norm over the 4 state dims with +1e-6 inside (guides.py:247-253) -> zero
// rows 0 and H-1 -> weight -> sum -> negate.  The gradient is taken w.r.t. the UN-normalised trajectory and added to
// the NORMALISED one (sample_functions.py:104), as the reference does.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/mmd_amd.h"
#include "../../include/mmd_amd_debug.h"
#include "common.h"
#include "guide_dev.h"

namespace mmd {

// clip_grad_by_norm (guides.py:247-253): scale = clip(||g + 1e-6||, 0, max) / ||g + 1e-6||
// n >= 2e-6 always (the +1e-6), so min(n, max) / n = min(1, max / n): one v_rsq_f32 (1 ulp) instead of the IEEE sqrt and
// divide sequences (~25 VALU instructions, five times per guide iteration); the factor is exactly 1 whenever the term is
// not clipped, as in the reference.
__device__ __forceinline__ float clip_scale(float gx, float gy, float gz, float gw, float max_norm) {
  const float ax = gx + 1e-6f, ay = gy + 1e-6f, az = gz + 1e-6f, aw = gw + 1e-6f;
  const float n2 = ax * ax + ay * ay + az * az + aw * aw;
  return fminf(max_norm * __builtin_amdgcn_rsqf(n2), 1.f);
}

// GuideManager.clip_gradient (guides.py:228-259) of one cost's per-point gradient: by norm (the planners' setting), by value
// (torch.clip to +-max_grad_value per component) or not at all (clip_grad = False).  The rule is wave uniform.
// Returns whether the rule was ACTIVE on this point (the norm / a component beyond the limit): used by the trace instantiation
// only, dead code elsewhere.
__device__ __forceinline__ bool clip_grad(const GuideDev& g, float& gx, float& gy, float& gz, float& gw) {
  if (g.clip_rule == 0) {
    const float sc = clip_scale(gx, gy, gz, gw, g.max_norm);
    gx = sc * gx; gy = sc * gy; gz = sc * gz; gw = sc * gw;
    return sc < 1.f;
  } else if (g.clip_rule == 1) {
    const float m = g.max_value;
    const bool on = fabsf(gx) > m || fabsf(gy) > m || fabsf(gz) > m || fabsf(gw) > m;
    gx = fminf(fmaxf(gx, -m), m); gy = fminf(fmaxf(gy, -m), m); gz = fminf(fmaxf(gz, -m), m); gw = fminf(fmaxf(gw, -m), m);
    return on;
  }
  return false;
}
__device__ __forceinline__ bool clip_grad(const GuideDev& g, float& gx, float& gy) {
  float z = 0.f, w = 0.f;
  return clip_grad(g, gx, gy, z, w);
}

// wave shift by one lane through DPP (GFX9 wave_shr:1 / wave_shl:1): lane t reads lane t-1 / t+1, no LDS round trip.
// Lanes shifted in from outside the wave read 0 (bound_ctrl) -- callers mask t == 0 / t == H-1 anyway.
__device__ __forceinline__ float lane_prev(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_next(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// Constraint slots staged per workgroup (1 KiB each).  A workgroup is WPB waves = WPB trajectories of one robot sharing
// the table: 4 waves x 40 slots normally, 8 waves x up to 144 slots (dynamic LDS) for bigger tables when samples_per_robot
// is a multiple of 8 -- so a 128..256-robot instance (weak scaling over 4-8 GPUs) still reads most of its table from LDS.
constexpr int LDS_SLOTS_SMALL = 40;
constexpr int LDS_SLOTS_MAX = 144;

// -sum over a slot range of d/||d|| for points with ||d|| <= R  (CostConstraint, cost_functions.py:297-326), from a
// [slot][t] table `tab` (LDS or global) of (qx, qy, R, R|R|).  A point is active iff R >= 0 and not (dist > R)
// <=> not (dist^2 > R|R|): one transcendental (rsq) per point, no branch (a wave-uniform skip of inactive slot pairs was
// measured slower: most pairs have an active lane somewhere along the horizon).  Two accumulator pairs break the
// dependent add chain.
// 1 / sqrt(d2) for the unit direction d / ||d||.  A point that coincides exactly with a constraint centre has d = 0: the
// reference's torch.norm backward gives a zero gradient there (subgradient 0), so the reciprocal is taken of
// max(d2, tiny) and d * m = 0 instead of 0 * inf = NaN.
__device__ __forceinline__ float rsq_pos(float d2) { return __builtin_amdgcn_rsqf(fmaxf(d2, 1e-30f)); }

// The slot sum of a constraint group is DEFINED as four interleaved partial sums -- accumulator k takes the group's slots k, k + 4,
// k + 8, ... (slot index relative to the group's first slot) in increasing order, each term one explicit fma -- combined as
// (A0 + A1) + (A2 + A3).  The one-wave-per-trajectory kernel evaluates that tree with four accumulators; the cooperative kernel
// (four waves per trajectory, wave k owns accumulator k) produces the very same bits, so a robot's samples do not depend on
// which of the two a launch size selects (the multi-GPU shards stay bitwise equal to the one-GPU run).
typedef float f32x2g __attribute__((ext_vector_type(2)));
struct Acc4 { f32x2g a0, a1, a2, a3; };
__device__ __forceinline__ Acc4 acc4_zero() {
  const f32x2g z = {0.f, 0.f};
  return Acc4{z, z, z, z};
}
__device__ __forceinline__ f32x2g acc4_total(const Acc4& r) {
  return f32x2g{(r.a0.x + r.a1.x) + (r.a2.x + r.a3.x), (r.a0.y + r.a1.y) + (r.a2.y + r.a3.y)};
}
// a <- a - d m,  d = p - q,  m = [||d||^2 <= R|R|] / ||d||   (one slot, one lane).  Plain fp32 VALU ops with explicit fmas (the same
// instruction sequence in both kernels).  The packed form (v_pk_add / v_pk_mul / v_pk_fma on the (x, y) pair: 8 instead of 12
// instructions per slot, -DMMD_GUIDE_PK) was measured and is NOT used: next to another wave's MFMA stream a v_pk_* issues once
// per MFMA where a plain VALU op issues every ~10 cycles (tools/ubench/mfma_valu_overlap.hip), and the guided step of one stream
// chunk runs beside the other chunk's UNet launch -- 108 instead of 83 us per launch in the loop, 86.2 k instead of 92.3 k
// trajectories/s (profiles/r04_guide_packed_ab.txt).
__device__ __forceinline__ f32x2g cons_term(f32x2g a, f32x2g p, f32x2g q, float r2) {
#ifdef MMD_GUIDE_PK
  const f32x2g d = p - q, sq = d * d;
  const float d2 = sq.x + sq.y;
  const float m = (d2 > r2) ? 0.f : rsq_pos(d2);
  return __builtin_elementwise_fma(-d, f32x2g{m, m}, a);
#else
  const float dx = p.x - q.x, dy = p.y - q.y;
  const float d2 = __builtin_fmaf(dx, dx, dy * dy);
  const float m = (d2 > r2) ? 0.f : rsq_pos(d2);
  return f32x2g{__builtin_fmaf(-dx, m, a.x), __builtin_fmaf(-dy, m, a.y)};
#endif
}
// one table entry as (q, R|R|): the general table holds (qx, qy, R, R|R|) (R < 0: empty, R|R| < 0 <= d2), the compact on-chip
// table (qx, qy) with one radius for every active point and "no point" stored as (1e30, 1e30): dist^2 = inf > R^2
template <bool COMPACT> struct ConsTab;
template <> struct ConsTab<false> {
  const float4* tab;
  float r2;
  __device__ __forceinline__ f32x2g add(f32x2g a, f32x2g p, int slot, int t) const {
    const float4 c = tab[slot * H + t];
    return cons_term(a, p, f32x2g{c.x, c.y}, c.w);
  }
};
template <> struct ConsTab<true> {
  const float2* tab;
  float r2;
  __device__ __forceinline__ f32x2g add(f32x2g a, f32x2g p, int slot, int t) const {
    const float2 c = tab[slot * H + t];
    return cons_term(a, p, f32x2g{c.x, c.y}, r2);
  }
};
// accumulator (rel + i) & 3 += slot (first + i) of `tab` for i in [0, n): rel = the first slot's index relative to its group
template <class TAB>
__device__ __forceinline__ void cons_accumulate4(Acc4& acc, const TAB& tab, int first, int n, int rel, int t, f32x2g p) {
  auto one = [&](int k, int slot) {                          // (k is wave-uniform: scalar branches)
    if (k == 0) acc.a0 = tab.add(acc.a0, p, slot, t);
    else if (k == 1) acc.a1 = tab.add(acc.a1, p, slot, t);
    else if (k == 2) acc.a2 = tab.add(acc.a2, p, slot, t);
    else acc.a3 = tab.add(acc.a3, p, slot, t);
  };
  int i = 0;
  for (; i < n && ((rel + i) & 3); ++i) one((rel + i) & 3, first + i);   // up to the next multiple of four
#pragma unroll 2
  for (; i + 3 < n; i += 4) {
    acc.a0 = tab.add(acc.a0, p, first + i, t);
    acc.a1 = tab.add(acc.a1, p, first + i + 1, t);
    acc.a2 = tab.add(acc.a2, p, first + i + 2, t);
    acc.a3 = tab.add(acc.a3, p, first + i + 3, t);
  }
  for (int k = 0; i < n; ++i, ++k) one(k, first + i);         // the last 1 .. 3 slots: accumulators 0, 1, 2
}
// ... the slots of ONE accumulator (wave `k` of the cooperative kernel): first + j for the j in [0, n) with (rel + j) & 3 == k
template <class TAB>
__device__ __forceinline__ f32x2g cons_accumulate1(f32x2g a, const TAB& tab, int first, int n, int rel, int k, int t, f32x2g p) {
#pragma unroll 4
  for (int j = (k - rel) & 3; j < n; j += 4) a = tab.add(a, p, first + j, t);
  return a;
}

// A robot's first two constraint groups' bounds and weights, read ONCE per launch: inside the 20 guide iterations every group evaluation
// otherwise starts with three dependent loads from the (L2-resident) group tables -- the iterations store x / chain rows, so the compiler
// cannot keep them -- in front of its first slot (a robot has one or two groups: soft constraints from the other robots, hard ones from a
// conflict).  Wave-uniform values, selected by wave-uniform branches.
struct GroupMeta {
  int s0[2], s1[2];
  float w[2];
  int grp0, n;
  __device__ __forceinline__ void load(const GuideDev& g, int g0, int g1) {
    grp0 = g0;
    n = min(g1 - g0, 2);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      s0[k] = s1[k] = 0;
      w[k] = 0.f;
      if (k < n) {
        s0[k] = __builtin_amdgcn_readfirstlane(g.grp_slot_off[g0 + k]);
        s1[k] = __builtin_amdgcn_readfirstlane(g.grp_slot_off[g0 + k + 1]);
        w[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, g.grp_weight[g0 + k])));
      }
    }
  }
  __device__ __forceinline__ void bounds(const GuideDev& g, int grp, int& a, int& b) const {
    const int k = grp - grp0;
    if (k == 0 && n > 0) { a = s0[0]; b = s1[0]; }
    else if (k == 1 && n > 1) { a = s0[1]; b = s1[1]; }
    else { a = __builtin_amdgcn_readfirstlane(g.grp_slot_off[grp]); b = __builtin_amdgcn_readfirstlane(g.grp_slot_off[grp + 1]); }
  }
  __device__ __forceinline__ float weight(const GuideDev& g, int grp) const {
    const int k = grp - grp0;
    return (k == 0 && n > 0) ? w[0] : (k == 1 && n > 1) ? w[1] : g.grp_weight[grp];
  }
};

// group_sum(grp, p) = the slot sum of constraint group grp at the lane's position p (the canonical four-accumulator tree above)
// DUMP (mmd_debug_ddpm_step_trace only): the iteration's discrete decisions of this support point -> tr[0 .. MMD_TRACE_WORDS)
// (layout: include/mmd_amd_debug.h).  The constraint masks are re-derived slot by slot from the L2-resident table with
// cons_term's own expression, so they are the decisions the sums above took.
template <bool DUMP = false, class GROUPSUM, class GROUPW>
__device__ __forceinline__ float4 guide_grad(const GuideDev& g, float4 xn, int t, const float4* __restrict__ grid,
                                             int grp0, int grp1, GROUPSUM group_sum, GROUPW group_weight, unsigned int* tr = nullptr) {
  unsigned int flags = 0;
  // LimitsNormalizer.unnormalize (normalization.py:157-168), clip applied unconditionally
  float xu[4];
  const float xv[4] = {xn.x, xn.y, xn.z, xn.w};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    float v = fminf(fmaxf(xv[d], -1.f), 1.f);
    if constexpr (DUMP) flags |= (fabsf(xv[d]) > 1.f ? 1u : 0u) << (16 + d);
    v = (v + 1.f) / 2.f;
    xu[d] = v * g.nscale[d] + g.nmin[d];
  }
  const float px = xu[0], py = xu[1], vx = xu[2], vy = xu[3];
  const bool interior = t > 0 && t < H - 1;   // rows 0 and H-1 are zeroed for every term (guides.py:217-218)

  // SDF gathers are issued first and consumed last: their L2 latency hides behind the GP / constraint arithmetic
  int ix = (int)floorf((px - g.lo[0]) / g.dim[0] * (float)g.nx);
  int iy = (int)floorf((py - g.lo[1]) / g.dim[1] * (float)g.ny);
  ix = min(max(ix, 0), g.nx - 1);
  iy = min(max(iy, 0), g.ny - 1);
  // n_grids == 0: the map has no fixed objects (sdf == 1 everywhere, primitives.py:109-110) -> the term is identically 0
  const float4 cell0 = g.n_grids > 0 ? grid[(size_t)ix * g.ny + iy] : make_float4(1e30f, 0.f, 0.f, 0.f);

  // --- CostCollision over the workspace boundaries (distance_fields.py:354-367)
  float wsx, wsy;
  {
    const float d0 = px - g.ws_min[0], d1 = py - g.ws_min[1], d2 = g.ws_max[0] - px, d3 = g.ws_max[1] - py;
    float best = fmaxf(g.margin - d0, 0.f), gx = -1.f, gy = 0.f;
    float v = fmaxf(g.margin - d1, 0.f);
    if (v > best) { best = v; gx = 0.f; gy = -1.f; }
    v = fmaxf(g.margin - d2, 0.f);
    if (v > best) { best = v; gx = 1.f; gy = 0.f; }
    v = fmaxf(g.margin - d3, 0.f);
    if (v > best) { best = v; gx = 0.f; gy = 1.f; }
    if (!(best > 0.f)) { gx = 0.f; gy = 0.f; }
    if constexpr (DUMP) {
      const unsigned arg = gx < 0.f ? 0u : gy < 0.f ? 1u : gx > 0.f ? 2u : 3u;
      if (best > 0.f) flags |= 1u << 4 | arg << 5;
    }
    const bool clipped = clip_grad(g, gx, gy);
    if constexpr (DUMP) flags |= (clipped ? 1u : 0u) << 9;
    wsx = g.w_coll * gx; wsy = g.w_coll * gy;
  }
  // --- CostGPTrajectory (cost_functions.py:532-542, gp_factor.py): e_t = s_{t+1} - Phi s_t, w_t = 2 Q^-1 e_t,
  //     g_t = w_{t-1} - Phi^T w_t
  float gpx, gpy, gpz, gpw;
  {
    const float npx = lane_next(px), npy = lane_next(py), nvx = lane_next(vx), nvy = lane_next(vy);
    float wpx = 0.f, wpy = 0.f, wvx = 0.f, wvy = 0.f;
    if (t < H - 1) {
      const float epx = npx - (px + g.dt * vx), epy = npy - (py + g.dt * vy);
      const float evx = nvx - vx, evy = nvy - vy;
      wpx = 2.f * (g.m1 * epx + g.m2 * evx);
      wpy = 2.f * (g.m1 * epy + g.m2 * evy);
      wvx = 2.f * (g.m2 * epx + g.m3 * evx);
      wvy = 2.f * (g.m2 * epy + g.m3 * evy);
    }
    const float lpx = lane_prev(wpx), lpy = lane_prev(wpy), lvx = lane_prev(wvx), lvy = lane_prev(wvy);
    float gx = lpx - wpx, gy = lpy - wpy;
    float gz = lvx - (g.dt * wpx + wvx), gw = lvy - (g.dt * wpy + wvy);
    const bool clipped = clip_grad(g, gx, gy, gz, gw);
    if constexpr (DUMP) flags |= (clipped ? 1u : 0u) << 10;
    gpx = g.w_smooth * gx; gpy = g.w_smooth * gy;
    gpz = g.w_smooth * gz; gpw = g.w_smooth * gw;
  }
  // --- CostConstraint groups (cost_functions.py:297-326): per group the slot sum, its own clip and weight
  float cx = 0.f, cy = 0.f;
  for (int grp = grp0; grp < grp1; ++grp) {
    const f32x2g gs = group_sum(grp, f32x2g{px, py});
    float gx = gs.x, gy = gs.y;
    const bool clipped = clip_grad(g, gx, gy);
    if constexpr (DUMP) {
      const int k = grp - grp0, s0 = g.grp_slot_off[grp], s1 = g.grp_slot_off[grp + 1];
      unsigned long long mask = 0;
      unsigned int n_act = 0;
      for (int sl = s0; sl < s1; ++sl) {
        const float4 c = g.cons[(size_t)sl * H + t];
        const float dx = px - c.x, dy = py - c.y;
        const float d2 = __builtin_fmaf(dx, dx, dy * dy);
        if (!(d2 > c.w)) {
          ++n_act;
          if (sl - s0 < 64) mask |= 1ull << (sl - s0);
        }
      }
      if (k < 4) {
        tr[2 + 2 * k] = (unsigned int)mask;
        tr[3 + 2 * k] = (unsigned int)(mask >> 32);
        flags |= (clipped ? 1u : 0u) << (11 + k);
      }
      tr[10] += n_act;
    }
    const float w = group_weight(grp);
    cx += w * gx; cy += w * gy;
  }
  // --- CostCollision over the SDF grids: d/dp max_k relu(margin - sdf_k(p))  (t >= 1; field_factor.py range [1,None])
  float ox, oy;
  {
    float best = fmaxf(g.margin - cell0.x, 0.f), gx = 0.f, gy = 0.f;
    unsigned win = 0;
    if (best > 0.f) { gx = -cell0.y; gy = -cell0.z; }
    for (int k = 1; k < g.n_grids; ++k) {
      const float4 c = grid[((size_t)k * g.nx + ix) * g.ny + iy];
      const float v = fmaxf(g.margin - c.x, 0.f);
      if (v > best) { best = v; gx = -c.y; gy = -c.z; win = (unsigned)k; }
    }
    if (g.n_xs + g.n_xb > 0) {                             // the env's extra objects: one more field, analytic
      float ex, ey;
      const float v = fmaxf(g.margin - extra_sdf(g.xs, g.n_xs, g.xb, g.n_xb, px, py, ex, ey), 0.f);
      if (v > best) { best = v; gx = -ex; gy = -ey; win = 7u; }
    }
    const bool clipped = clip_grad(g, gx, gy);
    if constexpr (DUMP) {
      if (best > 0.f) flags |= 1u | (win & 7u) << 1;
      flags |= (clipped ? 1u : 0u) << 8;
    }
    ox = g.w_coll * gx; oy = g.w_coll * gy;
  }
  if constexpr (DUMP) {
    tr[0] = (unsigned int)(ix * g.ny + iy);
    tr[1] = flags | (unsigned int)min(grp1 - grp0, 15) << 20;
  }
  // sum in the reference's cost order (objects, ws boundaries, GP, constraints), zero rows 0 / H-1, negate
  float tx = ((ox + wsx) + gpx) + cx, ty = ((oy + wsy) + gpy) + cy;
  float tz = gpz, tw = gpw;
  if (!interior) { tx = ty = tz = tw = 0.f; }
  return make_float4(-tx, -ty, -tz, -tw);
}

// One ddpm_sample_fn (sample_functions.py:40-86) + the apply_hard_conditioning after it, for one trajectory per wave.
template <int WPB, bool COMPACT, bool DUMP = false>
__global__ __launch_bounds__(WPB * 64) void ddpm_guide_kernel(GuideDev g, StepDev s, int lds_slots, float4* __restrict__ x,
                                                         const float4* __restrict__ eps,
                                                         const float4* __restrict__ noise, float4* __restrict__ chain,
                                                         const float4* __restrict__ hard,
                                                         int samples_per_robot) {
  extern __shared__ __attribute__((aligned(16))) float4 lds_cons[];
#ifdef MMD_GUIDE_PRIO   // (A/B build: the step kernel is on the critical chain of its stream chunk while the other chunk's UNet launch shares the SIMDs)
  __builtin_amdgcn_s_setprio(MMD_GUIDE_PRIO);
#endif
#ifdef MMD_GUIDE_PLAIN   // (A/B side build: the configuration every planner runs -- clip by norm, no extra objects, at most one grid -- folded in)
  g.clip_rule = 0; g.n_xs = 0; g.n_xb = 0; g.n_grids = g.n_grids > 0 ? 1 : 0;
#endif
  const int t = threadIdx.x & 63;
  const int traj_b = s.traj0 + blockIdx.x * WPB;
  const int traj = traj_b + (threadIdx.x >> 6);
  const bool valid = traj < s.traj_end;

  // constraint table of the workgroup's robot -> LDS (the 4 trajectories of a workgroup share a robot whenever
  // samples_per_robot is a multiple of 4; otherwise every wave reads the L2-resident table directly)
  int lds_slot0 = 0, lds_n = 0;
  if (s.do_guide && g.robot_grp_off) {
    const int rb0 = traj_b / samples_per_robot;
    const int rb1 = min(traj_b + WPB - 1, s.traj_end - 1) / samples_per_robot;
    if (rb0 == rb1) {
      lds_slot0 = __builtin_amdgcn_readfirstlane(g.grp_slot_off[g.robot_grp_off[rb0]]);
      lds_n = min(__builtin_amdgcn_readfirstlane(g.grp_slot_off[g.robot_grp_off[rb0 + 1]]) - lds_slot0, lds_slots);
      const float4* src = g.cons + (size_t)lds_slot0 * H;
      if constexpr (COMPACT) {
        float2* dst = reinterpret_cast<float2*>(lds_cons);
        for (int i = threadIdx.x; i < lds_n * H; i += WPB * 64) {
          const float4 c = src[i];
          dst[i] = c.z < 0.f ? make_float2(1e30f, 1e30f) : make_float2(c.x, c.y);
        }
      } else {
        for (int i = threadIdx.x; i < lds_n * H; i += WPB * 64) lds_cons[i] = src[i];
      }
    }
  }
  __syncthreads();
  if (!valid) return;

  const int robot = traj / samples_per_robot;
  const size_t idx = (size_t)traj * H + t;
  float4 v = x[idx];

  if (s.do_model)
    v = s.ddim == 2 ? ddim_update_x0(v, eps[idx], s.a_t, s.b_t, s.c1, s.c2)
        : s.ddim ? ddim_update(v, eps[idx], s.a_t, s.b_t, s.c1, s.c2) : ddpm_posterior_mean(v, eps[idx], s.a_t, s.b_t, s.c1, s.c2);

  float4 hv = v;
  const bool is_hard = hard_row(s.hard_rows, s.n_hard, hard, robot, t, hv);
  if constexpr (DUMP) {
    // (the posterior mean as the first guide evaluation sees it: rows 0 / H - 1 are pinned AFTER every iteration, not before the
    // first one -- guide_gradient_steps, sample_functions.py:89-107)
    if (s.mu_out) s.mu_out[idx] = v;
  }

  if (s.do_guide) {
    const int map = g.robot_map ? g.robot_map[robot] : 0;
    const float4* grid = g.grids + (size_t)map * g.n_grids * g.nx * g.ny;
    int grp0 = 0, grp1 = 0;
    if (g.robot_grp_off) {
      grp0 = __builtin_amdgcn_readfirstlane(g.robot_grp_off[robot]);
      grp1 = __builtin_amdgcn_readfirstlane(g.robot_grp_off[robot + 1]);
    }
    GroupMeta gm;
    gm.load(g, grp0, grp1);
    auto group_weight = [&](int grp) { return gm.weight(g, grp); };
    // the group's slots: LDS-resident ones first, any overflow straight from the L2-resident table
    auto group_sum = [&](int grp, f32x2g p) {
      // (group bounds are the same for the whole wave: scalar registers, scalar loop control)
      int s0, s1;
      gm.bounds(g, grp, s0, s1);
      const int l0 = min(max(s0 - lds_slot0, 0), lds_n), l1 = min(max(s1 - lds_slot0, 0), lds_n);   // LDS part
      Acc4 acc = acc4_zero();
      if (l1 > l0) {
        if constexpr (COMPACT)
          cons_accumulate4(acc, ConsTab<true>{reinterpret_cast<const float2*>(lds_cons), g.uniform_r2}, l0, l1 - l0,
                           lds_slot0 + l0 - s0, t, p);
        else
          cons_accumulate4(acc, ConsTab<false>{lds_cons, 0.f}, l0, l1 - l0, lds_slot0 + l0 - s0, t, p);
      }
      const int g0 = lds_n > 0 ? max(s0, lds_slot0 + lds_n) : s0;                                    // global part
      if (s1 > g0) cons_accumulate4(acc, ConsTab<false>{g.cons, 0.f}, g0, s1 - g0, g0 - s0, t, p);
      return acc4_total(acc);
    };
    for (int it = 0; it < s.n_guide_steps; ++it) {
      unsigned int* tr = nullptr;
      if constexpr (DUMP) {
        tr = s.trace + ((size_t)it * s.guide_chain_stride + idx) * MMD_TRACE_WORDS;
        for (int w = 0; w < MMD_TRACE_WORDS; ++w) tr[w] = 0u;
      }
      const float4 gr = guide_grad<DUMP>(g, v, t, grid, grp0, grp1, group_sum, group_weight, tr);
      // (x + model_var * grad with scale_grad_by_std, sample_functions.py:100-104; grad_scale = 1 otherwise: the fma is then the add)
      v.x = __builtin_fmaf(s.grad_scale, gr.x, v.x); v.y = __builtin_fmaf(s.grad_scale, gr.y, v.y);
      v.z = __builtin_fmaf(s.grad_scale, gr.z, v.z); v.w = __builtin_fmaf(s.grad_scale, gr.w, v.w);
      if (is_hard) v = hv;
      if (s.guide_chain) s.guide_chain[(size_t)it * s.guide_chain_stride + idx] = v;
    }
  }

  if (s.do_noise) v = add_step_noise(v, noise ? noise[idx] : traj_normal4(s.seed, s.robot_seeds, s.draw, s.traj_base, idx, robot, samples_per_robot), s.sigma, s.noise_std_extra);
  if (is_hard) v = hv;
  x[idx] = v;
  if (chain) chain[idx] = v;
}

// The same step with FOUR waves per trajectory (small launches: <= 512 trajectories leave most SIMDs without a wave, and a
// wave's 20 dependent guide iterations of ~600 instructions are pure latency).  Every wave carries the whole trajectory (lane =
// support point) and computes everything but the constraint sums redundantly -- identical instructions, identical bits, so the
// four copies of v never diverge; wave k evaluates accumulator k of every group's slot sum (a quarter of the slots), the four
// partials meet in LDS (one barrier per group and iteration, double-buffered by parity) and every wave combines them in the
// canonical order.  Wave 0 stores.  Bitwise equal to ddpm_guide_kernel.
constexpr int COOP_XCH_BYTES = 2 * 4 * H * 8;
template <bool COMPACT>
__global__ __launch_bounds__(256) void ddpm_guide_coop_kernel(GuideDev g, StepDev s, int lds_slots, float4* __restrict__ x,
                                                              const float4* __restrict__ eps, const float4* __restrict__ noise,
                                                              float4* __restrict__ chain, const float4* __restrict__ hard,
                                                              int samples_per_robot) {
  extern __shared__ __attribute__((aligned(16))) float4 lds_cons[];
#ifdef MMD_GUIDE_PLAIN   // (A/B side build: the configuration every planner runs -- clip by norm, no extra objects, at most one grid -- folded in)
  g.clip_rule = 0; g.n_xs = 0; g.n_xb = 0; g.n_grids = g.n_grids > 0 ? 1 : 0;
#endif
  const int t = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int traj = s.traj0 + blockIdx.x;                   // (the grid is exactly the launch's trajectories)
  const int robot = traj / samples_per_robot;
  float2* const xch = reinterpret_cast<float2*>(reinterpret_cast<char*>(lds_cons) + (size_t)lds_slots * H * (COMPACT ? 8 : 16));
  int lds_slot0 = 0, lds_n = 0;
  if (s.do_guide && g.robot_grp_off) {
    lds_slot0 = __builtin_amdgcn_readfirstlane(g.grp_slot_off[g.robot_grp_off[robot]]);
    lds_n = min(__builtin_amdgcn_readfirstlane(g.grp_slot_off[g.robot_grp_off[robot + 1]]) - lds_slot0, lds_slots);
    const float4* src = g.cons + (size_t)lds_slot0 * H;
    if constexpr (COMPACT) {
      float2* dst = reinterpret_cast<float2*>(lds_cons);
      for (int i = threadIdx.x; i < lds_n * H; i += 256) {
        const float4 c = src[i];
        dst[i] = c.z < 0.f ? make_float2(1e30f, 1e30f) : make_float2(c.x, c.y);
      }
    } else {
      for (int i = threadIdx.x; i < lds_n * H; i += 256) lds_cons[i] = src[i];
    }
  }
  __syncthreads();
  const size_t idx = (size_t)traj * H + t;
  float4 v = x[idx];
  if (s.do_model)
    v = s.ddim == 2 ? ddim_update_x0(v, eps[idx], s.a_t, s.b_t, s.c1, s.c2)
        : s.ddim ? ddim_update(v, eps[idx], s.a_t, s.b_t, s.c1, s.c2) : ddpm_posterior_mean(v, eps[idx], s.a_t, s.b_t, s.c1, s.c2);
  float4 hv = v;
  const bool is_hard = hard_row(s.hard_rows, s.n_hard, hard, robot, t, hv);
  if (s.do_guide) {
    const int map = g.robot_map ? g.robot_map[robot] : 0;
    const float4* grid = g.grids + (size_t)map * g.n_grids * g.nx * g.ny;
    int grp0 = 0, grp1 = 0;
    if (g.robot_grp_off) {
      grp0 = __builtin_amdgcn_readfirstlane(g.robot_grp_off[robot]);
      grp1 = __builtin_amdgcn_readfirstlane(g.robot_grp_off[robot + 1]);
    }
    int parity = 0;
    GroupMeta gm;
    gm.load(g, grp0, grp1);
    auto group_weight = [&](int grp) { return gm.weight(g, grp); };
    auto group_sum = [&](int grp, f32x2g p) {
      int s0, s1;
      gm.bounds(g, grp, s0, s1);
      const int l0 = min(max(s0 - lds_slot0, 0), lds_n), l1 = min(max(s1 - lds_slot0, 0), lds_n);
      f32x2g a = {0.f, 0.f};
      if (l1 > l0) {
        if constexpr (COMPACT)
          a = cons_accumulate1(a, ConsTab<true>{reinterpret_cast<const float2*>(lds_cons), g.uniform_r2}, l0, l1 - l0,
                               lds_slot0 + l0 - s0, wave, t, p);
        else
          a = cons_accumulate1(a, ConsTab<false>{lds_cons, 0.f}, l0, l1 - l0, lds_slot0 + l0 - s0, wave, t, p);
      }
      const int g0 = lds_n > 0 ? max(s0, lds_slot0 + lds_n) : s0;
      if (s1 > g0) a = cons_accumulate1(a, ConsTab<false>{g.cons, 0.f}, g0, s1 - g0, g0 - s0, wave, t, p);
      float2* const slot = xch + parity * 4 * H;
      slot[wave * H + t] = make_float2(a.x, a.y);
      __syncthreads();                                       // (the buffer of the other parity is free again: every wave has
      parity ^= 1;                                           //  passed the barrier after reading it)
      const float2 q0 = slot[t], q1 = slot[H + t], q2 = slot[2 * H + t], q3 = slot[3 * H + t];
      return acc4_total(Acc4{f32x2g{q0.x, q0.y}, f32x2g{q1.x, q1.y}, f32x2g{q2.x, q2.y}, f32x2g{q3.x, q3.y}});
    };
    for (int it = 0; it < s.n_guide_steps; ++it) {
      const float4 gr = guide_grad(g, v, t, grid, grp0, grp1, group_sum, group_weight);
      // (x + model_var * grad with scale_grad_by_std, sample_functions.py:100-104; grad_scale = 1 otherwise: the fma is then the add)
      v.x = __builtin_fmaf(s.grad_scale, gr.x, v.x); v.y = __builtin_fmaf(s.grad_scale, gr.y, v.y);
      v.z = __builtin_fmaf(s.grad_scale, gr.z, v.z); v.w = __builtin_fmaf(s.grad_scale, gr.w, v.w);
      if (is_hard) v = hv;
      if (s.guide_chain && wave == 0) s.guide_chain[(size_t)it * s.guide_chain_stride + idx] = v;
    }
  }
  if (wave != 0) return;
  if (s.do_noise) v = add_step_noise(v, noise ? noise[idx] : traj_normal4(s.seed, s.robot_seeds, s.draw, s.traj_base, idx, robot, samples_per_robot), s.sigma, s.noise_std_extra);
  if (is_hard) v = hv;
  x[idx] = v;
  if (chain) chain[idx] = v;
}

// x <- conditioned init: optional Philox draw of x_T, apply_hard_conditioning, optional chain[0] write
__global__ void init_kernel(float4* __restrict__ x, float4* __restrict__ chain, const float4* __restrict__ hard,
                            unsigned long long hard_rows, int draw_noise, unsigned long long seed,
                            const unsigned long long* __restrict__ robot_seeds, long long traj_base, int n_traj, int samples_per_robot) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n_traj * H) return;
  const int t = idx % H;
  const int robot = (idx / H) / samples_per_robot;
  float4 v = draw_noise ? traj_normal4(seed, robot_seeds, 0xFFFFFFFFu, traj_base, idx, robot, samples_per_robot) : x[idx];
  float4 hv;
  if (hard_row(hard_rows, __popcll(hard_rows), hard, robot, t, hv)) v = hv;
  x[idx] = v;
  if (chain) chain[idx] = v;
}

__global__ void q_sample_kernel(float4* __restrict__ x, const float4* __restrict__ x0, const float4* __restrict__ noise,
                                float a, float b, unsigned long long seed, unsigned int draw, long long traj_base,
                                size_t n_pts) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_pts) return;
  const float4 z = noise ? noise[idx] : normal4(seed, draw, (unsigned long long)traj_base * H + idx);
  const float4 s = x0[idx];
  x[idx] = make_float4(a * s.x + b * z.x, a * s.y + b * z.y, a * s.z + b * z.z, a * s.w + b * z.w);
}

// apply_cross_conditioning for one (m1, m2) pair (sample_functions.py:28-29); c1 / c2: the chain rows of the current
// outer step (or NULL), kept equal to x1 / x2
__global__ void cross_condition_kernel(float4* __restrict__ x1, float4* __restrict__ x2, float4* __restrict__ c1,
                                       float4* __restrict__ c2, int ind1, int ind2, float4 rel, float4 bnd,
                                       const float4* __restrict__ by_robot, int spr, int n_traj) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_traj) return;
  if (by_robot) {                         // batched planner calls: every robot has its own tile offsets
    rel = by_robot[(size_t)(b / spr) * 2];
    bnd = by_robot[(size_t)(b / spr) * 2 + 1];
  }
  const float4 v2 = x2[(size_t)b * H + ind2];
  float4 v1;
  v1.x = fminf(v2.x + rel.x, bnd.x); v1.y = fminf(v2.y + rel.y, bnd.y);
  v1.z = fminf(v2.z + rel.z, bnd.z); v1.w = fminf(v2.w + rel.w, bnd.w);
  x1[(size_t)b * H + ind1] = v1;
  float4 w;
  w.x = fmaxf(v1.x - rel.x, -bnd.x); w.y = fmaxf(v1.y - rel.y, -bnd.y);
  w.z = fmaxf(v1.z - rel.z, -bnd.z); w.w = fmaxf(v1.w - rel.w, -bnd.w);
  x2[(size_t)b * H + ind2] = w;
  if (c1) c1[(size_t)b * H + ind1] = v1;
  if (c2) c2[(size_t)b * H + ind2] = w;
}

// all-pairs soft constraints from best paths (cbs.py:468-508): slot j of local robot i = other robot j (+1 past i)
__global__ void soft_cons_kernel(const float2* __restrict__ paths, int n_all, int robot0, int n_local, float radius,
                                 float weight, float4* __restrict__ ell, int* __restrict__ grp_slot_off,
                                 float* __restrict__ grp_weight, int* __restrict__ robot_grp_off) {
  const int slots = n_all - 1;
  const size_t tot = (size_t)n_local * slots * H;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
    const int t = idx % H;
    const int j = (idx / H) % slots;
    const int i = idx / ((size_t)H * slots);
    const int other = j + (j >= robot0 + i ? 1 : 0);
    const float2 p = paths[(size_t)other * H + t];
    const float r = t >= 1 ? radius : -1.f;                               // constraints cover t in [1, H-1]
    ell[idx] = make_float4(p.x, p.y, r, r * fabsf(r));
  }
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid <= n_local) {
    grp_slot_off[tid] = tid * slots;
    robot_grp_off[tid] = tid;
    if (tid < n_local) grp_weight[tid] = weight;
  }
}

int fill_guide(const mmd_guide_desc* d, GuideDev& g) {
  MMD_REQUIRE(d->n_grids == 0 || (d->sdf_grids_dev && d->grid_nx >= 1 && d->grid_ny >= 1), "guide: SDF grid missing");
  for (int k = 0; k < 4; ++k) { g.nmin[k] = d->norm_min[k]; g.nscale[k] = d->norm_max[k] - d->norm_min[k]; }
  for (int k = 0; k < 2; ++k) {
    g.lo[k] = d->limits_lo[k];
    g.dim[k] = fabsf(d->limits_hi[k] - d->limits_lo[k]);
    g.inv_dim[k] = 1.f / g.dim[k];
    g.ws_min[k] = d->ws_min[k]; g.ws_max[k] = d->ws_max[k];
  }
  g.nx = d->grid_nx; g.ny = d->grid_ny; g.n_grids = d->n_grids;
  g.grids = reinterpret_cast<const float4*>(d->sdf_grids_dev);
  g.robot_map = d->robot_map_dev;
  g.margin = d->margin; g.dt = d->dt; g.w_coll = d->weight_collision; g.w_smooth = d->weight_smoothness;
  g.max_norm = d->max_grad_norm;
  MMD_REQUIRE(d->clip_grad_rule >= 0 && d->clip_grad_rule <= 2, "guide: clip_grad_rule must be 0 (norm), 1 (value) or 2 (off)");
  g.clip_rule = d->clip_grad_rule;
  g.max_value = d->max_grad_value;
  const double dt = d->dt, qc = 1.0 / ((double)d->sigma_gp * d->sigma_gp);
  g.m1 = (float)(12.0 / (dt * dt * dt) * qc);
  g.m2 = (float)(-6.0 / (dt * dt) * qc);
  g.m3 = (float)(4.0 / dt * qc);
  g.max_slots = d->max_slots_per_robot > 0 ? d->max_slots_per_robot : LDS_SLOTS_SMALL;
  g.uniform_r2 = d->cons_uniform_radius > 0.f ? d->cons_uniform_radius * fabsf(d->cons_uniform_radius) : 0.f;
  g.xs = reinterpret_cast<const float4*>(d->extra_spheres_dev); g.n_xs = d->extra_spheres_dev ? d->n_extra_spheres : 0;
  g.xb = reinterpret_cast<const float4*>(d->extra_boxes_dev); g.n_xb = d->extra_boxes_dev ? d->n_extra_boxes : 0;
  g.cons = reinterpret_cast<const float4*>(d->cons_ell_dev);
  g.grp_slot_off = d->grp_slot_off_dev; g.grp_weight = d->grp_weight_dev; g.robot_grp_off = d->robot_grp_off_dev;
  if (!g.cons || !g.grp_slot_off || !g.grp_weight) g.robot_grp_off = nullptr;
  return 0;
}

// the launch size up to which a guided step runs four waves per trajectory (mmd_sampler_desc.guide_coop_max overrides: A/B)
constexpr int kCoopMaxTrajDefault = 512;

int launch_step(const GuideDev& g, StepDev s, float* x, const float* eps, const float* noise, float* chain,
                const float* hard, int traj0, int n_traj, int spr, hipStream_t st) {
  s.traj0 = traj0;
  s.traj_end = traj0 + n_traj;
  const bool guided = s.do_guide && g.robot_grp_off;
  // LDS staging of the workgroup's robot's table: 16 B per (slot, t), or 8 B when every active point has the same radius
  const bool compact = guided && g.uniform_r2 > 0.f;
  const int bytes_per_slot = H * (compact ? 8 : 16);
  const int small = LDS_SLOTS_SMALL * (compact ? 2 : 1), big = LDS_SLOTS_MAX * (compact ? 2 : 1);
  auto launch = [&](auto kern, int wpb, int slots) {
    hipLaunchKernelGGL(kern, dim3((n_traj + wpb - 1) / wpb), dim3(wpb * 64), (size_t)slots * bytes_per_slot, st, g, s, slots,
                       (float4*)x, (const float4*)eps, (const float4*)noise, (float4*)chain, (const float4*)hard, spr);
  };
  if (s.trace) {
    // measurement hook: the one-wave kernel, every slot from the L2-resident table (the slot sums are canonical: same bits as any
    // production launch shape), with the decision dump compiled in
    hipLaunchKernelGGL((ddpm_guide_kernel<4, false, true>), dim3((n_traj + 3) / 4), dim3(256), 0, st, g, s, 0, (float4*)x,
                       (const float4*)eps, (const float4*)noise, (float4*)chain, (const float4*)hard, spr);
  } else if (guided && n_traj <= (s.coop_max > 0 ? s.coop_max : s.coop_max < 0 ? 0 : kCoopMaxTrajDefault)) {
    // four waves per trajectory (ddpm_guide_coop_kernel): the whole table of the trajectory's robot in LDS up to 60 KiB (+ the
    // exchange buffer: inside the 64 KiB a launch gets without an opt-in), the rest from L2
    const int fit = 60 * 1024 / bytes_per_slot;
    const int slots = g.max_slots < fit ? g.max_slots : fit;
    auto kern = compact ? ddpm_guide_coop_kernel<true> : ddpm_guide_coop_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(n_traj), dim3(256), (size_t)slots * bytes_per_slot + COOP_XCH_BYTES, st, g, s, slots, (float4*)x,
                       (const float4*)eps, (const float4*)noise, (float4*)chain, (const float4*)hard, spr);
  } else if (guided && g.max_slots > small && spr % 8 == 0 && traj0 % 8 == 0) {
    // 8 trajectories of one robot per workgroup (2048 trajectories = 256 workgroups = one per CU); LDS sized to the
    // largest table any robot can have (g.max_slots), the rest of a larger table is read from L2
    const int slots = g.max_slots < big ? g.max_slots : big;
    // the 144 KiB dynamic-LDS opt-in is a per-device function attribute: set it once for every device this process
    // launches on (two threads racing on the same device both set the same value, which is harmless)
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpm_guide_kernel<8, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_SLOTS_MAX * H * 16);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ddpm_guide_kernel<8, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_SLOTS_MAX * H * 16);
      attr_devices.fetch_or(bit, std::memory_order_release);
    }
    if (compact) launch(ddpm_guide_kernel<8, true>, 8, slots);
    else launch(ddpm_guide_kernel<8, false>, 8, slots);
  } else {
    const int slots = guided ? small : 0;
    if (compact) launch(ddpm_guide_kernel<4, true>, 4, slots);
    else launch(ddpm_guide_kernel<4, false>, 4, slots);
  }
  return 0;
}

void launch_cross(float* x1, float* x2, float* c1, float* c2, int ind1, int ind2, const float* rel, const float* bnd,
                  const float* by_robot, int spr, int n_traj, hipStream_t st) {
  hipLaunchKernelGGL(cross_condition_kernel, dim3((n_traj + 255) / 256), dim3(256), 0, st, (float4*)x1, (float4*)x2,
                     (float4*)c1, (float4*)c2, ind1, ind2, make_float4(rel[0], rel[1], rel[2], rel[3]),
                     make_float4(bnd[0], bnd[1], bnd[2], bnd[3]), (const float4*)by_robot, spr > 0 ? spr : 1, n_traj);
}

int launch_init(float* x, float* chain, const float* hard, unsigned long long hard_rows, int draw, unsigned long long seed,
                const unsigned long long* robot_seeds, long long traj_base, int n_traj, int spr, hipStream_t st) {
  const size_t n = (size_t)n_traj * H;
  hipLaunchKernelGGL(init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (float4*)x, (float4*)chain,
                     (const float4*)hard, hard_rows, draw, seed, robot_seeds, traj_base, n_traj, spr);
  return 0;
}

}  // namespace mmd

using namespace mmd;

extern "C" {

int mmd_pack_constraints(int n_groups, const int32_t* n_pts, const float* const* q, const float* const* t_range,
                         const float* const* radius, int horizon, float* ell_out, int max_slots, int32_t* slots_out) {
  MMD_REQUIRE(horizon == H, "mmd_pack_constraints: horizon must be %d", H);
  int used = 0;
  for (int g = 0; g < n_groups; ++g) {
    // a point is active at integer t iff t >= t0 and t < t1 (float ranges, cost_functions.py:304-305)
    std::vector<int> fill(H, 0);
    for (int c = 0; c < n_pts[g]; ++c) {
      const int t0 = (int)ceilf(t_range[g][2 * c]), t1 = (int)ceilf(t_range[g][2 * c + 1]);
      for (int t = t0 < 0 ? 0 : t0; t < t1 && t < H; ++t) ++fill[t];
    }
    int slots = 0;
    for (int t = 0; t < H; ++t) slots = fill[t] > slots ? fill[t] : slots;
    slots_out[g] = slots;
    if (!ell_out) { used += slots; continue; }      // sizing pass
    MMD_REQUIRE(used + slots <= max_slots, "mmd_pack_constraints: need more than %d slots", max_slots);
    for (int s = 0; s < slots; ++s)
      for (int t = 0; t < H; ++t) {
        float* e = ell_out + ((size_t)(used + s) * H + t) * 4;
        e[0] = 0.f; e[1] = 0.f; e[2] = -1.f; e[3] = -1.f;
      }
    std::fill(fill.begin(), fill.end(), 0);
    for (int c = 0; c < n_pts[g]; ++c) {
      const int t0 = (int)ceilf(t_range[g][2 * c]), t1 = (int)ceilf(t_range[g][2 * c + 1]);
      for (int t = t0 < 0 ? 0 : t0; t < t1 && t < H; ++t) {
        float* e = ell_out + ((size_t)(used + fill[t]) * H + t) * 4;
        e[0] = q[g][2 * c]; e[1] = q[g][2 * c + 1]; e[2] = radius[g][c];
        e[3] = radius[g][c] * fabsf(radius[g][c]);
        ++fill[t];
      }
    }
    used += slots;
  }
  return 0;
}

int mmd_soft_constraints_from_paths(const float* paths_dev, int n_all, int robot0, int n_local, int horizon,
                                    float radius, float weight, float* ell_out_dev, int32_t* grp_slot_off_dev,
                                    float* grp_weight_dev, int32_t* robot_grp_off_dev, void* stream) {
  MMD_REQUIRE(horizon == H, "horizon must be %d", H);
  MMD_REQUIRE(n_all >= 2 && n_local >= 1 && robot0 >= 0 && robot0 + n_local <= n_all, "bad robot range");
  const size_t tot = (size_t)n_local * (n_all - 1) * H;
  int grid = (int)((tot + 255) / 256);
  if (grid > 2048) grid = 2048;
  if (grid * 256 < n_local + 1) grid = (n_local + 1 + 255) / 256;
  hipLaunchKernelGGL(soft_cons_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float2*)paths_dev, n_all,
                     robot0, n_local, radius, weight, (float4*)ell_out_dev, grp_slot_off_dev, grp_weight_dev,
                     robot_grp_off_dev);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_guide_steps(const mmd_guide_desc* d, float* x_dev, const float* hard_dev, uint64_t hard_rows, int n_robots,
                    int samples_per_robot, int n_steps, float* chain_dev, void* stream) {
  MMD_REQUIRE(d && x_dev && hard_dev, "mmd_guide_steps: NULL argument");
  GuideDev g{};
  if (int rc = fill_guide(d, g)) return rc;
  StepDev s{};
  s.do_guide = 1; s.n_guide_steps = n_steps; s.hard_rows = hard_rows; s.n_hard = __builtin_popcountll(hard_rows); s.grad_scale = 1.f;
  s.guide_chain = reinterpret_cast<float4*>(chain_dev);
  s.guide_chain_stride = (long long)n_robots * samples_per_robot * H;
  launch_step(g, s, x_dev, nullptr, nullptr, nullptr, hard_dev, 0, n_robots * samples_per_robot, samples_per_robot,
              (hipStream_t)stream);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_q_sample(float* x_dev, const float* x_start_dev, const float* noise_dev, float a, float b, uint64_t seed,
                 uint32_t draw_index, int64_t traj_index_base, int n_traj, void* stream) {
  MMD_REQUIRE(x_dev && x_start_dev && n_traj >= 1, "mmd_q_sample: bad arguments");
  const size_t n = (size_t)n_traj * H;
  hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (float4*)x_dev, (const float4*)x_start_dev, (const float4*)noise_dev, a, b,
                     (unsigned long long)seed, draw_index, (long long)traj_index_base, n);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_cross_condition(float* x1_dev, float* x2_dev, int ind1, int ind2, const float* rel, const float* boundary,
                        int n_traj, void* stream) {
  MMD_REQUIRE(x1_dev && x2_dev && rel && boundary, "mmd_cross_condition: NULL argument");
  MMD_REQUIRE(ind1 >= 0 && ind1 < H && ind2 >= 0 && ind2 < H, "row index out of range");
  launch_cross(x1_dev, x2_dev, nullptr, nullptr, ind1, ind2, rel, boundary, nullptr, 1, n_traj, (hipStream_t)stream);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
