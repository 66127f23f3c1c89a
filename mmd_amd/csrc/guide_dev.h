// Device-side parameter blocks of the guide / DDPM-step kernel, shared by guide.hip and api.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mmd_amd.h"
#include "common.h"

namespace mmd {

struct GuideDev {
  float nmin[4], nscale[4];          // x_u = ((clip(x)+1)/2) * nscale + nmin,  nscale = max - min
  float lo[2], inv_dim[2];           // grid index = floor((p - lo) * inv_dim?  -- see sdf_cell: kept as division
  float dim[2];
  int nx, ny, n_grids;
  const float4* grids;               // [n_maps][n_grids][nx][ny]
  const int* robot_map;
  float ws_min[2], ws_max[2];
  float margin, dt, w_coll, w_smooth, max_norm;
  int clip_rule;                     // 0: clip_grad_by_norm (max_norm), 1: clip_grad_by_value (max_value), 2: clip_grad = False
  float max_value;
  float m1, m2, m3;                  // GP prior Q^-1 blocks: 12/dt^3, -6/dt^2, 4/dt  (x 1/sigma^2)
  const float4* cons;                // [n_slots][H]
  const int* grp_slot_off;
  const float* grp_weight;
  const int* robot_grp_off;
  int max_slots;                     // largest number of constraint slots any robot owns (LDS sizing)
  float uniform_r2;                  // > 0: every active point has radius^2 = this (compact float2 staging); else 0
  const float4* xs;                  // extra spheres (cx, cy, r, 0)
  const float4* xb;                  // extra boxes (cx, cy, hx, hy)
  int n_xs, n_xb;
};

// signed distance of the env's extra objects at p and its gradient (torch's sub-gradients: the first minimum / maximum);
// no object at all: sdf = 1 (an empty MultiSphereField, primitives.py:109-110)
__device__ __forceinline__ float extra_sdf(const float4* __restrict__ xs, int n_xs, const float4* __restrict__ xb, int n_xb,
                                           float px, float py, float& gx, float& gy) {
  float best = n_xs + n_xb > 0 ? 1e30f : 1.f;
  gx = 0.f; gy = 0.f;
  for (int i = 0; i < n_xs; ++i) {
    const float4 s = xs[i];
    const float dx = px - s.x, dy = py - s.y;
    const float n = sqrtf(dx * dx + dy * dy), d = n - s.z;
    if (d < best) { best = d; gx = n > 0.f ? dx / n : 0.f; gy = n > 0.f ? dy / n : 0.f; }
  }
  for (int i = 0; i < n_xb; ++i) {
    // MultiBoxField (an alias of MultiRoundedBoxField, primitives.py:345) = the rounded box of the fixed objects (primitives.py:326-333): q = |p - c| - half + rad, sdf = min(max q,
    // 0) + ||relu(q)|| - rad; b = (cx, cy, half x, half y), rad = 0.15 x the smaller SIZE
    const float4 b = xb[i];
    const float rad = 0.3f * fminf(b.z, b.w);
    const float dx = px - b.x, dy = py - b.y, qx = fabsf(dx) - b.z + rad, qy = fabsf(dy) - b.w + rad;
    const float mq = fmaxf(qx, qy), rx = fmaxf(qx, 0.f), ry = fmaxf(qy, 0.f), n = sqrtf(rx * rx + ry * ry);
    const float d = fminf(mq, 0.f) + n - rad;
    if (d < best) {
      best = d;
      const float sx = dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f), sy = dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f);
      if (mq <= 0.f) {                                     // inside the inner box: d = max q - rad
        gx = qx >= qy ? sx : 0.f;
        gy = qx >= qy ? 0.f : sy;
      } else {                                             // outside: d = ||relu(q)|| - rad
        gx = rx / n * sx;
        gy = ry / n * sy;
      }
    }
  }
  return best;
}

// ---- Philox4x32-10 + Box-Muller -------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0;
    const unsigned int n1 = (unsigned int)p1;
    const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1;
    const unsigned int n3 = (unsigned int)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// counter = (GLOBAL point index lo, draw, GLOBAL point index hi, 0), key = seed: a (robot, sample, t) point draws the same
// noise whichever rank / stream chunk / batch position it is sampled in
__device__ __forceinline__ float4 normal4(unsigned long long seed, unsigned int draw, unsigned long long point) {
  unsigned int c[4] = {(unsigned int)point, draw, (unsigned int)(point >> 32), 0u};
  philox4x32(c, (unsigned int)seed, (unsigned int)(seed >> 32));
  const float s = 2.3283064365386963e-10f;   // 2^-32
  const float u0 = ((float)c[0] + 0.5f) * s, u1 = ((float)c[1] + 0.5f) * s;
  const float u2 = ((float)c[2] + 0.5f) * s, u3 = ((float)c[3] + 0.5f) * s;
  const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u1, &s0, &c0);
  sincosf(6.283185307179586f * u3, &s1, &c1);
  return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}

// The Philox stream of one (trajectory, support point) of a sampling call.  Default: ONE stream per call, keyed by the global
// trajectory index (traj_base + index in the arrays).  robot_seeds != NULL (mmd_sampler_desc.robot_seeds_dev: R independent planner
// calls batched into one launch sequence): one stream per ROBOT, keyed by the index within the robot -- exactly what R separate calls
// with those seeds (and traj_index_base 0) draw.  idx = index of the point in the arrays (trajectory * H + t).
__device__ __forceinline__ float4 traj_normal4(unsigned long long seed, const unsigned long long* robot_seeds, unsigned int draw,
                                               long long traj_base, size_t idx, int robot, int spr) {
  if (robot_seeds) return normal4(robot_seeds[robot], draw, (unsigned long long)(idx - (size_t)robot * spr * H));
  return normal4(seed, draw, (unsigned long long)traj_base * H + idx);
}

// p_mean_variance (diffusion_model_base.py:148-160): x0 = a x - b eps; clamp; mean = c1 x0 + c2 x.  Explicit fmas: the step
// kernel and the UNet kernel's fused unguided step (unet.hip) must round identically.
__device__ __forceinline__ float ddpm_mean1(float x, float e, float a, float b, float c1, float c2) {
  const float x0 = fminf(fmaxf(__builtin_fmaf(a, x, -(b * e)), -1.f), 1.f);
  return __builtin_fmaf(c1, x0, c2 * x);
}
__device__ __forceinline__ float4 ddpm_posterior_mean(float4 v, float4 e, float a, float b, float c1, float c2) {
  return make_float4(ddpm_mean1(v.x, e.x, a, b, c1, c2), ddpm_mean1(v.y, e.y, a, b, c1, c2), ddpm_mean1(v.z, e.z, a, b, c1, c2),
                     ddpm_mean1(v.w, e.w, a, b, c1, c2));
}
// ddim_sample, eta = 0 (diffusion_model_base.py:245-262): x_start = a x - b eps (not clamped); x = x_start sqrt(alpha_next) +
// sqrt(1 - alpha_next) eps   (c1 = 1, c2 = 0 on the last pair: x = x_start)
__device__ __forceinline__ float4 ddim_update(float4 v, float4 e, float a, float b, float c1, float c2) {
  auto f = [&](float x, float ee) { return __builtin_fmaf(c1, __builtin_fmaf(a, x, -(b * ee)), c2 * ee); };
  return make_float4(f(v.x, e.x), f(v.y, e.y), f(v.z, e.z), f(v.w, e.w));
}
// the same with GaussianDiffusionModel(predict_epsilon=False): the network output o IS x_start, pred_noise = (a x - o) / b
// (predict_noise_from_start, diffusion_model_base.py:114-124); b_inv = 1 / b is passed in `b`
__device__ __forceinline__ float4 ddim_update_x0(float4 v, float4 o, float a, float b_inv, float c1, float c2) {
  auto f = [&](float x, float oo) { return __builtin_fmaf(c1, oo, c2 * (__builtin_fmaf(a, x, -oo) * b_inv)); };
  return make_float4(f(v.x, o.x), f(v.y, o.y), f(v.z, o.z), f(v.w, o.w));
}
// x + model_std * noise * noise_std  (sample_functions.py:86)
__device__ __forceinline__ float4 add_step_noise(float4 v, float4 z, float sigma, float noise_std_extra) {
  return make_float4(__builtin_fmaf(sigma * z.x, noise_std_extra, v.x), __builtin_fmaf(sigma * z.y, noise_std_extra, v.y),
                     __builtin_fmaf(sigma * z.z, noise_std_extra, v.z), __builtin_fmaf(sigma * z.w, noise_std_extra, v.w));
}

// An UNGUIDED ddpm_sample_fn step fused into the tail of the UNet launch that produces its eps (unet.hip; mmd_p_sample_loop uses
// it for every step without guidance: one launch and one dependent-dispatch bubble less per step).  Pointers are those of the full
// arrays, traj0 = first trajectory of the launch in them.
struct FusedStep {
  int enabled;
  float a_t, b_t, c1, c2, sigma, noise_std_extra;
  int do_noise, n_hard;
  unsigned long long hard_rows;
  unsigned long long seed;
  const unsigned long long* robot_seeds;   // or NULL (traj_normal4)
  unsigned int draw;
  long long traj_base;
  int traj0, spr;
  float4* x;
  const float4* noise;
  float4* chain;
  const float4* hard;
  int t_row;              // persistent run only: the step's row of the time table (t = max(i, 0))
};

// A persistent run of unguided steps (unet.hip: unet_persist_kernel): one complete FusedStep per step (schedule coefficients, Philox
// draw index, the step's noise / chain rows, row t of the time table) in a table at the start of the sampler workspace, the run's
// kernel-argument block behind it.
constexpr int PERSIST_MAX_STEPS = 64;
constexpr int PERSIST_ARGS_OFF = 8192;
constexpr int PERSIST_TABLE_BYTES = 12288;
static_assert(PERSIST_MAX_STEPS * sizeof(FusedStep) <= PERSIST_ARGS_OFF, "the step table fits its workspace region");

// apply_hard_conditioning (sample_functions.py:8-14: x[:, t, :] = val for every (t, val) of the hard_conds dict): bit t of `rows` set =
// support point t of every trajectory of a robot is pinned to hard[robot][slot], slot = number of pinned rows below t (the dict's
// rows in ascending order; {0, H-1} = the start / goal pair MPD passes).  Returns whether row t is pinned; hv is its value then.
__device__ __forceinline__ bool hard_row(unsigned long long rows, int n_hard, const float4* hard, int robot, int t, float4& hv) {
  const bool on = (rows >> t) & 1ull;
  if (on) hv = hard[robot * n_hard + __popcll(rows & ((1ull << t) - 1ull))];
  return on;
}

struct StepDev {
  float a_t, b_t, c1, c2;            // sqrt_recip_alphas_cumprod[t], sqrt_recipm1[t], posterior_mean_coef1/2[t]
  float sigma;                       // exp(0.5 * posterior_log_variance_clipped[t])
  float noise_std_extra;
  float grad_scale;                  // scale_grad_by_std: model_var = exp(posterior_log_variance_clipped[t]), else 1
  int do_model, do_guide, do_noise;  // do_model = 0: guide-only launch (mmd_guide_steps)
  int ddim;                          // 1: DDIM update x <- c1 * (a x - b eps) + c2 * eps, x0 not clamped (mmd_ddim_sample); 2: the
                                     // network predicts x0 (b_t holds 1 / sqrt_recipm1_alphas_cumprod[t])
  int n_guide_steps;
  int n_hard;                        // popcount(hard_rows)
  unsigned long long hard_rows;      // bit t: support point t is hard-conditioned (hard[robot][slot], slot = pinned rows below t)
  unsigned long long seed;
  const unsigned long long* robot_seeds;   // or NULL: one Philox stream per robot (traj_normal4; mmd_sampler_desc.robot_seeds_dev)
  int coop_max;                      // the cooperative guide kernel is used up to this many trajectories per launch (0 = default)
  unsigned int draw;
  int traj0, traj_end;               // this launch covers trajectories [traj0, traj_end) of the full arrays
  float4* guide_chain;               // optional [n_guide_steps][n_traj_total][H]: state after every guide iteration
  long long guide_chain_stride;      // float4 elements between consecutive iterations
  long long traj_base;               // global index of trajectory 0 of the arrays (mmd_sampler_desc.traj_index_base)
  // measurement hook (include/mmd_amd_debug.h, mmd_debug_ddpm_step_trace; NULL in every product call -- only the DUMP
  // instantiation of the step kernel reads them): the discrete decisions of every guide iteration, MMD_TRACE_WORDS uint32 per
  // (iteration, trajectory, support point) at the guide_chain stride, and the state the iterations start from
  unsigned int* trace;
  float4* mu_out;                    // optional [n_traj_total][H]: posterior mean before the first iteration (hard rows not pinned yet)
};

int fill_guide(const mmd_guide_desc* d, GuideDev& g);
// the UNet forward with the unguided step fused into its tail (unet.hip); only the fused kernel's configuration has it
bool unet_fused_step_supported(mmd_unet_t u);
int unet_forward_fused(mmd_unet_t u, const float* x, int t, float* eps, int n, void* ws, size_t ws_bytes, ::mmd_profiler_s* prof,
                       hipStream_t st, const FusedStep& fs);
int unet_persist_steps(mmd_unet_t u, int n, void* ws, size_t ws_bytes, hipStream_t st, const FusedStep* steps, int n_steps);
int launch_step(const GuideDev& g, StepDev s, float* x, const float* eps, const float* noise, float* chain,
                const float* hard, int traj0, int n_traj, int spr, hipStream_t st);
int launch_init(float* x, float* chain, const float* hard, unsigned long long hard_rows, int draw, unsigned long long seed,
                const unsigned long long* robot_seeds, long long traj_base, int n_traj, int spr, hipStream_t st);

// by_robot: optional device table [n_robots][8] = (rel[4], boundary[4]) per robot (mmd_cross_cond.by_robot_dev), spr = samples per robot
void launch_cross(float* x1, float* x2, float* c1, float* c2, int ind1, int ind2, const float* rel, const float* bnd,
                  const float* by_robot, int spr, int n_traj, hipStream_t st);

}  // namespace mmd
