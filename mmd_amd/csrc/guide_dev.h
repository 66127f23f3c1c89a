// Device-side parameter blocks of the guide / DDPM-step kernel, shared by guide.hip and api.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mmd_amd.h"

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
    // MultiBoxField = the rounded box of the fixed objects (primitives.py:326-333): q = |p - c| - half + rad, sdf = min(max q,
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

struct StepDev {
  float a_t, b_t, c1, c2;            // sqrt_recip_alphas_cumprod[t], sqrt_recipm1[t], posterior_mean_coef1/2[t]
  float sigma;                       // exp(0.5 * posterior_log_variance_clipped[t])
  float noise_std_extra;
  int do_model, do_guide, do_noise;  // do_model = 0: guide-only launch (mmd_guide_steps)
  int ddim;                          // 1: DDIM update x <- c1 * (a x - b eps) + c2 * eps, x0 not clamped (mmd_ddim_sample)
  int n_guide_steps;
  int hard_mask;
  unsigned long long seed;
  unsigned int draw;
  int traj0, traj_end;               // this launch covers trajectories [traj0, traj_end) of the full arrays
  float4* guide_chain;               // optional [n_guide_steps][n_traj_total][H]: state after every guide iteration
  long long guide_chain_stride;      // float4 elements between consecutive iterations
  long long traj_base;               // global index of trajectory 0 of the arrays (mmd_sampler_desc.traj_index_base)
};

int fill_guide(const mmd_guide_desc* d, GuideDev& g);
int launch_step(const GuideDev& g, StepDev s, float* x, const float* eps, const float* noise, float* chain,
                const float* hard, int traj0, int n_traj, int spr, hipStream_t st);
int launch_init(float* x, float* chain, const float* hard, int hard_mask, int draw, unsigned long long seed,
                long long traj_base, int n_traj, int spr, hipStream_t st);

void launch_cross(float* x1, float* x2, float* c1, float* c2, int ind1, int ind2, const float* rel, const float* bnd,
                  int n_traj, hipStream_t st);

}  // namespace mmd
