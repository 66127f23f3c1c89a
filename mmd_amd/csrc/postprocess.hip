// Post-sampling selection on the device: what MPD.__call__ does with the sampled batch right after the hot
// path (mmd/planners/single_agent/mpd.py:344-405):
//   * PlanningTask.get_trajs_collision_and_free (deps/torch_robotics/torch_robotics/tasks/tasks.py:236-311): 5 linearly
//     interpolated points per segment (trajectory/utils.py:73-86), occupancy check of each against the fixed-object SDF
//     grid (nearest cell, grid_map_sdf.py:84-114) and the workspace boundaries (distance_fields.py:318-326, :361-367)
//     with margin = robot radius (tasks.py:251-253), then the joint-limit test of the support points;
//   * compute_path_length / compute_smoothness (trajectory/metrics.py:7-39);
//   * smooth_trajs (mmd/common/trajectory_utils.py:31-40): the Savitzky-Golay filter as its precomputed [H,H] operator;
//   * compute_variance_waypoints (trajectory/metrics.py:17-27);
//   * the per-robot pick: argmin of the cost over the free samples (mpd.py:366-370) or the first free sample with the
//     fewest robot-robot collisions (cbs.py:446-458).
// One wavefront per block of H = 64 support points (a trajectory is one block; K blocks for an MPDEnsemble trajectory of K
// tiles), lane = support point: neighbours through DPP, reductions through ballots / wave shuffles, the SDF texture is
// the guide's resident one.  The interpolation and the grid index are computed with the
// reference's exact fp32 operation sequence (no FMA contraction), so the collision / free split is bit-exact.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/mmd_amd.h"
#include "common.h"
#include "guide_dev.h"

#pragma clang fp contract(off)

namespace mmd {

constexpr int MAX_INTERP = 16;

struct EnvDev {
  float lo[2], dim[2];
  int nx, ny, n_grids;
  const float4* grids;        // [n_maps][n_grids][nx][ny]
  const int* robot_map;       // per-robot map index or null
  float ws_min[2], ws_max[2];
  const float4* xs;           // the env's extra objects (mmd_guide_desc.extra_spheres_dev / extra_boxes_dev)
  const float4* xb;
  int n_xs, n_xb;
};

static int fill_env(const mmd_guide_desc* d, EnvDev& e) {
  MMD_REQUIRE(d->n_grids == 0 || (d->sdf_grids_dev && d->grid_nx >= 1 && d->grid_ny >= 1), "postprocess: SDF grid missing");
  for (int k = 0; k < 2; ++k) {
    e.lo[k] = d->limits_lo[k];
    e.dim[k] = fabsf(d->limits_hi[k] - d->limits_lo[k]);
    e.ws_min[k] = d->ws_min[k];
    e.ws_max[k] = d->ws_max[k];
  }
  e.nx = d->grid_nx; e.ny = d->grid_ny; e.n_grids = d->n_grids;
  e.grids = reinterpret_cast<const float4*>(d->sdf_grids_dev);
  e.robot_map = d->robot_map_dev;
  e.xs = reinterpret_cast<const float4*>(d->extra_spheres_dev); e.n_xs = d->extra_spheres_dev ? d->n_extra_spheres : 0;
  e.xb = reinterpret_cast<const float4*>(d->extra_boxes_dev); e.n_xb = d->extra_boxes_dev ? d->n_extra_boxes : 0;
  return 0;
}

// occupancy of one point: any fixed-object SDF (nearest cell) or any of the 4 workspace-boundary distances below `margin`
__device__ __forceinline__ bool point_collides(const EnvDev& e, const float4* __restrict__ grid, float px, float py,
                                               float margin) {
  bool c = false;
  if (e.n_grids > 0) {
    int ix = (int)floorf((px - e.lo[0]) / e.dim[0] * (float)e.nx);
    int iy = (int)floorf((py - e.lo[1]) / e.dim[1] * (float)e.ny);
    ix = min(max(ix, 0), e.nx - 1);
    iy = min(max(iy, 0), e.ny - 1);
    for (int k = 0; k < e.n_grids; ++k) c = c || grid[((size_t)k * e.nx + ix) * e.ny + iy].x < margin;
  }
  if (e.n_xs + e.n_xb > 0) {                                // env.get_df_obj_list(): the fixed grid + obj_extra_list
    float gx, gy;
    c = c || extra_sdf(e.xs, e.n_xs, e.xb, e.n_xb, px, py, gx, gy) < margin;
  }
  c = c || (px - e.ws_min[0] < margin) || (py - e.ws_min[1] < margin) || (e.ws_max[0] - px < margin) ||
      (e.ws_max[1] - py < margin);
  return c;
}

__device__ __forceinline__ float lane_next_f(float v) {   // lane t reads lane t + 1 (wave_shl:1)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

struct PostArgs {
  EnvDev env;
  const float4* trajs;          // [n_traj][K * H] un-normalised
  int n_traj, samples_per_robot, n_interp, all_free;
  int K;                        // blocks of H = 64 support points per trajectory (MPDEnsemble: K tiles along the horizon)
  float alpha[MAX_INTERP], one_minus_alpha[MAX_INTERP];
  float margin, q_min[2], q_max[2];
  const float* savgol;          // [K*H][K*H] row-major operator or null
  int savgol_band;              // non-zeros of row r lie in columns [r - band, r + band]
  unsigned char* waypoint_coll; // [n_traj][(K*H-1) * n_interp] or null
  unsigned char* free_mask;     // [n_traj]
  float* path_length;           // [n_traj]
  float* smoothness;            // [n_traj]
  float4* smoothed;             // [n_traj][K*H] or null
};

constexpr int MAX_BLOCKS = 16;

// one workgroup per trajectory: wave k = support points [64 k, 64 k + 64), lane = point
__global__ __launch_bounds__(MAX_BLOCKS * 64) void postprocess_kernel(PostArgs a) {
  __shared__ float red[MAX_BLOCKS][2];
  __shared__ int bad_any[MAX_BLOCKS];
  const int t = threadIdx.x & 63, k = threadIdx.x >> 6;
  const int traj = blockIdx.x;
  const int L = a.K * H, p = k * H + t;                 // trajectory length, this lane's support point
  const float4* tr = a.trajs + (size_t)traj * L;
  const float4 x = tr[p];
  float nx = lane_next_f(x.x), ny = lane_next_f(x.y), nz = lane_next_f(x.z), nw = lane_next_f(x.w);
  if (t == H - 1 && p + 1 < L) {                        // the segment that crosses into the next block
    const float4 n = tr[p + 1];
    nx = n.x; ny = n.y; nz = n.z; nw = n.w;
  }
  const int robot = traj / a.samples_per_robot;
  const int map = a.env.robot_map ? a.env.robot_map[robot] : 0;
  const float4* grid = a.env.grids + (size_t)map * a.env.n_grids * a.env.nx * a.env.ny;

  // interpolated points of segment p -> p+1: x_p * alpha + x_{p+1} * (1 - alpha)   (trajectory/utils.py:80-81)
  bool coll = false;
  if (p < L - 1 && !a.all_free) {
    for (int j = 0; j < a.n_interp; ++j) {
      const float px = x.x * a.alpha[j] + nx * a.one_minus_alpha[j];
      const float py = x.y * a.alpha[j] + ny * a.one_minus_alpha[j];
      const bool c = point_collides(a.env, grid, px, py, a.margin);
      coll = coll || c;
      if (a.waypoint_coll) a.waypoint_coll[((size_t)traj * (L - 1) + p) * a.n_interp + j] = c ? 1 : 0;
    }
  } else if (p < L - 1 && a.waypoint_coll) {
    for (int j = 0; j < a.n_interp; ++j) a.waypoint_coll[((size_t)traj * (L - 1) + p) * a.n_interp + j] = 0;
  }
  // joint limits of the support points (tasks.py:270-276)
  const bool inside = x.x >= a.q_min[0] && x.x <= a.q_max[0] && x.y >= a.q_min[1] && x.y <= a.q_max[1];
  const bool bad = a.all_free ? false : (coll || !inside);
  const unsigned long long wave_bad = __ballot(bad);
  // path length / smoothness: sum_p || diff ||  (metrics.py:13-14, :36-38)
  float pl = 0.f, sm = 0.f;
  if (p < L - 1) {
    const float dx = nx - x.x, dy = ny - x.y, dvx = nz - x.z, dvy = nw - x.w;
    pl = sqrtf(dx * dx + dy * dy);
    sm = sqrtf(dvx * dvx + dvy * dvy);
  }
  pl = wave_sum(pl);
  sm = wave_sum(sm);
  if (t == 0) { red[k][0] = pl; red[k][1] = sm; bad_any[k] = wave_bad ? 1 : 0; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float spl = 0.f, ssm = 0.f;
    int any = 0;
    for (int j = 0; j < a.K; ++j) { spl += red[j][0]; ssm += red[j][1]; any |= bad_any[j]; }
    a.free_mask[traj] = any ? 0 : 1;
    a.path_length[traj] = spl;
    a.smoothness[traj] = ssm;
  }
  if (a.smoothed) {
    float4 o = x;
    if (a.savgol) {
      o = make_float4(0.f, 0.f, 0.f, 0.f);
      const int j0 = max(p - a.savgol_band, 0), j1 = min(p + a.savgol_band, L - 1);
      const float* srow = a.savgol + (size_t)p * L;
      for (int j = j0; j <= j1; ++j) {
        const float s = srow[j];
        const float4 v = tr[j];
        o.x = fmaf(s, v.x, o.x); o.y = fmaf(s, v.y, o.y); o.z = fmaf(s, v.z, o.z); o.w = fmaf(s, v.w, o.w);
      }
    }
    a.smoothed[(size_t)traj * L + p] = o;
  }
}

// per robot: index of the best sample among the free ones.  counts == null: argmin of cost_a (+ cost_b) (torch.argmin:
// first minimum); counts != null: first free sample with the fewest collisions (strict '<' scan, cbs.py:452).  When no
// sample is free, idx = the same criterion over ALL samples and n_free = 0 (the caller decides what to do with it).
__global__ __launch_bounds__(64) void select_best_kernel(const unsigned char* __restrict__ free_mask,
                                                         const float* __restrict__ cost_a, const float* __restrict__ cost_b,
                                                         const int* __restrict__ counts, int B, int* __restrict__ idx_best,
                                                         int* __restrict__ n_free, float* __restrict__ summary) {
  const int r = blockIdx.x, lane = threadIdx.x;
  int nf = 0;
  for (int b = lane; b < B; b += 64) nf += free_mask[(size_t)r * B + b] ? 1 : 0;
  if (summary)                                                // the host's one transfer: free flags as floats, then the picks
    for (int b = lane; b < B; b += 64) summary[(size_t)r * B + b] = free_mask[(size_t)r * B + b] ? 1.f : 0.f;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) nf += __shfl_xor(nf, m);
  float best = INFINITY;
  int best_i = 0x7fffffff;
  for (int b = lane; b < B; b += 64) {
    const size_t i = (size_t)r * B + b;
    if (nf > 0 && !free_mask[i]) continue;
    const float key = counts ? (float)counts[i] : (cost_b ? cost_a[i] + cost_b[i] : cost_a[i]);
    if (key < best || (key == best && b < best_i) || best_i == 0x7fffffff) { best = key; best_i = b; }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float ob = __shfl_xor(best, m);
    const int oi = __shfl_xor(best_i, m);
    if (oi != 0x7fffffff && (best_i == 0x7fffffff || ob < best || (ob == best && oi < best_i))) { best = ob; best_i = oi; }
  }
  if (lane == 0) {
    idx_best[r] = best_i == 0x7fffffff ? -1 : best_i;
    n_free[r] = nf;
    if (summary) summary[(size_t)gridDim.x * B + r] = (float)(best_i == 0x7fffffff ? -1 : best_i);
  }
}

__global__ void points_collision_kernel(EnvDev e, const float* __restrict__ pts, int n, int stride, int map, float margin,
                                        unsigned char* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4* grid = e.grids + (size_t)map * e.n_grids * e.nx * e.ny;
  out[i] = point_collides(e, grid, pts[(size_t)i * stride], pts[(size_t)i * stride + 1], margin) ? 1 : 0;
}

// compute_variance_waypoints (metrics.py:17-27): for waypoint t the unbiased variance of ALL B*B entries of
// triu(cdist(p_t, p_t), diagonal=1) (the zeros of the lower triangle and the diagonal included); one workgroup per t.
__global__ __launch_bounds__(256) void variance_waypoints_kernel(const float4* __restrict__ trajs, int B, int L,
                                                                 float* __restrict__ var_t) {
  __shared__ double s1[256], s2[256];
  const int t = blockIdx.x;
  double a = 0.0, b = 0.0;
  const long long pairs = (long long)B * B;
  for (long long p = threadIdx.x; p < pairs; p += 256) {
    const int i = (int)(p / B), j = (int)(p % B);
    if (j <= i) continue;
    const float4 u = trajs[(size_t)i * L + t], v = trajs[(size_t)j * L + t];
    const float dx = u.x - v.x, dy = u.y - v.y;
    const double d = (double)sqrtf(dx * dx + dy * dy);
    a += d; b += d * d;
  }
  s1[threadIdx.x] = a; s2[threadIdx.x] = b;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) { s1[threadIdx.x] += s1[threadIdx.x + m]; s2[threadIdx.x] += s2[threadIdx.x + m]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double n = (double)pairs;
    var_t[t] = n > 1.0 ? (float)((s2[0] - s1[0] * s1[0] / n) / (n - 1.0)) : NAN;
  }
}

// LimitsNormalizer.unnormalize (mmd/datasets/normalization.py:157-168) of a whole chain on the device, with the reference's
// data-dependent clip -- the WHOLE tensor is clipped to [-1, 1] iff ANY element lies outside [-1 - eps, 1 + eps] -- decided by a
// reduction kernel into a device flag (no host round trip: the torch form `if x.max() > 1 + eps or x.min() < -1 - eps` costs two
// reductions and two synchronisations per planner call), then x_u = (x + 1) / 2 * (max - min) + min with torch's separate
// roundings (no FMA contraction: this file is compiled with fp contract off).
__global__ __launch_bounds__(256) void range_flag_kernel(const float4* __restrict__ x, size_t n, size_t period, size_t segment, float eps,
                                                         uint32_t* __restrict__ flags) {
  const float hi = 1.f + eps, lo = -1.f - eps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    const bool out = v.x > hi || v.y > hi || v.z > hi || v.w > hi || v.x < lo || v.y < lo || v.z < lo || v.w < lo;
    // a relaxed store (a plain global_store) of the one value a flag ever takes (a wave's lanes with the same flag coalesce into one write; an atomic per element
    // serialises on a DDPM chain, whose x_T rows are a third out of range: +0.35 ms per planner call, measured)
    if (out) __hip_atomic_store(flags + (i % period) / segment, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
struct UnnormArgs { float mins[4], range[4]; };
__global__ __launch_bounds__(256) void unnormalize_kernel(const float4* __restrict__ x, float4* __restrict__ out, size_t n, size_t period,
                                                          size_t segment, UnnormArgs a, const uint32_t* __restrict__ flags) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool clip = flags[(i % period) / segment] != 0u;
  auto f = [&](float v, int d) {
    if (clip) v = fminf(fmaxf(v, -1.f), 1.f);                // torch.clip(x, -1, 1)
    v = (v + 1.f) / 2.f;
    return v * a.range[d] + a.mins[d];
  };
  const float4 v = x[i];
  out[i] = make_float4(f(v.x, 0), f(v.y, 1), f(v.z, 2), f(v.w, 3));
}

}  // namespace mmd

using namespace mmd;

extern "C" {

int mmd_postprocess_trajs(const mmd_guide_desc* env, const float* trajs_dev, int n_robots, int samples_per_robot,
                          int horizon, int num_interpolation, const float* alpha, float margin, const float* q_min,
                          const float* q_max, int all_free, const float* savgol_dev, int savgol_band,
                          uint8_t* waypoint_collisions_dev,
                          uint8_t* free_dev, float* path_length_dev, float* smoothness_dev, float* smoothed_dev,
                          void* stream) {
  MMD_REQUIRE(env && trajs_dev && free_dev && path_length_dev && smoothness_dev && q_min && q_max,
              "mmd_postprocess_trajs: NULL argument");
  MMD_REQUIRE(horizon >= H && horizon % H == 0 && horizon / H <= MAX_BLOCKS, "horizon must be a multiple of %d (at most %d)",
              H, MAX_BLOCKS * H);
  MMD_REQUIRE(savgol_band >= 0, "mmd_postprocess_trajs: negative savgol band");
  MMD_REQUIRE(num_interpolation >= 0 && num_interpolation <= MAX_INTERP && (num_interpolation == 0 || alpha),
              "mmd_postprocess_trajs: 0 <= num_interpolation <= %d with its alpha table", MAX_INTERP);
  MMD_REQUIRE(n_robots >= 1 && samples_per_robot >= 1, "mmd_postprocess_trajs: empty batch");
  PostArgs a{};
  if (int rc = fill_env(env, a.env)) return rc;
  a.trajs = reinterpret_cast<const float4*>(trajs_dev);
  a.n_traj = n_robots * samples_per_robot;
  a.samples_per_robot = samples_per_robot;
  a.n_interp = num_interpolation;
  a.all_free = all_free;
  for (int j = 0; j < num_interpolation; ++j) {
    a.alpha[j] = alpha[j];
    a.one_minus_alpha[j] = 1.f - alpha[j];        // torch: (1 - alpha) in fp32
  }
  a.margin = margin;
  for (int k = 0; k < 2; ++k) { a.q_min[k] = q_min[k]; a.q_max[k] = q_max[k]; }
  a.savgol = savgol_dev;
  a.savgol_band = savgol_band;
  a.K = horizon / H;
  a.waypoint_coll = waypoint_collisions_dev;
  a.free_mask = free_dev;
  a.path_length = path_length_dev;
  a.smoothness = smoothness_dev;
  a.smoothed = reinterpret_cast<float4*>(smoothed_dev);
  hipLaunchKernelGGL(postprocess_kernel, dim3(a.n_traj), dim3(a.K * 64), 0, (hipStream_t)stream, a);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_select_best(const uint8_t* free_dev, const float* cost_a_dev, const float* cost_b_dev, const int32_t* counts_dev,
                    int n_robots, int samples_per_robot, int32_t* idx_best_dev, int32_t* n_free_dev, float* summary_dev,
                    void* stream) {
  MMD_REQUIRE(free_dev && (cost_a_dev || counts_dev) && idx_best_dev && n_free_dev, "mmd_select_best: NULL argument");
  MMD_REQUIRE(n_robots >= 1 && samples_per_robot >= 1, "mmd_select_best: empty batch");
  hipLaunchKernelGGL(select_best_kernel, dim3(n_robots), dim3(64), 0, (hipStream_t)stream, free_dev, cost_a_dev, cost_b_dev,
                     counts_dev, samples_per_robot, idx_best_dev, n_free_dev, summary_dev);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_points_collision(const mmd_guide_desc* env, const float* points_dev, int n_points, int point_stride, int map_index,
                         float margin, uint8_t* out_dev, void* stream) {
  MMD_REQUIRE(env && points_dev && out_dev && n_points >= 1 && point_stride >= 2, "mmd_points_collision: bad arguments");
  EnvDev e{};
  if (int rc = fill_env(env, e)) return rc;
  MMD_REQUIRE(map_index >= 0 && map_index < (env->n_maps > 0 ? env->n_maps : 1), "mmd_points_collision: map index");
  hipLaunchKernelGGL(points_collision_kernel, dim3((n_points + 255) / 256), dim3(256), 0, (hipStream_t)stream, e, points_dev,
                     n_points, point_stride, map_index, margin, out_dev);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_unnormalize_trajs(const float* x_dev, size_t n_points, size_t period_points, size_t segment_points, const float* mins,
                          const float* maxs, float eps, float* out_dev, uint32_t* flags_dev, void* stream) {
  MMD_REQUIRE(x_dev && out_dev && mins && maxs && flags_dev, "mmd_unnormalize_trajs: NULL argument");
  if (n_points == 0) return 0;
  if (period_points == 0) period_points = n_points;
  if (segment_points == 0) segment_points = period_points;
  MMD_REQUIRE(period_points % segment_points == 0 && n_points % period_points == 0, "mmd_unnormalize_trajs: n_points %% period %% segment");
  hipStream_t st = (hipStream_t)stream;
  MMD_HIP_CHECK(hipMemsetAsync(flags_dev, 0, sizeof(uint32_t) * (period_points / segment_points), st));
  const unsigned blocks = (unsigned)((n_points + 1023) / 1024);
  hipLaunchKernelGGL(range_flag_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, st, (const float4*)x_dev, n_points, period_points,
                     segment_points, eps, flags_dev);
  UnnormArgs a;
  for (int d = 0; d < 4; ++d) { a.mins[d] = mins[d]; a.range[d] = maxs[d] - mins[d]; }
  hipLaunchKernelGGL(unnormalize_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, st, (const float4*)x_dev,
                     (float4*)out_dev, n_points, period_points, segment_points, a, (const uint32_t*)flags_dev);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_variance_waypoints(const float* trajs_dev, int n_traj, int horizon, float* var_per_waypoint_dev, void* stream) {
  MMD_REQUIRE(trajs_dev && var_per_waypoint_dev && n_traj >= 1 && horizon >= 1, "mmd_variance_waypoints: bad arguments");
  hipLaunchKernelGGL(variance_waypoints_kernel, dim3(horizon), dim3(256), 0, (hipStream_t)stream, (const float4*)trajs_dev,
                     n_traj, horizon, var_per_waypoint_dev);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
