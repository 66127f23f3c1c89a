/*
 * mmd_amd.h -- C ABI of libmmd_amd.so: the MI355X (gfx950) guided-diffusion trajectory sampler that drops in
 * behind yoraish/mmd's MPD / MPDEnsemble planners.
 *
 * The reference has no FFI layer (it is pure Python); the boundary a maintainer would bind is the
 * "inner" call contract of the planner.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference checkout).  All pointers suffixed _dev are device (HIP) pointers to contiguous
 * fp32 / int32 data; everything else is host memory.  `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  Every function returns 0 on success and a non-zero code on failure; mmd_last_error() gives the text.
 * No global state besides the last-error string (thread-local) and lazily created per-(thread, device) side streams;
 * handles are immutable after creation, so entry points are re-entrant per (handle, stream).  The library never reads the
 * environment: every switch, measurement ones included, is a field of a descriptor passed with the call (mmd_unet_options,
 * mmd_sampler_desc.flags / .n_streams / .guide_coop_max).
 *
 * Trajectory tensors are [n_traj, H, D] fp32 with D = 4 (x, y, vx, vy) and H = 64 support points, in the
 * NORMALISED space of the diffusion model; n_traj = n_robots * samples_per_robot, robot-major.
 */
#ifndef MMD_AMD_H
#define MMD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMD_AMD_ABI_VERSION 7
#define MMD_STATE_DIM 4
#define MMD_HORIZON 64

typedef struct mmd_unet_s* mmd_unet_t; /* opaque: packed TemporalUnet weights + time-embedding table on device */

int mmd_abi_version(void);
const char* mmd_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * TemporalUnet  (replaces mmd/models/diffusion_models/temporal_unet.py:23-174 `TemporalUnet.__init__/forward`,
 * layers mmd/models/layers/layers.py:232-358)
 * ---------------------------------------------------------------------------------------------------------- */

/* Number of parameter tensors, and the element count of tensor i, in the reference's state_dict order for
 * TemporalUnet(state_dim=4, n_support_points=64, unet_input_dim, dim_mults=(1, 2, 4, 8)[:n_levels]).
 * unet_input_dim: a multiple of 8 in [8, 64]; n_levels 1 .. 4 (UNET_DIM_MULTS, mmd/models/__init__.py:8-11: option 0 =
 * 3 levels, option 1 = 4 levels); anything else returns -1.  unet_input_dim == 32 with 3 levels (the released checkpoints)
 * runs the fused one-launch kernel, every other shape the layer-by-layer kernels (csrc/unet_layers.hip). */
int mmd_unet_num_tensors(int unet_input_dim, int n_levels);
int64_t mmd_unet_tensor_numel(int unet_input_dim, int n_levels, int index);

/* Build the device-side model from HOST fp32 tensors given in state_dict order (the values of
 * `{k: v for k, v in diffusion_model.state_dict().items() if k.startswith('model.')}`; replaces
 * `diffusion_model.load_state_dict(...)`, mmd/planners/single_agent/mpd.py:167-172).  `numels[i]` is checked
 * against mmd_unet_tensor_numel.  Also precomputes, on the GPU, the time-embedding projections of every integer
 * diffusion step t in [0, n_diffusion_steps) (TimeEncoder + the 12 cond_mlp heads; t is identical across the
 * batch, mmd/models/diffusion_models/diffusion_model_base.py:27-29). */
typedef struct mmd_unet_options {     /* creation-time choices, fixed for the life of the handle; NULL / all zero = defaults */
  uint32_t flags;                     /* MMD_UNET_* below */
  int32_t rtb_fused;                  /* layer-by-layer path, A/B: the widest ResidualTemporalBlock run as ONE launch (channels);
                                       * 0 = none, < 0 = default (64) */
  int32_t mconv_max_cs;               /* layer-by-layer path, A/B: the widest column slice of its matrix-pipe kernel; 0 = default (128) */
  int32_t two_per_workgroup_max;      /* fused kernel, A/B: batches up to this size run two trajectories per workgroup; 0 = default
                                       * (512), < 0 = never */
} mmd_unet_options;
#define MMD_UNET_LAYERED 1u           /* the layer-by-layer kernels for the fused kernel's own configuration too (the two
                                       * implementations share no device code: tests hold one against the other) */
#define MMD_UNET_LAYERED_VALU 2u      /* layer-by-layer path: every layer on the vector-ALU kernels (A/B of its matrix-pipe kernel) */

int mmd_unet_create(mmd_unet_t* out, int unet_input_dim, int n_levels, int n_diffusion_steps,
                    const float* const* tensors, const int64_t* numels, int n_tensors, const mmd_unet_options* options,
                    void* stream);
int mmd_unet_destroy(mmd_unet_t unet);

/* Scratch needed by mmd_unet_forward for n_traj trajectories; allocate it with the host framework.  (The fused kernel keeps
 * every activation on chip: 12 KiB, the step table + argument block of a persistent run of unguided steps in mmd_p_sample_loop;
 * the layer-by-layer path keeps (5 + n_levels) tensors of n_traj x 64 x unet_input_dim floats there.) */
size_t mmd_unet_workspace_bytes(mmd_unet_t unet, int n_traj);

/* eps = model(x, t, context=None)  (temporal_unet.py:121; called from p_mean_variance,
 * diffusion_model_base.py:152).  t is one integer for the whole batch. */
int mmd_unet_forward(mmd_unet_t unet, const float* x_dev, int t, float* eps_dev, int n_traj, void* workspace_dev,
                     size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Guide  (replaces GuideManagerTrajectoriesWithVelocity.forward, mmd/models/diffusion_models/guides.py:180-226,
 * with the cost terms of deps/motion_planning_baselines/mp_baselines/planners/costs/cost_functions.py:149-193,
 * :275-326, :505-542 and the SDF fields of deps/torch_robotics/.../distance_fields.py:333-367,
 * environments/grid_map_sdf.py:84-114)
 * ---------------------------------------------------------------------------------------------------------- */

typedef struct mmd_guide_desc {
  /* LimitsNormalizer (mmd/datasets/normalization.py:145-168): x_u = (clip(x,-1,1)+1)/2*(max-min)+min.
   * The clip is applied unconditionally (the reference clips only if the batch leaves [-1-1e-4, 1+1e-4]). */
  float norm_min[MMD_STATE_DIM];
  float norm_max[MMD_STATE_DIM];
  /* SDF grids of the fixed objects: GridMapSDF (grid_map_sdf.py:9-114).  Cell (ix,iy) holds float4
   * (sdf, d sdf/dx, d sdf/dy, 0); layout [n_maps][n_grids][nx][ny][4]; index = floor((p-lo)/(hi-lo)*n) clamped.
   * n_grids == 0 declares an obstacle-free map (EnvEmpty2D: sdf == 1, primitives.py:109-110): no gather is issued. */
  float limits_lo[2];
  float limits_hi[2];
  int32_t grid_nx, grid_ny, n_grids, n_maps;
  const float* sdf_grids_dev;
  const int32_t* robot_map_dev;      /* [n_robots] map index per robot, or NULL (all robots use map 0) */
  /* workspace boundaries (tasks.py:75-86, already scaled by 1.08) */
  float ws_min[2];
  float ws_max[2];
  float margin;                      /* 1.1 * robot radius + obstacle cutoff margin (distance_fields.py:117) */
  float dt;                          /* trajectory_duration / n_support_points (mpd.py:140) */
  float sigma_gp;                    /* 1.0 (mpd.py:237) */
  float weight_collision;            /* weight_grad_cost_collision  (mmd_params.py:40) */
  float weight_smoothness;           /* weight_grad_cost_smoothness (mmd_params.py:41) */
  float max_grad_norm;               /* 1.0 (guides.py:154) */
  /* Constraints: one group per CostConstraint (= per MultiPointConstraint, mmd/common/constraints.py:46-85),
   * stored time-bucketed (ELL): slot j of a group holds, for every time step t, at most one active point
   * float4 (qx, qy, radius, radius * |radius|) with radius < 0 meaning "no point".  cons_ell_dev is [n_slots][H][4];
   * group g owns slots [grp_slot_off[g], grp_slot_off[g+1]); robot r owns groups
   * [robot_grp_off[r], robot_grp_off[r+1]).  Build it with mmd_pack_constraints or
   * mmd_soft_constraints_from_paths.  NULL pointers = no constraints. */
  const float* cons_ell_dev;
  const int32_t* grp_slot_off_dev;
  const float* grp_weight_dev;
  const int32_t* robot_grp_off_dev;
  int32_t max_slots_per_robot;       /* max over robots of their total slot count (sizes the LDS staging; 0 = unknown) */
  float cons_uniform_radius;         /* > 0: every active point of the table has exactly this radius (tables made by
                                      * mmd_soft_constraints_from_paths): the kernel then keeps only (qx, qy) on chip,
                                      * twice the slots per workgroup.  0 = radii vary, general path. */
  /* Extra objects of the environment (EnvBase.obj_extra_list, env_base.py:76-89: an ObjectField of primitive fields at the
   * identity pose), evaluated ANALYTICALLY as the reference does -- one more signed-distance field next to the grids of the
   * fixed objects (df_obj_l = [grid, *obj_extra_list]; cost = max over the fields, distance_fields.py:110-126): spheres
   * |p - c| - r (MultiSphereField, primitives.py:108-115), boxes as the rounded box of the fixed objects (MultiBoxField --
   * an ALIAS of MultiRoundedBoxField, primitives.py:345 -- i.e. primitives.py:326-333: corner radius 0.15 x the smaller size), minimum over all of them.  n = 0 / NULL: the env has none (every shipped map: an empty sphere list, sdf = 1). */
  const float* extra_spheres_dev;    /* [n_extra_spheres][4]: (cx, cy, r, 0) */
  const float* extra_boxes_dev;      /* [n_extra_boxes][4]: (cx, cy, half size x, half size y) */
  int32_t n_extra_spheres, n_extra_boxes;
  /* GuideManager.clip_gradient (guides.py:228-259), applied to every cost's per-point gradient: 0 = clip_grad_by_norm with
   * max_grad_norm (what MPD / MPDEnsemble set, mpd.py:258-265), 1 = clip_grad_by_value: torch.clip(grad, -max_grad_value,
   * max_grad_value) (the class default 0.1), 2 = clip_grad = False */
  int32_t clip_grad_rule;
  float max_grad_value;
} mmd_guide_desc;

/* Host helper: time-bucket one robot's constraint groups.  For group g (n_pts[g] points): q [n,2], t_range [n,2]
 * as [t0, t1) (exclusive end, cost_functions.py:305), radius [n].  Writes the ELL block into `ell_out`
 * ([max_slots][H][4], host) and returns the number of slots used by each group in slots_out[g]; returns an error
 * if max_slots is too small.  With ell_out == NULL only slots_out is filled (sizing pass).  (Replaces the per-call CostConstraint construction, mpd.py:329-342.) */
int mmd_pack_constraints(int n_groups, const int32_t* n_pts, const float* const* q, const float* const* t_range,
                         const float* const* radius, int horizon, float* ell_out, int max_slots,
                         int32_t* slots_out);

/* Device helper: all-pairs soft constraints from the robots' current best paths (replaces
 * CBS.create_soft_constraints_from_other_agents_paths, mmd/planners/multi_agent/cbs.py:468-508, for equal start
 * times).  paths_dev [n_all, H, 2] un-normalised positions of ALL robots (after the all-gather); this rank owns
 * robots [robot0, robot0 + n_local).  Writes one group of (n_all-1) slots per local robot into ell_out_dev
 * ([n_local*(n_all-1)][H][4]) plus the three offset/weight arrays (sizes n_local+1, n_local, n_local+1). */
int mmd_soft_constraints_from_paths(const float* paths_dev, int n_all, int robot0, int n_local, int horizon,
                                    float radius, float weight, float* ell_out_dev, int32_t* grp_slot_off_dev,
                                    float* grp_weight_dev, int32_t* robot_grp_off_dev, void* stream);

/* n_steps x { x += guide(x); apply_hard_conditioning }  (guide_gradient_steps,
 * mmd/models/diffusion_models/sample_functions.py:89-107).  Hard conditions (apply_hard_conditioning,
 * sample_functions.py:8-14: the dict {support point: state}): bit t of hard_rows set = support point t of every trajectory
 * is pinned; hard_dev [n_robots][popcount(hard_rows)][4] holds each robot's pinned states in ascending row order
 * (MPD's {0: start, H-1: goal} = hard_rows 0x8000000000000001, hard_dev [n_robots][2][4]).  chain_dev, if not NULL,
 * receives the state after EVERY iteration, [n_steps][n_traj, H, 4] (the post-diffusion guide steps of planner_alg
 * 'diffusion_prior_then_guide', mmd/planners/single_agent/mpd.py:429-453, in one launch). */
int mmd_guide_steps(const mmd_guide_desc* g, float* x_dev, const float* hard_dev, uint64_t hard_rows, int n_robots,
                    int samples_per_robot, int n_steps, float* chain_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * DDPM sampling  (replaces ddpm_sample_fn, sample_functions.py:40-86, and GaussianDiffusionModel.p_sample_loop /
 * run_inference / run_local_inference, diffusion_model_base.py:162-211, :320-421)
 * ---------------------------------------------------------------------------------------------------------- */

typedef struct mmd_sampler_desc {
  int32_t n_diffusion_steps;                /* T of the schedule tables */
  /* [T] host tables (GaussianDiffusionModel buffers, diffusion_model_base.py:83-105) */
  const float* sqrt_recip_alphas_cumprod;
  const float* sqrt_recipm1_alphas_cumprod;
  const float* posterior_mean_coef1;
  const float* posterior_mean_coef2;
  const float* posterior_log_variance_clipped;
  int32_t n_guide_steps;                    /* 20 (mmd_params.py:38) */
  int32_t t_start_guide;                    /* guide iff loop index i < t_start_guide (sample_functions.py:63) */
  float noise_std_extra;                    /* 0.5 (mpd.py:303) */
  uint64_t hard_rows;                       /* as in mmd_guide_steps */
  int32_t n_streams;                        /* mmd_p_sample_loop splits the robots into this many concurrent HIP
                                             * streams (forked from / joined to `stream`) so one chunk's staging and
                                             * epilogues overlap the other's MFMA phases; 0 = auto (2 above 512
                                             * trajectories, else 1), 1 = off */
  int64_t traj_index_base;                  /* GLOBAL index of this call's trajectory 0 (= first global robot * samples per
                                             * robot).  The in-kernel Philox4x32-10 draws are keyed by (seed, draw, global
                                             * trajectory * H + t), so a rank that samples robots [r0, r1) of an N-robot
                                             * instance draws exactly the noise those rows get in the unsharded call
                                             * (per-robot outputs bitwise identical for G = 1, 2, 4, 8) */
  const float* noise_std_extra_by_t;        /* optional [T] host table: noise_std_extra_schedule_fn(t) evaluated for every
                                             * t (sample_functions.py:83-86 calls it per step); NULL = the constant above */
  void* profiler;                           /* optional mmd_profiler_t (include/mmd_amd_debug.h) that brackets UNet launches
                                             * with HIP events; NULL in production */
  int32_t scale_grad_by_std;                /* 1: every guide gradient is multiplied by model_var = exp(posterior_log_variance_
                                             * clipped[t]) before it is added (guide_gradient_steps, sample_functions.py:100-101) */
  int32_t model_predicts_x0;                /* 1: GaussianDiffusionModel(predict_epsilon=False): the network output IS x_recon
                                             * (predict_start_from_noise / predict_noise_from_start, diffusion_model_base.py:114-141) */
  uint32_t flags;                           /* MMD_SAMPLER_* below (measurement switches; 0 in production) */
  int32_t guide_coop_max;                   /* A/B: a guided step of up to this many trajectories per launch runs four waves per
                                             * trajectory; 0 = default (512), < 0 = never */
  const uint64_t* robot_seeds_dev;          /* optional DEVICE array [n_robots]: one Philox stream per ROBOT -- robot r's draws are
                                             * keyed by (robot_seeds[r], draw, index within the robot * H + t) and `seed` /
                                             * traj_index_base are ignored.  R independent planner calls (cbs.py:316-324,
                                             * inference_multi_agent.py:225-237: one MPD call per agent) batched into ONE launch
                                             * sequence then draw exactly the noise of the R separate calls with those seeds.
                                             * NULL = one stream per call */
} mmd_sampler_desc;
#define MMD_SAMPLER_NO_FUSED_STEP 1u        /* unguided steps as separate step-kernel launches instead of the UNet launch's tail */
#define MMD_SAMPLER_PERSIST 2u              /* the leading run of unguided steps of mmd_p_sample_loop as persistent launches (<= 64
                                             * steps each; the first 12 KiB of the workspace then hold the step table) */

/* Scratch needed by mmd_ddpm_step / mmd_p_sample_loop: [mmd_unet_workspace_bytes][eps: n_traj * H * 4 floats].  The chunked
 * (n_streams > 1) loop uses slices of the same eps block, so this size is exact for every n_streams. */
size_t mmd_sampler_workspace_bytes(mmd_unet_t unet, int n_traj);

/* One ddpm_sample_fn call + the apply_hard_conditioning that follows it in p_sample_loop
 * (diffusion_model_base.py:199-203): x <- step(x) for loop index i (i < 0 means t = 0, no noise).
 * guide may be NULL (no guidance).  noise_dev [n_traj,H,4] is the injected randn_like draw, or NULL to draw it
 * in-kernel from Philox4x32-10 keyed by (seed, draw_index). */
int mmd_ddpm_step(mmd_unet_t unet, const mmd_sampler_desc* s, const mmd_guide_desc* guide, float* x_dev,
                  const float* hard_dev, int n_robots, int samples_per_robot, int i, const float* noise_dev,
                  uint64_t seed, uint32_t draw_index, void* workspace_dev, size_t workspace_bytes, void* stream);

/* The whole loop: for i = n_steps-1 ... -n_steps_without_noise.  x_dev holds x_T (or the warm start; NULL noise
 * + init_noise != 0 draws x_T in-kernel) on entry and the final sample on exit.  chain_dev, if not NULL, receives
 * [n_steps + n_steps_without_noise + 1][n_traj,H,4] (chain[0] = conditioned x_T).  step_noise_dev, if not NULL, is
 * [n_steps + n_steps_without_noise][n_traj,H,4] injected draws in loop order. */
int mmd_p_sample_loop(mmd_unet_t unet, const mmd_sampler_desc* s, const mmd_guide_desc* guide, float* x_dev,
                      const float* hard_dev, int n_robots, int samples_per_robot, int n_steps,
                      int n_steps_without_noise, int init_noise, const float* step_noise_dev, uint64_t seed,
                      float* chain_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* GaussianDiffusionModel.ddim_sample (diffusion_model_base.py:213-290) with eta = 0: x_T (drawn with Philox when
 * init_noise != 0, else the contents of x_dev) -> conditioned -> one step per consecutive pair (times[k], times[k+1]) of
 * the host array `times` ([n_times], strictly decreasing, as the reference builds it: reversed int(linspace(0, T-1,
 * T/5 + 1)) followed by -1): eps = model(x, times[k]); x_start = predict_start_from_noise (not clamped);
 * x = x_start sqrt(acp[t_next]) + sqrt(1 - acp[t_next]) eps; s->n_guide_steps guide steps iff a guide is given and
 * t_next < s->t_start_guide (the reference effectively runs ONE: it does not forward n_guide_steps to
 * guide_gradient_steps); hard conditioning; on the pair that ends in -1, x = x_start.  alphas_cumprod is the host [T]
 * buffer of that name.  chain_dev: [n_times][n_traj, H, 4] or NULL. */
int mmd_ddim_sample(mmd_unet_t unet, const mmd_sampler_desc* s, const float* alphas_cumprod, const int32_t* times,
                    int n_times, const mmd_guide_desc* guide, float* x_dev, const float* hard_dev, int n_robots,
                    int samples_per_robot, int init_noise, uint64_t seed, float* chain_dev, void* workspace_dev,
                    size_t workspace_bytes, void* stream);

/* q_sample (diffusion_model_base.py:425-433): x = a * x_start + b * noise (noise injected or Philox keyed by
 * traj_index_base like the sampler).  n_traj counts blocks of H = 64 support points: a [B, K*64, 4] ensemble seed
 * (diffusion_ensemble.py:279-281) is n_traj = B * K. */
int mmd_q_sample(float* x_dev, const float* x_start_dev, const float* noise_dev, float sqrt_alphas_cumprod_t,
                 float sqrt_one_minus_alphas_cumprod_t, uint64_t seed, uint32_t draw_index, int64_t traj_index_base,
                 int n_traj, void* stream);

/* apply_cross_conditioning for one (m1, m2) tile pair (sample_functions.py:17-31): row ind1 of x1 := min(row ind2
 * of x2 + rel, boundary); then row ind2 of x2 := max(row ind1 of x1 - rel, -boundary); rel / boundary are [4] host. */
int mmd_cross_condition(float* x1_dev, float* x2_dev, int ind1, int ind2, const float* rel, const float* boundary,
                        int n_traj, void* stream);

/* DiffusionsEnsemble.p_sample_loop (mmd/models/diffusion_models/diffusion_ensemble.py:55-106): K tile models chained
 * along the horizon.  Per outer step the tiles step IN ORDER (UNet + fused DDPM/guide kernel of tile m on its own
 * x_dev), each followed by apply_cross_conditioning over all (m1, m2) pairs (sample_functions.py:17-31) -- the whole
 * loop is enqueued by this ONE call (no per-step host round trip).  Every tile has its own model handle, sampler
 * (schedule, guide-step counts, hard mask, Philox seed via `seed` or sampler->robot_seeds_dev), guide, state, chain and injected-noise buffers; all
 * tiles share n_robots / samples_per_robot and the workspace (sized by mmd_sampler_workspace_bytes of the largest). */
typedef struct mmd_ensemble_tile {
  mmd_unet_t unet;
  const mmd_sampler_desc* sampler;
  const mmd_guide_desc* guide;       /* or NULL */
  float* x_dev;                      /* [n_traj, H, 4]: x_T / warm start on entry (init_noise != 0: drawn), result on exit */
  const float* hard_dev;             /* [n_robots][2][4] */
  const float* step_noise_dev;       /* [n_steps + n_steps_without_noise][n_traj, H, 4] injected draws, or NULL */
  float* chain_dev;                  /* [n_steps + n_steps_without_noise + 1][n_traj, H, 4], or NULL.  Rows as the reference's
                                        chains hold them (it stores the tensor object and stitches in place,
                                        diffusion_ensemble.py:86-103): row k of tile m >= 1 also carries the boundary rows stitched
                                        after the earlier tiles' steps of outer step k + 1; the last row is the result */
  uint64_t seed;
} mmd_ensemble_tile;

typedef struct mmd_cross_cond {
  int32_t m1, m2, ind1, ind2;        /* row ind1 of tile m1 is stitched to row ind2 of tile m2 */
  float rel[4];                      /* transforms[m2] - transforms[m1], zero padded to the state dim */
  float boundary[4];                 /* rel / ||rel|| with zeros replaced by 1e6 */
  const float* by_robot_dev;         /* optional DEVICE table [n_robots][2][4] = (rel, boundary) per robot, replacing the two above:
                                        batched planner calls whose robots traverse different tile skeletons
                                        (inference_multi_agent.py:418-431: [[0,0],[0,1]] and [[0,1],[0,0]]); NULL = one pair */
} mmd_cross_cond;

int mmd_p_sample_loop_ensemble(const mmd_ensemble_tile* tiles, int n_tiles, const mmd_cross_cond* cross, int n_cross,
                               int n_robots, int samples_per_robot, int n_steps, int n_steps_without_noise,
                               int init_noise, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Multi-agent layer next to the sampler
 * ---------------------------------------------------------------------------------------------------------- */

/* RobotPlanarDisk.check_rr_collisions (deps/torch_robotics/torch_robotics/robots/robot_planar_disk.py:173-203) as
 * CBS.get_conflicts calls it (mmd/planners/multi_agent/cbs.py:185-190, equal start times, densification 1):
 * mask_dev [T][N][N] uint8 = (||p_i(t) - p_j(t)|| < margin) && i != j; midpoints_dev [T][N][N][2] = (p_i + p_j)/2 or
 * NaN where there is no collision (may be NULL).  paths_dev [N,T,2] un-normalised positions, T = `horizon` >= 1 (T = 1:
 * the start / goal validity check of mmd/common/multi_agent_utils.py:74-79); margin = 2.1 * radius. */
int mmd_rr_collisions(const float* paths_dev, int n_robots, int horizon, float margin, uint8_t* mask_dev,
                      float* midpoints_dev, void* stream);

/* The 'least_collisions' scan of CBS.expand (cbs.py:446-458) without the per-sample get_conflicts loop:
 * counts_dev[r*B + b] = #{(t, j != robot0 + r) : ||x_{r,b}(t) - p_j(t)|| < margin} for the local robots' sample
 * batches trajs_dev [n_local*B, H, 4] (un-normalised; only x, y are read) against ALL robots' best paths
 * paths_dev [n_all, H, 2].  (The reference's conflict count for sample b is a constant plus twice this number.) */
int mmd_count_collisions(const float* trajs_dev, const float* paths_dev, int robot0, int n_local,
                         int samples_per_robot, int n_all, int horizon, float margin, int32_t* counts_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Post-sampling selection: the step right after the sampler in MPD.__call__
 * (mmd/planners/single_agent/mpd.py:344-405)
 * ---------------------------------------------------------------------------------------------------------- */

/* PlanningTask.get_trajs_collision_and_free (deps/torch_robotics/torch_robotics/tasks/tasks.py:236-311) +
 * compute_path_length / compute_smoothness (trajectory/metrics.py:7-39) + smooth_trajs
 * (mmd/common/trajectory_utils.py:31-40) for a batch of UN-normalised trajectories trajs_dev [n_traj, horizon, 4]
 * (horizon = 64, or K * 64 for MPDEnsemble's K tiles chained along the horizon; H below = horizon):
 *   - every segment is linearly interpolated at `num_interpolation` points x_t * alpha_j + x_{t+1} * (1 - alpha_j)
 *     (alpha [num_interpolation] host = torch.linspace(0, 1, n + 2)[1:n+1], trajectory/utils.py:73-86) and each point
 *     is tested against the fixed-object SDF grids and workspace boundaries of `env` (only its map / boundary fields
 *     are read: limits_*, grid_*, n_grids, sdf_grids_dev, robot_map_dev, ws_*) with `margin` (= robot radius,
 *     tasks.py:251-253): waypoint_collisions_dev [n_traj][(H-1) * num_interpolation] (may be NULL);
 *   - free_dev[n] = 1 iff no interpolated point collides and every support point lies inside [q_min, q_max]
 *     (tasks.py:262-281); all_free != 0 skips both tests (PlanningTaskEnsemble, tasks_ensemble.py:271-277);
 *   - path_length_dev / smoothness_dev [n_traj]: sum_t ||p_{t+1} - p_t||, sum_t ||v_{t+1} - v_t||;
 *   - smoothed_dev [n_traj, H, 4] (may be NULL) = S @ trajectory with the Savitzky-Golay operator savgol_dev [H][H]
 *     (row-major, device; NULL = copy) whose row r is zero outside columns [r - savgol_band, r + savgol_band]. */
int mmd_postprocess_trajs(const mmd_guide_desc* env, const float* trajs_dev, int n_robots, int samples_per_robot,
                          int horizon, int num_interpolation, const float* alpha, float margin, const float* q_min,
                          const float* q_max, int all_free, const float* savgol_dev, int savgol_band,
                          uint8_t* waypoint_collisions_dev,
                          uint8_t* free_dev, float* path_length_dev, float* smoothness_dev, float* smoothed_dev,
                          void* stream);

/* Per robot, the index (within its samples_per_robot samples) of the best FREE sample: with counts_dev == NULL the
 * argmin of cost_a (+ cost_b if not NULL) (torch.argmin(cost_all), mpd.py:366-370); with counts_dev the first free
 * sample with the fewest robot-robot collisions (CBS 'least_collisions', cbs.py:446-458).  n_free_dev[r] = number of
 * free samples; when it is 0 the pick is made over all samples instead.  summary_dev (optional, fp32 [n_traj + n_robots]): the free
 * flags as 0 / 1 followed by the picks -- laid out so that, with path_length_dev / smoothness_dev of mmd_postprocess_trajs placed right
 * behind it, everything the host needs to assemble a PlannerOutput crosses in ONE device -> host copy. */
int mmd_select_best(const uint8_t* free_dev, const float* cost_a_dev, const float* cost_b_dev, const int32_t* counts_dev,
                    int n_robots, int samples_per_robot, int32_t* idx_best_dev, int32_t* n_free_dev, float* summary_dev,
                    void* stream);

/* PlanningTask.compute_collision (tasks.py:141-143, :204-232; occupancy of the fixed objects + workspace boundaries)
 * for n_points positions (x, y at points_dev[i * point_stride + {0, 1}]) on map `map_index` of `env`. */
int mmd_points_collision(const mmd_guide_desc* env, const float* points_dev, int n_points, int point_stride, int map_index,
                         float margin, uint8_t* out_dev, void* stream);

/* LimitsNormalizer.unnormalize (mmd/datasets/normalization.py:157-168; TrajectoryDataset.unnormalize_trajectories, what MPD.__call__
 * applies to the sampled chain, mpd.py:344-347) for n_points float4 states (x, y, vx, vy): a tensor is clipped to [-1, 1] as a WHOLE iff any of
 * its elements lies outside [-1 - eps, 1 + eps] (the reference's data-dependent clip, decided on the device), then x_u = (x + 1) / 2 *
 * (maxs - mins) + mins.  One call may hold SEVERAL tensors interleaved (the chains of R planner calls batched robot-major, [steps][R][B*H]):
 * point i belongs to tensor (i % period_points) / segment_points, each tensor gets its own clip decision; period_points = segment_points = 0
 * means one tensor.  flags_dev: period_points / segment_points uint32 of scratch.  mins / maxs: host [4].  out_dev may alias x_dev. */
int mmd_unnormalize_trajs(const float* x_dev, size_t n_points, size_t period_points, size_t segment_points, const float* mins,
                          const float* maxs, float eps, float* out_dev, uint32_t* flags_dev, void* stream);

/* compute_variance_waypoints (trajectory/metrics.py:17-27): var_per_waypoint_dev[t] = unbiased variance of all
 * n_traj^2 entries of triu(cdist(p_t, p_t), 1); the metric is their sum over t. */
int mmd_variance_waypoints(const float* trajs_dev, int n_traj, int horizon, float* var_per_waypoint_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMD_AMD_H */
