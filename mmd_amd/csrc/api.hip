// Sampling-loop orchestration and ABI bookkeeping of libmmd_amd.so.
//
// p_sample_loop (diffusion_model_base.py:162-211) = per outer step: ONE UNet launch (unet.hip) + ONE fused kernel doing
// posterior mean, the n_guide_steps guide iterations, noise and hard conditioning (guide.hip).  Everything is
// enqueued on the caller's stream with no host synchronisation: the reference's per-step `.item()`-style syncs
// (sample_functions.py:53,63; normalization.py:161) do not exist here because the branches depend only on the
// loop index, which the host knows.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>

#include "../../include/mmd_amd.h"
#include "../../include/mmd_amd_debug.h"
#include "common.h"
#include "guide_dev.h"

namespace mmd {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

constexpr int kMaxChunks = 4;
// Measurement switches travel in the descriptors (mmd_sampler_desc.flags / .n_streams / .guide_coop_max, mmd_unet_options): the
// library never reads the environment.
//   MMD_SAMPLER_NO_FUSED_STEP: unguided steps run as step-kernel launches instead of inside the UNet launch's tail (A/B of the
//     fused step).
//   MMD_SAMPLER_PERSIST (default OFF): the leading run of unguided steps as persistent launches (unet.hip: unet_persist_kernel).
//     Measured in round 5 and NOT kept as the default (profiles/r05_persist_ab.txt): bitwise-equal results, +1 .. 2 % on the
//     256 .. 1024-trajectory shards (the launch gaps of 50 steps), -1 .. -4 % on the 2048-trajectory headline (one whole-batch
//     persistent launch replaces the two interleaved stream chunks, which are worth more there).

// Number of concurrent stream chunks mmd_p_sample_loop splits n_robots x samples_per_robot trajectories into.  auto (n_streams
// <= 0): 2 chunks as soon as the batch exceeds 512 trajectories, 1 below.  From 2048 on (a chunk alone fills the chip: one workgroup
// per CU) one chunk's guided step kernel runs beside the other chunk's UNet launch and a chunk's forward no longer ends with CUs
// idling until its slowest workgroup is done: +6 .. 12 % on the 32-robot round (profiles/r03b_stream_chunks.txt).  Between 512 and
// 2048 (round 6, profiles/r06_chunk_sweep.txt) a single launch runs four trajectories per workgroup on a fraction of the CUs (640
// trajectories = 160 workgroups, 120 us); two chunks of <= 512 run the two-trajectory kernel side by side on twice the CUs:
// -15 % at 576 / 1280, -13 % at 640 / 1536, -5 % at 1024, -9 % at 2048.  At <= 512 the whole batch already is one
// two-trajectory launch and a split only adds launches (+8 %).  Results are bit-identical either way.
static int stream_chunks(int n_streams, int n_robots, int samples_per_robot) {
  int nch = n_streams;
  if (nch <= 0) nch = (long long)n_robots * samples_per_robot > 512 ? 2 : 1;
  if (nch > kMaxChunks) nch = kMaxChunks;
  if (nch > n_robots) nch = n_robots;
  return nch < 1 ? 1 : nch;
}


// side streams of the chunked sampling loop: created once per (thread, device), never destroyed
struct Streams {
  hipStream_t s[kMaxChunks] = {};
  hipEvent_t fork = nullptr, join[kMaxChunks] = {};
  bool tried = false, good = false;
  bool ok() {
    if (!tried) {
      tried = true;
      good = hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess;
      for (int c = 0; c < kMaxChunks && good; ++c)
        good = hipStreamCreateWithFlags(&s[c], hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&join[c], hipEventDisableTiming) == hipSuccess;
    }
    return good;
  }
};
static Streams& streams() {
  static thread_local std::map<int, Streams> S;   // keyed by the device that is current at the call
  int dev = 0;
  (void)hipGetDevice(&dev);
  return S[dev];
}

static int make_step(const mmd_sampler_desc* s, int i, bool guided, StepDev& sd) {
  const int t = i < 0 ? 0 : i;                                    // sample_functions.py:53-54
  MMD_REQUIRE(t < s->n_diffusion_steps, "loop index %d outside the %d-step schedule", i, s->n_diffusion_steps);
  sd.a_t = s->sqrt_recip_alphas_cumprod[t];
  sd.b_t = s->sqrt_recipm1_alphas_cumprod[t];
  if (s->model_predicts_x0) { sd.a_t = 0.f; sd.b_t = -1.f; }       // predict_epsilon = False: x_recon = a x - b out = out (diffusion_model_base.py:131-141)
  // scale_grad_by_std (sample_functions.py:59-61, 100-101): the guide gradient times model_var = exp(posterior_log_variance_clipped[t])
  sd.grad_scale = s->scale_grad_by_std ? expf(s->posterior_log_variance_clipped[t]) : 1.f;
  sd.c1 = s->posterior_mean_coef1[t];
  sd.c2 = s->posterior_mean_coef2[t];
  sd.sigma = expf(0.5f * s->posterior_log_variance_clipped[t]);   // model_std, sample_functions.py:60
  // noise_std_extra_schedule_fn(t_single), evaluated per step (sample_functions.py:83-86)
  sd.noise_std_extra = s->noise_std_extra_by_t ? s->noise_std_extra_by_t[t] : s->noise_std_extra;
  sd.do_model = 1;
  sd.do_guide = guided && i < s->t_start_guide ? 1 : 0;           // sample_functions.py:63
  sd.do_noise = t == 0 ? 0 : 1;                                   // noise[t == 0] = 0, sample_functions.py:76
  sd.n_guide_steps = s->n_guide_steps;
  sd.hard_rows = s->hard_rows; sd.n_hard = __builtin_popcountll(s->hard_rows);
  sd.traj_base = (long long)s->traj_index_base;
  sd.robot_seeds = reinterpret_cast<const unsigned long long*>(s->robot_seeds_dev);
  sd.coop_max = s->guide_coop_max;
  return 0;
}

static inline float* eps_of(void* workspace_dev, mmd_unet_t unet, int n) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(workspace_dev) + mmd_unet_workspace_bytes(unet, n));
}

}  // namespace mmd

using namespace mmd;

extern "C" {

int mmd_abi_version(void) { return MMD_AMD_ABI_VERSION; }
int mmd_sampler_stream_chunks(int n_streams, int n_robots, int samples_per_robot) {
  return stream_chunks(n_streams, n_robots, samples_per_robot);
}
const char* mmd_last_error(void) { return g_err; }

// [UNet token][eps of all n trajectories]; stream chunks use slices of the one eps block
size_t mmd_sampler_workspace_bytes(mmd_unet_t unet, int n_traj) {
  return mmd_unet_workspace_bytes(unet, n_traj) + (size_t)(n_traj > 0 ? n_traj : 0) * H * D * sizeof(float);
}

int mmd_ddpm_step(mmd_unet_t unet, const mmd_sampler_desc* s, const mmd_guide_desc* guide, float* x_dev,
                  const float* hard_dev, int n_robots, int samples_per_robot, int i, const float* noise_dev,
                  uint64_t seed, uint32_t draw_index, void* workspace_dev, size_t workspace_bytes, void* stream) {
  MMD_REQUIRE(unet && s && x_dev && hard_dev && workspace_dev, "mmd_ddpm_step: NULL argument");
  const int n = n_robots * samples_per_robot;
  MMD_REQUIRE(n >= 1, "mmd_ddpm_step: empty batch");
  MMD_REQUIRE(workspace_bytes >= mmd_sampler_workspace_bytes(unet, n), "mmd_ddpm_step: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const size_t uws = mmd_unet_workspace_bytes(unet, n);
  float* eps = eps_of(workspace_dev, unet, n);
  StepDev sd{};
  if (int rc = make_step(s, i, guide != nullptr, sd)) return rc;
  sd.seed = seed; sd.draw = draw_index;
  GuideDev g{};
  if (sd.do_guide)
    if (int rc = fill_guide(guide, g)) return rc;
  if (int rc = mmd_unet_forward_profiled(unet, x_dev, i < 0 ? 0 : i, eps, n, workspace_dev, uws,
                                         (mmd_profiler_t)s->profiler, stream))
    return rc;
  launch_step(g, sd, x_dev, eps, noise_dev, nullptr, hard_dev, 0, n, samples_per_robot, st);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_debug_ddpm_step_trace(mmd_unet_t unet, const mmd_sampler_desc* s, const mmd_guide_desc* guide, float* x_dev,
                              const float* hard_dev, int n_robots, int samples_per_robot, int i, const float* noise_dev,
                              uint64_t seed, uint32_t draw_index, void* workspace_dev, size_t workspace_bytes,
                              float* mu_dev, float* guide_chain_dev, uint32_t* trace_dev, void* stream) {
  MMD_REQUIRE(unet && s && guide && x_dev && hard_dev && workspace_dev && guide_chain_dev && trace_dev,
              "mmd_debug_ddpm_step_trace: NULL argument");
  const int n = n_robots * samples_per_robot;
  MMD_REQUIRE(n >= 1, "mmd_debug_ddpm_step_trace: empty batch");
  MMD_REQUIRE(workspace_bytes >= mmd_sampler_workspace_bytes(unet, n), "mmd_debug_ddpm_step_trace: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* eps = eps_of(workspace_dev, unet, n);
  StepDev sd{};
  if (int rc = make_step(s, i, true, sd)) return rc;
  MMD_REQUIRE(sd.do_guide, "mmd_debug_ddpm_step_trace: step %d is not a guided one (t_start_guide %d)", i, s->t_start_guide);
  sd.seed = seed; sd.draw = draw_index;
  sd.guide_chain = reinterpret_cast<float4*>(guide_chain_dev);
  sd.guide_chain_stride = (long long)n * H;
  sd.trace = trace_dev;
  sd.mu_out = reinterpret_cast<float4*>(mu_dev);
  GuideDev g{};
  if (int rc = fill_guide(guide, g)) return rc;
  if (int rc = mmd_unet_forward(unet, x_dev, i < 0 ? 0 : i, eps, n, workspace_dev, mmd_unet_workspace_bytes(unet, n), stream))
    return rc;
  launch_step(g, sd, x_dev, eps, noise_dev, nullptr, hard_dev, 0, n, samples_per_robot, st);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_p_sample_loop(mmd_unet_t unet, const mmd_sampler_desc* s, const mmd_guide_desc* guide, float* x_dev,
                      const float* hard_dev, int n_robots, int samples_per_robot, int n_steps,
                      int n_steps_without_noise, int init_noise, const float* step_noise_dev, uint64_t seed,
                      float* chain_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  MMD_REQUIRE(unet && s && x_dev && hard_dev && workspace_dev, "mmd_p_sample_loop: NULL argument");
  const int n = n_robots * samples_per_robot;
  MMD_REQUIRE(n >= 1, "mmd_p_sample_loop: empty batch");
  MMD_REQUIRE(n_steps >= 0 && n_steps <= s->n_diffusion_steps && n_steps_without_noise >= 0, "bad step counts");
  MMD_REQUIRE(workspace_bytes >= mmd_sampler_workspace_bytes(unet, n), "mmd_p_sample_loop: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const size_t traj_floats = (size_t)n * H * D;
  GuideDev g{};
  if (guide)
    if (int rc = fill_guide(guide, g)) return rc;
  launch_init(x_dev, chain_dev, hard_dev, s->hard_rows, init_noise, (unsigned long long)seed,
              reinterpret_cast<const unsigned long long*>(s->robot_seeds_dev), (long long)s->traj_index_base, n, samples_per_robot, st);
  const bool persist = (s->flags & MMD_SAMPLER_PERSIST) != 0, no_fused_step = (s->flags & MMD_SAMPLER_NO_FUSED_STEP) != 0;

  // OPT-IN (mmd_sampler_desc.flags & MMD_SAMPLER_PERSIST): the leading run of steps WITHOUT guidance (i >= t_start_guide; every step of a
  // prior-only call) as persistent launches on the caller's stream: a workgroup iterates the run's steps on its own trajectories
  // (unet.hip: unet_persist_kernel), <= 64 steps a launch.  Bitwise the launch-per-step result.  Not with a profiler attached (its
  // brackets are per launch).
  int k_start = 0;
  // (its step table travels by hipMemcpyAsync from host memory: not inside a stream capture)
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (persist) (void)hipStreamIsCapturing(st, &capturing);
  if (persist && capturing == hipStreamCaptureStatusNone && !no_fused_step && unet_fused_step_supported(unet) && !s->profiler) {
    FusedStep run[PERSIST_MAX_STEPS];
    int i = n_steps - 1, k0 = 0;
    while (i >= -n_steps_without_noise) {
      int m = 0;
      for (; m < PERSIST_MAX_STEPS && i - m >= -n_steps_without_noise; ++m) {
        StepDev sd{};
        if (int rc = make_step(s, i - m, guide != nullptr, sd)) return rc;
        if (sd.do_guide) break;
        FusedStep& fs = run[m];
        fs = FusedStep{};
        fs.enabled = 1;
        fs.a_t = sd.a_t; fs.b_t = sd.b_t; fs.c1 = sd.c1; fs.c2 = sd.c2; fs.sigma = sd.sigma; fs.noise_std_extra = sd.noise_std_extra;
        fs.do_noise = sd.do_noise; fs.hard_rows = sd.hard_rows; fs.n_hard = sd.n_hard; fs.seed = seed; fs.draw = (unsigned int)(k0 + m);
        fs.robot_seeds = sd.robot_seeds;
        fs.traj_base = sd.traj_base; fs.traj0 = 0; fs.spr = samples_per_robot;
        fs.x = reinterpret_cast<float4*>(x_dev);
        fs.noise = step_noise_dev ? reinterpret_cast<const float4*>(step_noise_dev + (size_t)(k0 + m) * traj_floats) : nullptr;
        fs.chain = chain_dev ? reinterpret_cast<float4*>(chain_dev + (size_t)(k0 + m + 1) * traj_floats) : nullptr;
        fs.hard = reinterpret_cast<const float4*>(hard_dev);
        fs.t_row = i - m < 0 ? 0 : i - m;
      }
      if (m == 0 || (m < 2 && k0 == 0)) break;              // (a single step gains nothing over the fused launch)
      if (int rc = unet_persist_steps(unet, n, workspace_dev, mmd_unet_workspace_bytes(unet, n), st, run, m)) return rc;
      k0 += m; i -= m;
    }
    k_start = k0;
  }
  // Split the robots into concurrent chunks: each chunk's kernels go to its own stream, launches interleaved layer by
  // layer so both queues stay fed.  Robots are independent and the noise is keyed by the global trajectory index, so
  // results are bit-identical to the unsplit run.
  int nch = stream_chunks(s->n_streams, n_robots, samples_per_robot);
  if (!unet_fused_step_supported(unet)) nch = 1;   // the layer-by-layer path keeps its activations in the ONE workspace
  Streams* S = nullptr;
  if (nch > 1) {
    S = &streams();
    if (!S->ok()) nch = 1;
  }
  hipStream_t cs[kMaxChunks];
  int r0[kMaxChunks + 1];
  for (int c = 0; c <= nch; ++c) r0[c] = (int)((long long)n_robots * c / nch);
  for (int c = 0; c < nch; ++c) cs[c] = nch == 1 ? st : S->s[c];
  const size_t uws = mmd_unet_workspace_bytes(unet, n);
  float* eps = eps_of(workspace_dev, unet, n);           // chunk c uses rows [r0[c] * spr, r0[c+1] * spr) of it
  if (nch > 1) {
    MMD_HIP_CHECK(hipEventRecord(S->fork, st));
    for (int c = 0; c < nch; ++c) MMD_HIP_CHECK(hipStreamWaitEvent(cs[c], S->fork, 0));
  }
  int rc = 0, k = k_start;                                  // (the loop resumes behind the persistent run)
  for (int i = n_steps - 1 - k_start; i >= -n_steps_without_noise && rc == 0; --i, ++k) {
    StepDev sd{};
    if ((rc = make_step(s, i, guide != nullptr, sd))) break;
    sd.seed = seed; sd.draw = (unsigned int)k;
    // A step without guidance is fused into the tail of the UNet launch (unet.hip: the wave that holds a trajectory's eps applies
    // ddpm_sample_fn to it): one launch and one dependent dispatch less per (step, chunk).
    const bool fused = !sd.do_guide && !no_fused_step && unet_fused_step_supported(unet);
    const float* noise_k = step_noise_dev ? step_noise_dev + (size_t)k * traj_floats : nullptr;
    float* chain_k = chain_dev ? chain_dev + (size_t)(k + 1) * traj_floats : nullptr;
    for (int c = 0; c < nch && rc == 0; ++c) {
      const int t0 = r0[c] * samples_per_robot, nc = (r0[c + 1] - r0[c]) * samples_per_robot;
      if (fused) {
        FusedStep fs{};
        fs.enabled = 1;
        fs.a_t = sd.a_t; fs.b_t = sd.b_t; fs.c1 = sd.c1; fs.c2 = sd.c2; fs.sigma = sd.sigma; fs.noise_std_extra = sd.noise_std_extra;
        fs.do_noise = sd.do_noise; fs.hard_rows = sd.hard_rows; fs.n_hard = sd.n_hard; fs.seed = sd.seed; fs.draw = sd.draw; fs.traj_base = sd.traj_base;
        fs.robot_seeds = sd.robot_seeds;
        fs.traj0 = t0; fs.spr = samples_per_robot;
        fs.x = reinterpret_cast<float4*>(x_dev); fs.noise = reinterpret_cast<const float4*>(noise_k);
        fs.chain = reinterpret_cast<float4*>(chain_k); fs.hard = reinterpret_cast<const float4*>(hard_dev);
        rc = unet_forward_fused(unet, x_dev + (size_t)t0 * H * D, i < 0 ? 0 : i, eps + (size_t)t0 * H * D, nc, workspace_dev, uws,
                                (mmd_profiler_t)s->profiler, cs[c], fs);
        prof_skip((mmd_profiler_t)s->profiler, 1);
      } else {
        rc = mmd_unet_forward_profiled(unet, x_dev + (size_t)t0 * H * D, i < 0 ? 0 : i, eps + (size_t)t0 * H * D, nc,
                                       workspace_dev, uws, (mmd_profiler_t)s->profiler, cs[c]);
      }
    }
    for (int c = 0; c < nch && rc == 0 && !fused; ++c) {
      const int t0 = r0[c] * samples_per_robot, nc = (r0[c + 1] - r0[c]) * samples_per_robot;
      const bool br = prof_begin((mmd_profiler_t)s->profiler, 1, sd.do_guide ? MMD_PROF_STEP_GUIDED : MMD_PROF_STEP_PLAIN, cs[c]);
      launch_step(g, sd, x_dev, eps, noise_k, chain_k, hard_dev, t0, nc, samples_per_robot, cs[c]);
      if (br) prof_end((mmd_profiler_t)s->profiler, cs[c]);
    }
  }
  // join on every exit path: an error above must not leave the caller's stream detached from the side streams
  if (nch > 1)
    for (int c = 0; c < nch; ++c) {
      if (hipEventRecord(S->join[c], cs[c]) != hipSuccess || hipStreamWaitEvent(st, S->join[c], 0) != hipSuccess) {
        if (rc == 0) { set_error("mmd_p_sample_loop: joining side stream %d failed", c); rc = 1; }
      }
    }
  if (rc) return rc;
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_p_sample_loop_ensemble(const mmd_ensemble_tile* tiles, int n_tiles, const mmd_cross_cond* cross, int n_cross,
                               int n_robots, int samples_per_robot, int n_steps, int n_steps_without_noise,
                               int init_noise, void* workspace_dev, size_t workspace_bytes, void* stream) {
  MMD_REQUIRE(tiles && n_tiles >= 1 && (n_cross == 0 || cross) && workspace_dev, "mmd_p_sample_loop_ensemble: NULL argument");
  const int n = n_robots * samples_per_robot;
  MMD_REQUIRE(n >= 1 && n_steps >= 0 && n_steps_without_noise >= 0, "mmd_p_sample_loop_ensemble: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const size_t traj_floats = (size_t)n * H * D;
  GuideDev g[8];
  MMD_REQUIRE(n_tiles <= 8, "mmd_p_sample_loop_ensemble: at most 8 tiles");
  for (int m = 0; m < n_tiles; ++m) {
    const mmd_ensemble_tile& T = tiles[m];
    MMD_REQUIRE(T.unet && T.sampler && T.x_dev && T.hard_dev, "tile %d: NULL member", m);
    MMD_REQUIRE(n_steps <= T.sampler->n_diffusion_steps, "tile %d: more steps than its schedule has", m);
    MMD_REQUIRE(workspace_bytes >= mmd_sampler_workspace_bytes(T.unet, n), "mmd_p_sample_loop_ensemble: workspace too small");
    g[m] = GuideDev{};
    if (T.guide)
      if (int rc = fill_guide(T.guide, g[m])) return rc;
  }
  for (int c = 0; c < n_cross; ++c)
    MMD_REQUIRE(cross[c].m1 >= 0 && cross[c].m1 < n_tiles && cross[c].m2 >= 0 && cross[c].m2 < n_tiles &&
                    cross[c].ind1 >= 0 && cross[c].ind1 < H && cross[c].ind2 >= 0 && cross[c].ind2 < H,
                "cross condition %d out of range", c);
  // apply_cross_conditioning (sample_functions.py:17-31) over all pairs, after tile `stepped` has taken its step of the outer step
  // that produces chain row `crow`.  The chains are kept in step the way the reference's are: it appends the tensor OBJECT x[m]
  // (diffusion_ensemble.py:101-103) and the stitching writes into x[m] in place, so a tile that has not stepped yet in this outer
  // step (index > stepped) still IS its previous chain row, crow - 1, and that row receives the stitched boundary too (from 3 tiles
  // on this is visible: re-stitching two not-yet-stepped neighbours is not idempotent in the clamped coordinate).
  auto cross_all = [&](int crow, int stepped) {
    for (int c = 0; c < n_cross; ++c) {
      const mmd_cross_cond& C = cross[c];
      auto row = [&](int m) -> float* {
        return tiles[m].chain_dev ? tiles[m].chain_dev + (size_t)(m > stepped ? crow - 1 : crow) * traj_floats : nullptr;
      };
      launch_cross(tiles[C.m1].x_dev, tiles[C.m2].x_dev, row(C.m1), row(C.m2), C.ind1, C.ind2, C.rel, C.boundary, C.by_robot_dev,
                   samples_per_robot, n, st);
    }
  };
  // x_T per tile (diffusion_ensemble.py:66-81): draw / keep, hard conditioning, then cross conditioning; chain[0]
  for (int m = 0; m < n_tiles; ++m)
    launch_init(tiles[m].x_dev, tiles[m].chain_dev, tiles[m].hard_dev, tiles[m].sampler->hard_rows, init_noise,
                (unsigned long long)tiles[m].seed, reinterpret_cast<const unsigned long long*>(tiles[m].sampler->robot_seeds_dev),
                (long long)tiles[m].sampler->traj_index_base, n, samples_per_robot, st);
  cross_all(0, n_tiles);
  int k = 0;
  for (int i = n_steps - 1; i >= -n_steps_without_noise; --i, ++k) {
    // tiles step IN ORDER inside an outer step and every tile's step is followed by the cross conditioning of all pairs
    // (diffusion_ensemble.py:86-100): tile m+1's UNet input already carries the boundary row stitched after tile m's
    // step, so the tiles of one outer step cannot be merged into one batched launch without changing the result
    for (int m = 0; m < n_tiles; ++m) {
      const mmd_ensemble_tile& T = tiles[m];
      StepDev sd{};
      if (int rc = make_step(T.sampler, i, T.guide != nullptr, sd)) return rc;
      sd.seed = T.seed; sd.draw = (unsigned int)k;
      float* eps = eps_of(workspace_dev, T.unet, n);
      const float* noise_k = T.step_noise_dev ? T.step_noise_dev + (size_t)k * traj_floats : nullptr;
      float* chain_k = T.chain_dev ? T.chain_dev + (size_t)(k + 1) * traj_floats : nullptr;
      if (!sd.do_guide && !(T.sampler->flags & MMD_SAMPLER_NO_FUSED_STEP) && unet_fused_step_supported(T.unet)) {
        // a step without guidance rides in the tail of the UNet launch, as in mmd_p_sample_loop (one launch and one dependent
        // dispatch less per tile step; bitwise the two-launch form)
        FusedStep fs{};
        fs.enabled = 1;
        fs.a_t = sd.a_t; fs.b_t = sd.b_t; fs.c1 = sd.c1; fs.c2 = sd.c2; fs.sigma = sd.sigma; fs.noise_std_extra = sd.noise_std_extra;
        fs.do_noise = sd.do_noise; fs.hard_rows = sd.hard_rows; fs.n_hard = sd.n_hard; fs.seed = sd.seed; fs.draw = sd.draw;
        fs.traj_base = sd.traj_base; fs.robot_seeds = sd.robot_seeds; fs.traj0 = 0; fs.spr = samples_per_robot;
        fs.x = reinterpret_cast<float4*>(T.x_dev); fs.noise = reinterpret_cast<const float4*>(noise_k);
        fs.chain = reinterpret_cast<float4*>(chain_k); fs.hard = reinterpret_cast<const float4*>(T.hard_dev);
        if (int rc = unet_forward_fused(T.unet, T.x_dev, i < 0 ? 0 : i, eps, n, workspace_dev, mmd_unet_workspace_bytes(T.unet, n),
                                        (mmd_profiler_t)T.sampler->profiler, st, fs))
          return rc;
      } else {
        if (int rc = mmd_unet_forward_profiled(T.unet, T.x_dev, i < 0 ? 0 : i, eps, n, workspace_dev,
                                               mmd_unet_workspace_bytes(T.unet, n), (mmd_profiler_t)T.sampler->profiler, st))
          return rc;
        launch_step(g[m], sd, T.x_dev, eps, noise_k, chain_k, T.hard_dev, 0, n, samples_per_robot, st);
      }
      cross_all(k + 1, m);
    }
  }
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_ddim_sample(mmd_unet_t unet, const mmd_sampler_desc* s, const float* alphas_cumprod, const int32_t* times,
                    int n_times, const mmd_guide_desc* guide, float* x_dev, const float* hard_dev, int n_robots,
                    int samples_per_robot, int init_noise, uint64_t seed, float* chain_dev, void* workspace_dev,
                    size_t workspace_bytes, void* stream) {
  MMD_REQUIRE(unet && s && alphas_cumprod && times && x_dev && hard_dev && workspace_dev, "mmd_ddim_sample: NULL argument");
  const int n = n_robots * samples_per_robot;
  MMD_REQUIRE(n >= 1 && n_times >= 2, "mmd_ddim_sample: empty batch or fewer than two times");
  MMD_REQUIRE(workspace_bytes >= mmd_sampler_workspace_bytes(unet, n), "mmd_ddim_sample: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const size_t traj_floats = (size_t)n * H * D;
  const size_t uws = mmd_unet_workspace_bytes(unet, n);
  float* eps = eps_of(workspace_dev, unet, n);
  GuideDev g{};
  if (guide)
    if (int rc = fill_guide(guide, g)) return rc;
  launch_init(x_dev, chain_dev, hard_dev, s->hard_rows, init_noise, (unsigned long long)seed,
              reinterpret_cast<const unsigned long long*>(s->robot_seeds_dev), (long long)s->traj_index_base, n, samples_per_robot, st);
  for (int k = 0; k + 1 < n_times; ++k) {
    const int t = times[k], tn = times[k + 1];
    MMD_REQUIRE(t >= 0 && t < s->n_diffusion_steps && tn < t, "mmd_ddim_sample: times must decrease inside the schedule");
    StepDev sd{};
    sd.ddim = s->model_predicts_x0 ? 2 : 1;
    sd.grad_scale = 1.f;                                              // (ddim_sample passes no scale_grad_by_std on)
    sd.a_t = s->sqrt_recip_alphas_cumprod[t];
    // (x0-predicting model: b_t carries 1 / sqrt_recipm1; on the last pair x = x_start and pred_noise is never used -- 0 there, so a
    // schedule whose sqrt_recipm1[t] is 0 or denormal cannot put 0 * inf into the update)
    sd.b_t = s->model_predicts_x0 ? (tn < 0 ? 0.f : 1.f / s->sqrt_recipm1_alphas_cumprod[t]) : s->sqrt_recipm1_alphas_cumprod[t];
    sd.c1 = tn < 0 ? 1.f : sqrtf(alphas_cumprod[tn]);               // x = x_start on the last pair (time_next = -1)
    sd.c2 = tn < 0 ? 0.f : sqrtf(1.f - alphas_cumprod[tn]);         // sigma = eta * ... = 0
    sd.do_model = 1;
    sd.do_guide = guide && tn >= 0 && tn < s->t_start_guide ? 1 : 0;  // torch.all(t_next < t_start_guide); none after the break
    sd.do_noise = 0;
    sd.n_guide_steps = s->n_guide_steps;
    sd.hard_rows = s->hard_rows; sd.n_hard = __builtin_popcountll(s->hard_rows);
    sd.seed = seed; sd.draw = (unsigned int)k;
    sd.traj_base = (long long)s->traj_index_base;
    if (int rc = mmd_unet_forward_profiled(unet, x_dev, t, eps, n, workspace_dev, uws, (mmd_profiler_t)s->profiler, stream))
      return rc;
    launch_step(g, sd, x_dev, eps, nullptr, chain_dev ? chain_dev + (size_t)(k + 1) * traj_floats : nullptr, hard_dev, 0, n,
                samples_per_robot, st);
    if (tn < 0) break;
  }
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
