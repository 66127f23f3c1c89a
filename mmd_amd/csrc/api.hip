// Sampling-loop orchestration and ABI bookkeeping of libmmd_amd.so.
//
// p_sample_loop (diffusion_model_base.py:162-211) = per outer step: 29 UNet kernels (unet.hip) + ONE fused kernel
// doing posterior mean, the n_guide_steps guide iterations, noise and hard conditioning (guide.hip).  Everything is
// enqueued on the caller's stream with no host synchronisation: the reference's per-step `.item()`-style syncs
// (sample_functions.py:53,63; normalization.py:161) do not exist here because the branches depend only on the
// loop index, which the host knows.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/mmd_amd.h"
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

// side streams of the chunked sampling loop: created once per process, never destroyed
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
  static thread_local Streams S;   // streams belong to the calling thread's current device
  return S;
}

static int make_step(const mmd_sampler_desc* s, int i, bool guided, StepDev& sd) {
  const int t = i < 0 ? 0 : i;                                    // sample_functions.py:53-54
  MMD_REQUIRE(t < s->n_diffusion_steps, "loop index %d outside the %d-step schedule", i, s->n_diffusion_steps);
  sd.a_t = s->sqrt_recip_alphas_cumprod[t];
  sd.b_t = s->sqrt_recipm1_alphas_cumprod[t];
  sd.c1 = s->posterior_mean_coef1[t];
  sd.c2 = s->posterior_mean_coef2[t];
  sd.sigma = expf(0.5f * s->posterior_log_variance_clipped[t]);   // model_std, sample_functions.py:60
  sd.noise_std_extra = s->noise_std_extra;
  sd.do_model = 1;
  sd.do_guide = guided && i < s->t_start_guide ? 1 : 0;           // sample_functions.py:63
  sd.do_noise = t == 0 ? 0 : 1;                                   // noise[t == 0] = 0, sample_functions.py:76
  sd.n_guide_steps = s->n_guide_steps;
  sd.hard_mask = s->hard_mask;
  return 0;
}

}  // namespace mmd

using namespace mmd;

extern "C" {

int mmd_abi_version(void) { return MMD_AMD_ABI_VERSION; }
const char* mmd_last_error(void) { return g_err; }

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
  float* eps = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace_dev) + uws);
  StepDev sd{};
  if (int rc = make_step(s, i, guide != nullptr, sd)) return rc;
  sd.seed = seed; sd.draw = draw_index;
  GuideDev g{};
  if (sd.do_guide)
    if (int rc = fill_guide(guide, g)) return rc;
  if (int rc = mmd_unet_forward(unet, x_dev, i < 0 ? 0 : i, eps, n, workspace_dev, uws, stream)) return rc;
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
  launch_init(x_dev, chain_dev, hard_dev, s->hard_mask, init_noise, (unsigned long long)seed, n, samples_per_robot, st);

  // Split the robots into concurrent chunks: each chunk's kernels go to its own stream, launches interleaved layer by
  // layer so both queues stay fed.  Robots are independent, so results are bit-identical to the unsplit run.
  int nch = s->n_streams;
  if (const char* e = getenv("MMD_AMD_STREAMS")) nch = atoi(e);
  if (nch <= 0) nch = 1;   // auto = off: 2 chunks measured +3 % only, and one stream keeps per-kernel accounting clean
  if (nch > kMaxChunks) nch = kMaxChunks;
  if (nch > n_robots) nch = n_robots;
  Streams& S = streams();
  if (nch > 1 && !S.ok()) nch = 1;
  hipStream_t cs[kMaxChunks];
  int r0[kMaxChunks + 1];
  char* wsp[kMaxChunks];
  size_t wsb[kMaxChunks];
  float* epsp[kMaxChunks];
  {
    char* w = reinterpret_cast<char*>(workspace_dev);
    for (int c = 0; c <= nch; ++c) r0[c] = (int)((long long)n_robots * c / nch);
    for (int c = 0; c < nch; ++c) {
      const int nc = (r0[c + 1] - r0[c]) * samples_per_robot;
      wsp[c] = w;
      wsb[c] = mmd_unet_workspace_bytes(unet, nc);
      epsp[c] = reinterpret_cast<float*>(w + wsb[c]);
      w += mmd_sampler_workspace_bytes(unet, nc);
      cs[c] = nch == 1 ? st : S.s[c];
    }
  }
  if (nch > 1) {
    MMD_HIP_CHECK(hipEventRecord(S.fork, st));
    for (int c = 0; c < nch; ++c) MMD_HIP_CHECK(hipStreamWaitEvent(cs[c], S.fork, 0));
  }
  int k = 0;
  for (int i = n_steps - 1; i >= -n_steps_without_noise; --i, ++k) {
    StepDev sd{};
    if (int rc = make_step(s, i, guide != nullptr, sd)) return rc;
    sd.seed = seed; sd.draw = (unsigned int)k;
    for (int c = 0; c < nch; ++c) {
      const int t0 = r0[c] * samples_per_robot, nc = (r0[c + 1] - r0[c]) * samples_per_robot;
      if (int rc = mmd_unet_forward(unet, x_dev + (size_t)t0 * H * D, i < 0 ? 0 : i, epsp[c], nc, wsp[c], wsb[c], cs[c]))
        return rc;
    }
    for (int c = 0; c < nch; ++c) {
      const int t0 = r0[c] * samples_per_robot, nc = (r0[c + 1] - r0[c]) * samples_per_robot;
      // eps of this chunk is indexed from its own buffer: pass a pointer rebased to the full-array indexing
      launch_step(g, sd, x_dev, epsp[c] - (size_t)t0 * H * D,
                  step_noise_dev ? step_noise_dev + (size_t)k * traj_floats : nullptr,
                  chain_dev ? chain_dev + (size_t)(k + 1) * traj_floats : nullptr, hard_dev, t0, nc, samples_per_robot,
                  cs[c]);
    }
  }
  if (nch > 1)
    for (int c = 0; c < nch; ++c) {
      MMD_HIP_CHECK(hipEventRecord(S.join[c], cs[c]));
      MMD_HIP_CHECK(hipStreamWaitEvent(st, S.join[c], 0));
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
  float* eps = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace_dev) + uws);
  GuideDev g{};
  if (guide)
    if (int rc = fill_guide(guide, g)) return rc;
  launch_init(x_dev, chain_dev, hard_dev, s->hard_mask, init_noise, (unsigned long long)seed, n, samples_per_robot, st);
  for (int k = 0; k + 1 < n_times; ++k) {
    const int t = times[k], tn = times[k + 1];
    MMD_REQUIRE(t >= 0 && t < s->n_diffusion_steps && tn < t, "mmd_ddim_sample: times must decrease inside the schedule");
    StepDev sd{};
    sd.ddim = 1;
    sd.a_t = s->sqrt_recip_alphas_cumprod[t];
    sd.b_t = s->sqrt_recipm1_alphas_cumprod[t];
    sd.c1 = tn < 0 ? 1.f : sqrtf(alphas_cumprod[tn]);               // x = x_start on the last pair (time_next = -1)
    sd.c2 = tn < 0 ? 0.f : sqrtf(1.f - alphas_cumprod[tn]);         // sigma = eta * ... = 0
    sd.do_model = 1;
    sd.do_guide = guide && tn >= 0 && tn < s->t_start_guide ? 1 : 0;  // torch.all(t_next < t_start_guide); none after the break
    sd.do_noise = 0;
    sd.n_guide_steps = s->n_guide_steps;
    sd.hard_mask = s->hard_mask;
    sd.seed = seed; sd.draw = (unsigned int)k;
    if (int rc = mmd_unet_forward(unet, x_dev, t, eps, n, workspace_dev, uws, stream)) return rc;
    launch_step(g, sd, x_dev, eps, nullptr, chain_dev ? chain_dev + (size_t)(k + 1) * traj_floats : nullptr, hard_dev, 0, n,
                samples_per_robot, st);
    if (tn < 0) break;
  }
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
