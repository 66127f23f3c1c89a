// Multi-agent layer pieces that sit right next to the sampler, so a planning round
// (gather -> sample -> pick the sample with the fewest robot-robot collisions -> conflicts) needs no host round trip:
//   * robot-robot collisions of the chosen best paths: RobotPlanarDisk.check_rr_collisions
//     (deps/torch_robotics/torch_robotics/robots/robot_planar_disk.py:173-203) as called by CBS.get_conflicts
//     (mmd/planners/multi_agent/cbs.py:166-246) for equal start times and densification 1;
//   * the 'least_collisions' batch scan (cbs.py:446-458): for every sample of a robot's batch, how many (t, other robot)
//     pairs collide with the other robots' best paths.
#include <hip/hip_runtime.h>

#include "../../include/mmd_amd.h"
#include "common.h"

namespace mmd {

__global__ void rr_collisions_kernel(const float2* __restrict__ paths, int n, int T, float margin,
                                     unsigned char* __restrict__ mask, float2* __restrict__ mid) {
  const size_t tot = (size_t)T * n * n;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
    const int j = idx % n, i = (idx / n) % n, t = idx / ((size_t)n * n);
    const float2 a = paths[(size_t)i * T + t], b = paths[(size_t)j * T + t];
    const float dx = a.x - b.x, dy = a.y - b.y;
    const bool c = sqrtf(dx * dx + dy * dy) < margin && i != j;
    mask[idx] = c ? 1 : 0;
    if (mid) {
      const float nanv = __builtin_nanf("");
      mid[idx] = c ? make_float2((a.x + b.x) / 2.f, (a.y + b.y) / 2.f) : make_float2(nanv, nanv);
    }
  }
}

// one wave per sample trajectory, lane = time step
__global__ __launch_bounds__(256) void count_collisions_kernel(const float4* __restrict__ trajs,
                                                               const float2* __restrict__ paths, int robot0,
                                                               int samples_per_robot, int n_traj, int n_all, float margin,
                                                               int* __restrict__ counts) {
  const int t = threadIdx.x & 63;
  const int traj = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (traj >= n_traj) return;
  const int self = robot0 + traj / samples_per_robot;
  const float4 p = trajs[(size_t)traj * H + t];
  int c = 0;
  for (int j = 0; j < n_all; ++j) {
    if (j == self) continue;
    const float2 q = paths[(size_t)j * H + t];
    const float dx = p.x - q.x, dy = p.y - q.y;
    c += sqrtf(dx * dx + dy * dy) < margin ? 1 : 0;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
  if (t == 0) counts[traj] = c;
}

}  // namespace mmd

using namespace mmd;

extern "C" {

int mmd_rr_collisions(const float* paths_dev, int n_robots, int horizon, float margin, uint8_t* mask_dev,
                      float* midpoints_dev, void* stream) {
  MMD_REQUIRE(paths_dev && mask_dev && n_robots >= 1 && horizon >= 1, "mmd_rr_collisions: bad arguments");
  const size_t tot = (size_t)horizon * n_robots * n_robots;
  int grid = (int)((tot + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(rr_collisions_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float2*)paths_dev,
                     n_robots, horizon, margin, mask_dev, (float2*)midpoints_dev);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

int mmd_count_collisions(const float* trajs_dev, const float* paths_dev, int robot0, int n_local,
                         int samples_per_robot, int n_all, int horizon, float margin, int32_t* counts_dev, void* stream) {
  MMD_REQUIRE(trajs_dev && paths_dev && counts_dev, "mmd_count_collisions: NULL argument");
  MMD_REQUIRE(horizon == H, "horizon must be %d", H);
  MMD_REQUIRE(n_local >= 1 && samples_per_robot >= 1 && robot0 >= 0 && robot0 + n_local <= n_all, "bad robot range");
  const int n_traj = n_local * samples_per_robot;
  hipLaunchKernelGGL(count_collisions_kernel, dim3((n_traj + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)trajs_dev, (const float2*)paths_dev, robot0, samples_per_robot, n_traj, n_all, margin,
                     counts_dev);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
