// Parameter layout of the reference TemporalUnet (state_dict order) and the time-embedding table, shared by the fused kernel's host
// side (unet.hip: unet_input_dim 32, dim_mults (1, 2, 4)) and the layer-by-layer path (unet_layers.hip: every other configuration).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "../../include/mmd_amd.h"

namespace mmd {

constexpr int MAX_LEVELS = 4;                  // dim_mults (1, 2, 4, 8)
constexpr int MAX_RTB = 4 * MAX_LEVELS;        // 2 per down level, 2 per up level, 2 mid blocks

struct Rtb { int cin, cout; bool res; int t_w0, t_b0, t_g0, t_be0, t_w1, t_b1, t_g1, t_be1, t_cw, t_cb, t_rw, t_rb; };

struct Spec {
  int uid = 0, n_levels = 0;
  int dims[MAX_LEVELS + 1] = {};  // channels: state_dim, uid, 2 uid, ...
  std::vector<int64_t> numel;
  std::vector<Rtb> rtb;           // downs.0.0, downs.0.1, downs.1.0, ..., ups.0.0, ..., mid1, mid2 (state_dict order)
  int t_time[4];
  int t_down[MAX_LEVELS - 1][2], t_up[MAX_LEVELS - 1][2];
  int t_final[6];
};

inline void add_rtb(std::vector<int64_t>& sp, std::vector<Rtb>& rtbs, int cin, int cout) {
  Rtb r{};
  r.cin = cin; r.cout = cout; r.res = cin != cout;
  r.t_w0 = sp.size(); sp.push_back((int64_t)cout * cin * 5);
  r.t_b0 = sp.size(); sp.push_back(cout);
  r.t_g0 = sp.size(); sp.push_back(cout);
  r.t_be0 = sp.size(); sp.push_back(cout);
  r.t_w1 = sp.size(); sp.push_back((int64_t)cout * cout * 5);
  r.t_b1 = sp.size(); sp.push_back(cout);
  r.t_g1 = sp.size(); sp.push_back(cout);
  r.t_be1 = sp.size(); sp.push_back(cout);
  r.t_cw = sp.size(); sp.push_back((int64_t)cout * 32);
  r.t_cb = sp.size(); sp.push_back(cout);
  if (r.res) {
    r.t_rw = sp.size(); sp.push_back((int64_t)cout * cin);
    r.t_rb = sp.size(); sp.push_back(cout);
  }
  rtbs.push_back(r);
}

// state_dict order of TemporalUnet(unet_input_dim = uid, dim_mults = (1, 2, 4, 8)[:n_levels]) (temporal_unet.py:25-119): time_mlp,
// downs, ups, mid_block1, mid_block2, final_conv.  Channels must be multiples of 8 (GroupNorm(8), layers.py:392-398).
inline bool build_spec(int uid, int n_levels, Spec& s) {
  if (n_levels < 1 || n_levels > MAX_LEVELS || uid < 8 || uid > 64 || uid % 8) return false;
  s.uid = uid; s.n_levels = n_levels;
  s.dims[0] = 4;
  for (int i = 0; i < n_levels; ++i) s.dims[i + 1] = uid << i;
  const int* dims = s.dims;
  auto& sp = s.numel;
  s.t_time[0] = sp.size(); sp.push_back(128 * 32);
  s.t_time[1] = sp.size(); sp.push_back(128);
  s.t_time[2] = sp.size(); sp.push_back(32 * 128);
  s.t_time[3] = sp.size(); sp.push_back(32);
  for (int i = 0; i < n_levels; ++i) {
    add_rtb(sp, s.rtb, dims[i], dims[i + 1]);
    add_rtb(sp, s.rtb, dims[i + 1], dims[i + 1]);
    if (i < n_levels - 1) {
      s.t_down[i][0] = sp.size(); sp.push_back((int64_t)dims[i + 1] * dims[i + 1] * 3);
      s.t_down[i][1] = sp.size(); sp.push_back(dims[i + 1]);
    }
  }
  for (int i = 0; i < n_levels - 1; ++i) {   // reversed(in_out[1:]): ups.i.0 = RTB(2 * dout, din)
    const int din = dims[n_levels - 1 - i], dout = dims[n_levels - i];
    add_rtb(sp, s.rtb, dout * 2, din);
    add_rtb(sp, s.rtb, din, din);
    s.t_up[i][0] = sp.size(); sp.push_back((int64_t)din * din * 4);
    s.t_up[i][1] = sp.size(); sp.push_back(din);
  }
  add_rtb(sp, s.rtb, dims[n_levels], dims[n_levels]);
  add_rtb(sp, s.rtb, dims[n_levels], dims[n_levels]);
  s.t_final[0] = sp.size(); sp.push_back((int64_t)uid * uid * 5);
  s.t_final[1] = sp.size(); sp.push_back(uid);
  s.t_final[2] = sp.size(); sp.push_back(uid);
  s.t_final[3] = sp.size(); sp.push_back(uid);
  s.t_final[4] = sp.size(); sp.push_back((int64_t)4 * uid);
  s.t_final[5] = sp.size(); sp.push_back(4);
  return true;
}

// time embedding table: TimeEncoder (layers.py:232-258) + every block's cond_mlp (layers.py:337-341) for all integer t
struct TimeArgs {
  const float* w1; const float* b1;   // [128,32], [128]
  const float* w3; const float* b3;   // [32,128], [32]
  const float* cw[MAX_RTB]; const float* cb[MAX_RTB];   // cond_mlp.1 weight [C,32], bias [C]
  int cout[MAX_RTB]; int off[MAX_RTB];
  int n_rtb; int total;
  float* table;                       // [T][total]
};
void launch_time_table(const TimeArgs& a, int T, hipStream_t st);   // (unet.hip)

// The layer-by-layer TemporalUnet (unet_layers.hip): any configuration build_spec accepts; activations in an HBM workspace.
struct LayeredUnet;
int layered_create(LayeredUnet** out, const Spec& s, int T, const float* const* tensors, const mmd_unet_options* opt, hipStream_t st);
void layered_destroy(LayeredUnet* u);
size_t layered_workspace_bytes(const LayeredUnet* u, int n_traj);
size_t layered_weight_bytes(const LayeredUnet* u);
int layered_forward(const LayeredUnet* u, const float* x, int t, float* eps, int n_traj, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace mmd
