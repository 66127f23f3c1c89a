// TemporalUnet forward, layer by layer, for the configurations the fused kernel (unet.hip) is not instantiated for -- first of all
// UNET_DIM_MULTS[1] = (1, 2, 4, 8) (mmd/models/diffusion_models/temporal_unet.py:17-20, selected by a checkpoint's args.yaml at
// mmd/planners/single_agent/mpd.py:158; the released checkpoints use option 0 and run the fused kernel).  Plain fp32 FMA
// arithmetic, one launch per Conv1dBlock / conv, activations channels-last [n][L][C] in an HBM workspace: a correct path for a
// rarely used configuration, NOT a tuned one (about 40 launches and ~0.5-5 ms per forward instead of one launch and 0.2 ms).
//
//   ResidualTemporalBlock (layers.py:323-358): out = Mish(GN(conv5(Mish(GN(conv5(x))) + time bias))) + res(x)
//   Downsample1d = Conv1d(k3, s2, p1), Upsample1d = ConvTranspose1d(k4, s2, p1) (layers.py:261-279)
//   final_conv = Conv1dBlock(k5) + Conv1d(k1) (temporal_unet.py:104-110)
//
// A workgroup owns one sample of a layer: the input rows are staged in LDS, every thread computes outputs (position, channel) with
// the weights transposed to [tap][c_in][c_out] at create time (adjacent threads = adjacent output channels read adjacent weights),
// the conv output stays in LDS for the GroupNorm (8 groups, two-pass statistics, eps 1e-5) + Mish + addend, then goes to HBM.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/mmd_amd.h"
#include "common.h"
#include "unet_spec.h"

namespace mmd {

namespace {

constexpr int N_GROUPS = 8;                    // group_norm_n_groups(c) = 8 for every multiple of 8 (layers.py:392-398)

__device__ __forceinline__ float mish_ref(float y) {          // torch.nn.Mish: y tanh(softplus(y)), softplus threshold 20
  const float sp = y > 20.f ? y : log1pf(expf(y));
  return y * tanhf(sp);
}

struct ConvArgs {
  const float* x1; const float* x2;   // input [n][L_in][C1] (+ [n][L_in][C2] concatenated behind it along the channels, or NULL)
  int c1, c2, l_in, l_out, c_out;
  const float* wt;                    // [taps][c1 + c2][c_out]
  const float* bias;                  // [c_out]
  const float* gamma; const float* beta;   // GroupNorm affine, or NULL: plain conv
  const float* add_c;                 // per-channel addend after Mish (time bias), or NULL
  const float* add_t;                 // [n][l_out][c_out] addend after Mish (residual), or NULL
  float* y;                           // [n][l_out][c_out]
};

// MODE 0: Conv1d, K taps, stride S, padding K / 2.  MODE 1: ConvTranspose1d(k4, s2, p1): out[2 m] = in[m - 1] W3 + in[m] W1,
// out[2 m + 1] = in[m] W2 + in[m + 1] W0 (taps stored in kernel-index order 0 .. 3)
template <int MODE, int K, int S>
__global__ __launch_bounds__(256) void conv_block_kernel(ConvArgs a) {
  extern __shared__ float lds[];
  const int cin = a.c1 + a.c2, tid = threadIdx.x;
  const size_t n = blockIdx.x;
  float* xin = lds;                              // [l_in][cin]
  float* out = lds + a.l_in * cin;               // [l_out][c_out]
  float* red = out + a.l_out * a.c_out;          // [2 * N_GROUPS] group statistics
  for (int i = tid; i < a.l_in * cin; i += 256) {
    const int l = i / cin, c = i % cin;
    xin[i] = c < a.c1 ? a.x1[(n * a.l_in + l) * a.c1 + c] : a.x2[(n * a.l_in + l) * a.c2 + (c - a.c1)];
  }
  __syncthreads();
  const int n_out = a.l_out * a.c_out;
  for (int o = tid; o < n_out; o += 256) {
    const int lo = o / a.c_out, co = o % a.c_out;
    float acc = a.bias[co];
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int li = lo * S + k - K / 2;
        if (li < 0 || li >= a.l_in) continue;
        const float* xr = xin + li * cin;
        const float* wr = a.wt + (size_t)k * cin * a.c_out + co;
        for (int ci = 0; ci < cin; ++ci) acc = fmaf(xr[ci], wr[(size_t)ci * a.c_out], acc);
      }
    } else {
      const int m = lo >> 1, par = lo & 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int li = par ? m + j : m - 1 + j;            // parity 0: (m - 1, W3), (m, W1); parity 1: (m, W2), (m + 1, W0)
        const int k = par ? 2 - 2 * j : 3 - 2 * j;
        if (li < 0 || li >= a.l_in) continue;
        const float* xr = xin + li * cin;
        const float* wr = a.wt + (size_t)k * cin * a.c_out + co;
        for (int ci = 0; ci < cin; ++ci) acc = fmaf(xr[ci], wr[(size_t)ci * a.c_out], acc);
      }
    }
    out[o] = acc;
  }
  __syncthreads();
  if (a.gamma) {
    // GroupNorm(8, c_out) over (c_out / 8 channels) x l_out positions per group: wave w reduces group w (256 threads = 4 waves: two
    // groups each), mean first, then the centred second moment
    const int cpg = a.c_out / N_GROUPS, per = cpg * a.l_out, lane = tid & 63, wave = tid >> 6;
    for (int g = wave; g < N_GROUPS; g += 4) {
      float s = 0.f;
      for (int i = lane; i < per; i += 64) s += out[(i / cpg) * a.c_out + g * cpg + i % cpg];
      for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
      const float mean = s / (float)per;
      float q = 0.f;
      for (int i = lane; i < per; i += 64) {
        const float d = out[(i / cpg) * a.c_out + g * cpg + i % cpg] - mean;
        q = fmaf(d, d, q);
      }
      for (int off = 32; off; off >>= 1) q += __shfl_xor(q, off);
      if (lane == 0) {
        red[2 * g] = mean;
        red[2 * g + 1] = 1.f / sqrtf(q / (float)per + 1e-5f);
      }
    }
    __syncthreads();
  }
  for (int o = tid; o < n_out; o += 256) {
    const int co = o % a.c_out;
    float v = out[o];
    if (a.gamma) {
      const int g = co / (a.c_out / N_GROUPS);
      v = mish_ref((v - red[2 * g]) * red[2 * g + 1] * a.gamma[co] + a.beta[co]);
    }
    if (a.add_c) v += a.add_c[co];
    if (a.add_t) v += a.add_t[n * n_out + o];
    a.y[n * n_out + o] = v;
  }
}

// weight [c_out][c_in][k] (Conv1d) or [c_in][c_out][k] (ConvTranspose1d) -> [k][c_in][c_out]
size_t push_wt(std::vector<float>& blob, const float* w, int cout, int cin, int k, bool transposed) {
  const size_t off = blob.size();
  blob.resize(off + (size_t)k * cin * cout);
  for (int kk = 0; kk < k; ++kk)
    for (int ci = 0; ci < cin; ++ci)
      for (int co = 0; co < cout; ++co)
        blob[off + ((size_t)kk * cin + ci) * cout + co] =
            transposed ? w[((size_t)ci * cout + co) * k + kk] : w[((size_t)co * cin + ci) * k + kk];
  return off;
}
size_t push_v(std::vector<float>& blob, const float* p, int64_t n) {
  const size_t off = blob.size();
  blob.insert(blob.end(), p, p + n);
  return off;
}

struct LRtb { int cin, cout; size_t wa, ba, ga, bea, wb, bb, gb, beb, wr, br; bool res; int tb_off; };

}  // namespace

struct LayeredUnet {
  Spec spec;
  int T = 0, tb_total = 0;
  float* blob = nullptr;
  float* ttable = nullptr;
  std::vector<LRtb> rtb;
  size_t down_w[MAX_LEVELS - 1], down_b[MAX_LEVELS - 1], up_w[MAX_LEVELS - 1], up_b[MAX_LEVELS - 1];
  size_t fin_w5, fin_b5, fin_g, fin_be, fin_w1, fin_b1;
  int per_sample = 0;                 // floats of the largest activation tensor of a sample (64 x unet_input_dim)
};

int layered_create(LayeredUnet** out, const Spec& s, int T, const float* const* tensors, hipStream_t st) {
  auto* u = new LayeredUnet();
  u->spec = s;
  u->T = T;
  u->per_sample = H * s.uid;
  std::vector<float> blob;
  size_t raw_time[4], raw_cw[MAX_RTB], raw_cb[MAX_RTB];
  for (int i = 0; i < 4; ++i) raw_time[i] = push_v(blob, tensors[s.t_time[i]], s.numel[s.t_time[i]]);
  int tb = 0;
  for (size_t r = 0; r < s.rtb.size(); ++r) {
    const Rtb& R = s.rtb[r];
    LRtb L{};
    L.cin = R.cin; L.cout = R.cout; L.res = R.res;
    L.wa = push_wt(blob, tensors[R.t_w0], R.cout, R.cin, 5, false);
    L.ba = push_v(blob, tensors[R.t_b0], R.cout); L.ga = push_v(blob, tensors[R.t_g0], R.cout); L.bea = push_v(blob, tensors[R.t_be0], R.cout);
    L.wb = push_wt(blob, tensors[R.t_w1], R.cout, R.cout, 5, false);
    L.bb = push_v(blob, tensors[R.t_b1], R.cout); L.gb = push_v(blob, tensors[R.t_g1], R.cout); L.beb = push_v(blob, tensors[R.t_be1], R.cout);
    if (R.res) { L.wr = push_wt(blob, tensors[R.t_rw], R.cout, R.cin, 1, false); L.br = push_v(blob, tensors[R.t_rb], R.cout); }
    raw_cw[r] = push_v(blob, tensors[R.t_cw], (int64_t)R.cout * 32);
    raw_cb[r] = push_v(blob, tensors[R.t_cb], R.cout);
    L.tb_off = tb;
    tb += R.cout;
    u->rtb.push_back(L);
  }
  u->tb_total = tb;
  for (int i = 0; i < s.n_levels - 1; ++i) {
    const int c = s.dims[i + 1];
    u->down_w[i] = push_wt(blob, tensors[s.t_down[i][0]], c, c, 3, false);
    u->down_b[i] = push_v(blob, tensors[s.t_down[i][1]], c);
    const int cu = s.dims[s.n_levels - 1 - i];
    u->up_w[i] = push_wt(blob, tensors[s.t_up[i][0]], cu, cu, 4, true);
    u->up_b[i] = push_v(blob, tensors[s.t_up[i][1]], cu);
  }
  u->fin_w5 = push_wt(blob, tensors[s.t_final[0]], s.uid, s.uid, 5, false);
  u->fin_b5 = push_v(blob, tensors[s.t_final[1]], s.uid);
  u->fin_g = push_v(blob, tensors[s.t_final[2]], s.uid);
  u->fin_be = push_v(blob, tensors[s.t_final[3]], s.uid);
  u->fin_w1 = push_wt(blob, tensors[s.t_final[4]], 4, s.uid, 1, false);
  u->fin_b1 = push_v(blob, tensors[s.t_final[5]], 4);
  if (hipMalloc(&u->blob, blob.size() * sizeof(float)) != hipSuccess ||
      hipMalloc(&u->ttable, (size_t)T * u->tb_total * sizeof(float)) != hipSuccess) {
    set_error("mmd_unet_create: hipMalloc failed");
    layered_destroy(u);
    return 1;
  }
  MMD_HIP_CHECK(hipMemcpyAsync(u->blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice, st));
  MMD_HIP_CHECK(hipStreamSynchronize(st));
  TimeArgs ta{};
  ta.w1 = u->blob + raw_time[0]; ta.b1 = u->blob + raw_time[1];
  ta.w3 = u->blob + raw_time[2]; ta.b3 = u->blob + raw_time[3];
  for (size_t r = 0; r < s.rtb.size(); ++r) {
    ta.cw[r] = u->blob + raw_cw[r]; ta.cb[r] = u->blob + raw_cb[r];
    ta.cout[r] = s.rtb[r].cout; ta.off[r] = u->rtb[r].tb_off;
  }
  ta.n_rtb = (int)s.rtb.size(); ta.total = u->tb_total; ta.table = u->ttable;
  launch_time_table(ta, T, st);
  MMD_HIP_CHECK(hipGetLastError());
  MMD_HIP_CHECK(hipStreamSynchronize(st));
  *out = u;
  return 0;
}

void layered_destroy(LayeredUnet* u) {
  if (!u) return;
  if (u->blob) (void)hipFree(u->blob);
  if (u->ttable) (void)hipFree(u->ttable);
  delete u;
}

// buffers of one forward, each n x 64 x unet_input_dim floats (L x C is the same at every level): two level inputs, the hidden
// tensor of a block, the output of a level's first block, a 1x1 residual, level 0's output, the (n_levels - 1) skip tensors; the
// concatenated up-path input is read from its two tensors in place
static int n_buffers(const LayeredUnet* u) { return 6 + (u->spec.n_levels - 1); }
size_t layered_workspace_bytes(const LayeredUnet* u, int n_traj) {
  return n_traj > 0 ? (size_t)n_buffers(u) * n_traj * u->per_sample * sizeof(float) + 256 : 0;
}

int layered_forward(const LayeredUnet* u, const float* x, int t, float* eps, int n, void* ws, size_t ws_bytes, hipStream_t st) {
  MMD_REQUIRE(ws_bytes >= layered_workspace_bytes(u, n), "mmd_unet_forward: workspace too small");
  const Spec& s = u->spec;
  const size_t per = (size_t)n * u->per_sample;
  float* base = reinterpret_cast<float*>(ws);
  float* in[2] = {base, base + per};
  float* tmp = base + 2 * per;
  float* mid = base + 3 * per;
  float* resb = base + 4 * per;
  float* out0 = base + 5 * per;
  float* skip[MAX_LEVELS - 1];
  for (int i = 0; i < s.n_levels - 1; ++i) skip[i] = base + (6 + i) * per;   // skip[j] = output of down level j + 1
  const float* B = u->blob;
  const float* tt = u->ttable + (size_t)t * u->tb_total;
  auto conv = [&](int mode, int k, const float* x1, int c1, const float* x2, int c2, int l_in, int l_out, int c_out,
                  size_t w, size_t b, const float* gamma, const float* beta, const float* add_c, const float* add_t, float* y) {
    ConvArgs a{x1, x2, c1, c2, l_in, l_out, c_out, B + w, B + b, gamma, beta, add_c, add_t, y};
    const size_t shm = ((size_t)l_in * (c1 + c2) + (size_t)l_out * c_out + 2 * N_GROUPS) * sizeof(float);
    if (mode == 1) hipLaunchKernelGGL((conv_block_kernel<1, 4, 2>), dim3(n), dim3(256), shm, st, a);
    else if (k == 5) hipLaunchKernelGGL((conv_block_kernel<0, 5, 1>), dim3(n), dim3(256), shm, st, a);
    else if (k == 3) hipLaunchKernelGGL((conv_block_kernel<0, 3, 2>), dim3(n), dim3(256), shm, st, a);
    else hipLaunchKernelGGL((conv_block_kernel<0, 1, 1>), dim3(n), dim3(256), shm, st, a);
  };
  // one ResidualTemporalBlock (layers.py:346-358): (x1 | x2) [L][cin] -> out [L][cout]
  auto rtb = [&](const LRtb& R, const float* x1, int c1, const float* x2, int c2, int L, float* out) {
    conv(0, 5, x1, c1, x2, c2, L, L, R.cout, R.wa, R.ba, B + R.ga, B + R.bea, tt + R.tb_off, nullptr, tmp);
    const float* res = x1;                                   // identity residual (cin == cout: never a concatenated input)
    if (R.res) {
      conv(0, 1, x1, c1, x2, c2, L, L, R.cout, R.wr, R.br, nullptr, nullptr, nullptr, nullptr, resb);
      res = resb;
    }
    conv(0, 5, tmp, R.cout, nullptr, 0, L, L, R.cout, R.wb, R.bb, B + R.gb, B + R.beb, nullptr, res, out);
  };
  const int NL = s.n_levels;
  int L = H, cin = 4;
  const float* xin = x;
  float* level_out = out0;
  for (int i = 0; i < NL; ++i) {                              // downs (temporal_unet.py:147-156)
    const int c = s.dims[i + 1];
    level_out = i == 0 ? out0 : skip[i - 1];
    rtb(u->rtb[2 * i], xin, cin, nullptr, 0, L, mid);
    rtb(u->rtb[2 * i + 1], mid, c, nullptr, 0, L, level_out);
    if (i < NL - 1) {
      conv(0, 3, level_out, c, nullptr, 0, L, L / 2, c, u->down_w[i], u->down_b[i], nullptr, nullptr, nullptr, nullptr, in[i & 1]);
      xin = in[i & 1];
      L /= 2;
    }
    cin = c;
  }
  {                                                          // mid blocks (state_dict order: behind the ups)
    const int m0 = 2 * NL + 2 * (NL - 1), c = s.dims[NL];
    rtb(u->rtb[m0], level_out, c, nullptr, 0, L, mid);
    rtb(u->rtb[m0 + 1], mid, c, nullptr, 0, L, in[0]);
  }
  for (int i = 0; i < NL - 1; ++i) {                          // ups: x = cat(x, h.pop()) (temporal_unet.py:164-171)
    const int din = s.dims[NL - 1 - i], dout = s.dims[NL - i];
    rtb(u->rtb[2 * NL + 2 * i], in[0], dout, skip[NL - 2 - i], dout, L, mid);
    rtb(u->rtb[2 * NL + 2 * i + 1], mid, din, nullptr, 0, L, in[1]);
    conv(1, 4, in[1], din, nullptr, 0, L, 2 * L, din, u->up_w[i], u->up_b[i], nullptr, nullptr, nullptr, nullptr, in[0]);
    L *= 2;
  }
  // final_conv (temporal_unet.py:104-110); with one level there are no ups and L is still 64
  conv(0, 5, in[0], s.uid, nullptr, 0, L, L, s.uid, u->fin_w5, u->fin_b5, B + u->fin_g, B + u->fin_be, nullptr, nullptr, mid);
  conv(0, 1, mid, s.uid, nullptr, 0, L, L, 4, u->fin_w1, u->fin_b1, nullptr, nullptr, nullptr, nullptr, eps);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace mmd
