// TemporalUnet forward, layer by layer, for the configurations the fused kernel (unet.hip) is not instantiated for -- first of all
// UNET_DIM_MULTS[1] = (1, 2, 4, 8) (mmd/models/diffusion_models/temporal_unet.py:17-20, selected by a checkpoint's args.yaml at
// mmd/planners/single_agent/mpd.py:158; the released checkpoints use option 0 and run the fused kernel).  Plain fp32 FMA
// arithmetic on the vector ALUs, one launch per Conv1dBlock / conv, activations channels-FIRST [n][C][L] in an HBM workspace (the
// trajectory input and the eps output are [n][L][4] as everywhere else): a correct path for a rarely used configuration, about
// 50 launches per forward instead of one.
//
//   ResidualTemporalBlock (layers.py:323-358): out = Mish(GN(conv5(Mish(GN(conv5(x))) + time bias))) + res(x)
//   Downsample1d = Conv1d(k3, s2, p1), Upsample1d = ConvTranspose1d(k4, s2, p1) (layers.py:261-279)
//   final_conv = Conv1dBlock(k5) + Conv1d(k1) (temporal_unet.py:104-110)
//
// conv5_block_kernel (the 33 Conv1dBlocks = all but a few percent of the arithmetic): a workgroup owns one sample and a slice of
// whole GroupNorm groups of the output channels (blockIdx.y; small launches are split 2 .. 8 ways so they still fill the chip).  The
// input rows are staged in LDS as [c_in][L + 8] with four zeros either side (no boundary tests in the loop); a thread holds a
// register tile of 4 positions x CT channels: per input channel it reads its 8-position window once (two 16-byte LDS reads) and
// 5 CT weights [tap][c_in][c_out] (adjacent threads = adjacent output channels: coalesced, L2-resident) for 20 CT FMAs.  The conv
// output goes to LDS for the GroupNorm (8 groups, two-pass statistics, eps 1e-5) + Mish + addend, then to HBM.
// conv_plain_kernel: the strided / transposed / 1x1 convs without a norm, one thread per output, split over blockIdx.y likewise.
#include <hip/hip_runtime.h>

#include <algorithm>
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
  const float* x1; const float* x2;   // input [n][C1][L_in] (+ [n][C2][L_in] concatenated behind it along the channels, or NULL)
  int c1, c2, l_in, l_out, c_out;
  int cs;                             // output channels per workgroup (blockIdx.y * cs = first)
  int in_cl, out_cl;                  // the input / output tensor is channels-last [n][L][C] (the trajectory, eps)
  const float* wt;                    // [taps][c1 + c2][c_out]
  const float* bias;                  // [c_out]
  const float* gamma; const float* beta;   // GroupNorm affine (conv5_block_kernel)
  const float* add_c;                 // per-channel addend after Mish (time bias), or NULL
  const float* add_t;                 // [n][c_out][l_out] addend after Mish (residual), or NULL
  float* y;                           // [n][c_out][l_out]
};

__device__ __forceinline__ float load_in(const ConvArgs& a, size_t n, int c, int l) {
  if (a.in_cl) return a.x1[(n * a.l_in + l) * a.c1 + c];
  return c < a.c1 ? a.x1[(n * a.c1 + c) * a.l_in + l] : a.x2[(n * a.c2 + (c - a.c1)) * a.l_in + l];
}

// Conv1dBlock (layers.py:232-258): Conv1d(k5, p2) -> GroupNorm(8) -> Mish, + per-channel addend, + tensor addend.
// blockDim = KS * NT, NT = min(64, the slice's (L / 4) * (cs / CT) register tiles rounded up to 16): always whole waves, at most
// 256 threads; a thread loops over the tiles r, r + NT, ... of its thread group (one trip for the power-of-two channel counts, more
// where a GroupNorm group is 5 or 7 channels wide: unet_input_dim 40 / 56).  The sum over the input channels is DEFINED as KS = 4
// interleaved partial sums (channels ci = r mod 4, taps in order) combined as ((s0 + s1) + (s2 + s3)) + bias: thread group r of a
// workgroup computes s_r, so the bits do not depend on how a launch is sliced (CT, cs follow the batch size).
constexpr int KS = 4;
template <int CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4))) void conv5_block_kernel(ConvArgs a) {
  extern __shared__ float lds[];
  const int cin = a.c1 + a.c2, L = a.l_in, Lp = L + 8, tid = threadIdx.x, nthr = blockDim.x;
  const size_t n = blockIdx.x;
  const int c0 = blockIdx.y * a.cs, tile = a.cs * L;
  float* xs = lds;                               // [cin][Lp]
  float* part = lds + cin * Lp;                  // [KS][cs][L] partial sums; [0] becomes the conv output
  float* red = part + KS * tile;                 // [2 * groups of the slice]
  for (int i = tid; i < cin * Lp; i += nthr) {
    const int c = i / Lp, l = i % Lp - 4;
    xs[i] = (l >= 0 && l < L) ? load_in(a, n, c, l) : 0.f;
  }
  __syncthreads();
  const int lanes = a.cs / CT, items = (L / 4) * lanes, nt = nthr / KS, ks = tid / nt;
  for (int r = tid % nt; r < items; r += nt) {
    const int cl = r % lanes, pg = r / lanes;
    float acc[CT][4];
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[j][p] = 0.f;
    // weight loads through a buffer descriptor: a per-thread byte offset (input channel ks, output channel lane cl) that never
    // changes + a scalar offset per (iteration, tap, channel tile): no vector address arithmetic in the loop
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wt), 0, 5 * cin * a.c_out * 4, 0x00020000);
    const int voff = 4 * (ks * a.c_out + cl * CT);                 // the thread's CT output channels are adjacent: one load per tap
    const int tap = cin * a.c_out;
    auto load_w = [&](float (&w)[5 * CT], int it) {
      const int so = c0 + it * KS * a.c_out;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        if constexpr (CT == 1) {
          w[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, voff, 4 * (so + k * tap), 0));
        } else if constexpr (CT == 2) {
          // (the b64 / b128 buffer-load builtins of this compiler return the first dword in every component: plain vector loads)
          const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.wt + so + k * tap) + voff);
          w[2 * k] = v.x; w[2 * k + 1] = v.y;
        } else {
          const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.wt + so + k * tap) + voff);
          w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
        }
      }
    };
    auto taps = [&](const float (&w)[5 * CT], int ci) {
      const float4* xr = reinterpret_cast<const float4*>(xs + ci * Lp + pg * 4);     // positions 4 pg - 4 .. 4 pg + 7
      const float4 v0 = xr[0], v1 = xr[1], v2 = xr[2];
      const float xw[8] = {v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y};           // positions 4 pg - 2 .. 4 pg + 5
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[j][p] = fmaf(xw[p + k], w[k * CT + j], acc[j][p]);
    };
    // two weight sets in flight alternately: the next input channel's loads are issued before the current one's 20 CT FMAs
    float w0[5 * CT], w1[5 * CT];
    const int nci = cin / KS;                            // this thread's input channels: ks, ks + KS, ... (c_in is 4 or a multiple of 8)
    if (nci > 0) load_w(w0, 0);
    int it = 0;
    for (; it + 2 <= nci; it += 2) {
      load_w(w1, it + 1);
      taps(w0, ks + it * KS);
      load_w(w0, it + 2 < nci ? it + 2 : it);
      taps(w1, ks + (it + 1) * KS);
    }
    if (it < nci) taps(w0, ks + it * KS);
#pragma unroll
    for (int j = 0; j < CT; ++j)
      *reinterpret_cast<float4*>(part + ks * tile + (cl * CT + j) * L + pg * 4) =
          make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  }
  __syncthreads();
  float* outs = part;
  for (int o = tid; o < tile; o += nthr)                // (each element is read and written by the same thread)
    outs[o] = ((part[o] + part[tile + o]) + (part[2 * tile + o] + part[3 * tile + o])) + a.bias[c0 + o / L];
  __syncthreads();
  // GroupNorm over (c_out / 8 channels) x L positions per group = one contiguous block of `outs`; a wave per group
  const int cpg = a.c_out / N_GROUPS, per = cpg * L, lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  for (int g = wave; g < a.cs / cpg; g += nwave) {
    const float* blk = outs + g * per;
    float s = 0.f;
    for (int i = lane; i < per; i += 64) s += blk[i];
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)per;
    float q = 0.f;
    for (int i = lane; i < per; i += 64) {
      const float d = blk[i] - mean;
      q = fmaf(d, d, q);
    }
    for (int off = 32; off; off >>= 1) q += __shfl_xor(q, off);
    if (lane == 0) {
      red[2 * g] = mean;
      red[2 * g + 1] = 1.f / sqrtf(q / (float)per + 1e-5f);
    }
  }
  __syncthreads();
  const size_t base = (n * a.c_out + c0) * L;
  for (int o = tid; o < tile; o += nthr) {
    const int c = o / L, co = c0 + c, g = c / cpg;
    float v = mish_ref((outs[o] - red[2 * g]) * red[2 * g + 1] * a.gamma[co] + a.beta[co]);
    if (a.add_c) v += a.add_c[co];
    if (a.add_t) v += a.add_t[base + o];
    a.y[base + o] = v;
  }
}

// MODE 0: Conv1d, K taps, stride S, padding K / 2.  MODE 1: ConvTranspose1d(k4, s2, p1): out[2 m] = in[m - 1] W3 + in[m] W1,
// out[2 m + 1] = in[m] W2 + in[m + 1] W0 (taps stored in kernel-index order 0 .. 3).  No norm; input staged as [l][c_in].
template <int MODE, int K, int S>
__global__ __launch_bounds__(256) void conv_plain_kernel(ConvArgs a) {
  extern __shared__ float lds[];
  const int cin = a.c1 + a.c2, tid = threadIdx.x;
  const size_t n = blockIdx.x;
  const int c0 = blockIdx.y * a.cs;
  float* xin = lds;                              // [l_in][cin]
  for (int i = tid; i < a.l_in * cin; i += 256) {
    const int c = a.in_cl ? i % cin : i / a.l_in, l = a.in_cl ? i / cin : i % a.l_in;
    xin[l * cin + c] = load_in(a, n, c, l);
  }
  __syncthreads();
  for (int o = tid; o < a.l_out * a.cs; o += 256) {
    // adjacent threads = adjacent output channels (coalesced weights); the store below is strided but small
    const int lo = o / a.cs, co = c0 + o % a.cs;
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
    if (a.out_cl) a.y[(n * a.l_out + lo) * a.c_out + co] = acc;
    else a.y[(n * a.c_out + co) * a.l_out + lo] = acc;
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
  size_t blob_bytes = 0;
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
  // widest Conv1dBlock input: ups.0.0 of a four-level net stages 2 x 8 uid x 8 channels x (8 + 8) positions (64 KB at uid 64)
  for (const void* f : {(const void*)conv5_block_kernel<1>, (const void*)conv5_block_kernel<2>, (const void*)conv5_block_kernel<4>})
    MMD_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  MMD_HIP_CHECK(hipMemcpyAsync(u->blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice, st));
  MMD_HIP_CHECK(hipStreamSynchronize(st));
  u->blob_bytes = blob.size() * sizeof(float);
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
size_t layered_weight_bytes(const LayeredUnet* u) { return u->blob_bytes; }
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
  // split a launch's output channels over blockIdx.y until it has >= 3 workgroups per CU (or 8 slices: one GroupNorm group each)
  auto slices = [&](int c_out, int unit, int max_cs) {
    int ns = 1;
    while (ns < N_GROUPS && ((long long)n * ns < 768 || c_out / ns > max_cs) && (c_out / (2 * ns)) % unit == 0) ns *= 2;
    return ns;
  };
  // Conv1dBlock: (x1 | x2) [c][L] -> Mish(GN(conv5)) + add_c[c] + add_t -> y.  A workgroup of KS x nconv threads, nconv =
  // (L / 4) * (cs / CT) in [16, 64]: the slice width cs (whole GroupNorm groups) follows from that, CT = 2 for the big launches
  auto block5 = [&](const float* x1, int c1, const float* x2, int c2, int in_cl, int L, int c_out, size_t w, size_t b, size_t gamma,
                    size_t beta, const float* add_c, const float* add_t, float* y) -> int {
    const int cpg = c_out / N_GROUPS;
    int ct = n >= 768 ? 4 : n >= 192 ? 2 : 1, cs = c_out;
    auto nconv = [&]() { return (L / 4) * (cs / ct); };
    while (nconv() > 64 && (cs / 2) % cpg == 0) cs /= 2;
    while (ct > 1 && (cs % ct || nconv() < 16)) ct /= 2;
    // threads per partial sum: the slice's register tiles, in whole quarter-waves, at most 64 -- a thread loops over more (a group
    // of 5 or 7 channels, or one group of 512 outputs at unet_input_dim 64, does not cut into 16 .. 64 tiles)
    const int nt = nconv() >= 64 ? 64 : (nconv() + 15) / 16 * 16;
    MMD_REQUIRE((ct == 1 || ct == 2 || ct == 4) && cs % ct == 0 && cs % cpg == 0 && c_out % cs == 0 && (KS * nt) % 64 == 0 && KS * nt <= 256,
                "layered_forward: no launch shape for a Conv1dBlock of %d channels at length %d", c_out, L);
    ConvArgs a{x1, x2, c1, c2, L, L, c_out, cs, in_cl, 0, B + w, B + b, B + gamma, B + beta, add_c, add_t, y};
    const size_t shm = ((size_t)(c1 + c2) * (L + 8) + (size_t)KS * cs * L + 2 * N_GROUPS) * sizeof(float);
    if (ct == 4) hipLaunchKernelGGL(conv5_block_kernel<4>, dim3(n, c_out / cs), dim3(KS * nt), shm, st, a);
    else if (ct == 2) hipLaunchKernelGGL(conv5_block_kernel<2>, dim3(n, c_out / cs), dim3(KS * nt), shm, st, a);
    else hipLaunchKernelGGL(conv5_block_kernel<1>, dim3(n, c_out / cs), dim3(KS * nt), shm, st, a);
    return 0;
  };
  auto plain = [&](int mode, int k, const float* x1, int c1, const float* x2, int c2, int in_cl, int l_in, int l_out, int c_out,
                   int out_cl, size_t w, size_t b, float* y) {
    const int ns = c_out >= 32 ? slices(c_out, 8, 1 << 30) : 1;
    ConvArgs a{x1, x2, c1, c2, l_in, l_out, c_out, c_out / ns, in_cl, out_cl, B + w, B + b, nullptr, nullptr, nullptr, nullptr, y};
    const size_t shm = (size_t)l_in * (c1 + c2) * sizeof(float);
    if (mode == 1) hipLaunchKernelGGL((conv_plain_kernel<1, 4, 2>), dim3(n, ns), dim3(256), shm, st, a);
    else if (k == 3) hipLaunchKernelGGL((conv_plain_kernel<0, 3, 2>), dim3(n, ns), dim3(256), shm, st, a);
    else hipLaunchKernelGGL((conv_plain_kernel<0, 1, 1>), dim3(n, ns), dim3(256), shm, st, a);
  };
  // one ResidualTemporalBlock (layers.py:346-358): (x1 | x2) [cin][L] -> out [cout][L]
  auto rtb = [&](const LRtb& R, const float* x1, int c1, const float* x2, int c2, int in_cl, int L, float* out) -> int {
    if (int rc = block5(x1, c1, x2, c2, in_cl, L, R.cout, R.wa, R.ba, R.ga, R.bea, tt + R.tb_off, nullptr, tmp)) return rc;
    const float* res = x1;                                   // identity residual (cin == cout: never a concatenated input)
    if (R.res) {
      plain(0, 1, x1, c1, x2, c2, in_cl, L, L, R.cout, 0, R.wr, R.br, resb);
      res = resb;
    }
    return block5(tmp, R.cout, nullptr, 0, 0, L, R.cout, R.wb, R.bb, R.gb, R.beb, nullptr, res, out);
  };
  const int NL = s.n_levels;
  int L = H, cin = 4;
  const float* xin = x;
  float* level_out = out0;
  for (int i = 0; i < NL; ++i) {                              // downs (temporal_unet.py:147-156)
    const int c = s.dims[i + 1];
    level_out = i == 0 ? out0 : skip[i - 1];
    if (int rc = rtb(u->rtb[2 * i], xin, cin, nullptr, 0, i == 0, L, mid)) return rc;   // (level 0 reads the trajectory: channels-last)
    if (int rc = rtb(u->rtb[2 * i + 1], mid, c, nullptr, 0, 0, L, level_out)) return rc;
    if (i < NL - 1) {
      plain(0, 3, level_out, c, nullptr, 0, 0, L, L / 2, c, 0, u->down_w[i], u->down_b[i], in[i & 1]);
      xin = in[i & 1];
      L /= 2;
    }
    cin = c;
  }
  {                                                          // mid blocks (state_dict order: behind the ups)
    const int m0 = 2 * NL + 2 * (NL - 1), c = s.dims[NL];
    if (int rc = rtb(u->rtb[m0], level_out, c, nullptr, 0, 0, L, mid)) return rc;
    if (int rc = rtb(u->rtb[m0 + 1], mid, c, nullptr, 0, 0, L, in[0])) return rc;
  }
  for (int i = 0; i < NL - 1; ++i) {                          // ups: x = cat(x, h.pop()) (temporal_unet.py:164-171)
    const int din = s.dims[NL - 1 - i], dout = s.dims[NL - i];
    if (int rc = rtb(u->rtb[2 * NL + 2 * i], in[0], dout, skip[NL - 2 - i], dout, 0, L, mid)) return rc;
    if (int rc = rtb(u->rtb[2 * NL + 2 * i + 1], mid, din, nullptr, 0, 0, L, in[1])) return rc;
    plain(1, 4, in[1], din, nullptr, 0, 0, L, 2 * L, din, 0, u->up_w[i], u->up_b[i], in[0]);
    L *= 2;
  }
  // final_conv (temporal_unet.py:104-110); with one level there are no ups and L is still 64
  if (int rc = block5(in[0], s.uid, nullptr, 0, 0, L, s.uid, u->fin_w5, u->fin_b5, u->fin_g, u->fin_be, nullptr, nullptr, mid)) return rc;
  plain(0, 1, mid, s.uid, nullptr, 0, 0, L, L, 4, 1, u->fin_w1, u->fin_b1, eps);
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace mmd
