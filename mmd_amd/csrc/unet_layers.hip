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

#include <memory>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/mmd_amd.h"
#include "common.h"
#include "f16x2.h"
#include "gn_mish.h"
#include <type_traits>
#include <map>
#include <mutex>
#include <utility>
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
  int in_bl;                          // x1 is a blocked activation tensor [n][c1 / 8][l_in][8] of the matrix-pipe path (mconv_kernel)
};

// element offset of (sample, channel c, position l) in a blocked activation tensor of C channels
__device__ __forceinline__ size_t bl_off(size_t smp, int C, int L, int c, int l) { return ((smp * (C >> 3) + (c >> 3)) * L + l) * 8 + (c & 7); }
__device__ __forceinline__ float load_in(const ConvArgs& a, size_t n, int c, int l) {
  if (a.in_bl) return a.x1[bl_off(n, a.c1, a.l_in, c, l)];
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

// The layer-by-layer path ON THE MATRIX PIPE (round 5): fp32 arithmetic as the two-piece fp16 split of f16x2.h (three
// v_mfma_f32_16x16x32_f16 per product, fp32 accumulate -- the fused kernel's building blocks), for a network whose every layer has GEMM
// columns in whole 16-column n-tiles.  KIND 0: Conv1dBlock (conv k5 + GroupNorm + Mish + addends); 1: Conv1d k1 (the 1x1 residual); 2:
// Conv1d k3 stride 2 (Downsample1d); 3: ConvTranspose1d(k4, s2, p1) (Upsample1d) as a k3 conv with 2 c_out columns (column 2 co + parity:
// out[2 m] = in[m - 1] W3 + in[m] W1, out[2 m + 1] = in[m] W2 + in[m + 1] W0) and an interleaving store.
// ACTIVATIONS between the layers are stored BLOCKED, [n][C / 8][L][8 channels]: the 8 channels of a position are 32 contiguous bytes, so
// the staging below reads whole rows of its LDS layout (two 16-byte loads) and the epilogues write runs of channels (the trajectory in
// and eps out keep the caller's channels-last layout).
// GEMM: M = (sample, output row) -- an ITEM is SPW = 64 / LR consecutive samples (LR = rows of a sample: l_in, or l_in / 2 for the strided
// conv), so M is always 64 rows = four M tiles and a weight fragment is used on all of them, as in the fused kernel; N = a
// slice of cs in {16, 32, 64, 128} columns (blockIdx.y; whole GroupNorm groups for KIND 0); K = taps x input channels, in CHUNKS of kch <=
// 128 channels:
//   * per chunk the samples' rows go to LDS as fp16 pieces in ROW form [piece][channel block b = KCj g + kc][row = (l_in + 4) sample + 2 +
//     position][8 channels] (16 bytes per (block, row): an A fragment is one ds_read_b128; two zero rows either side of a sample are the
//     conv's padding, a tap is a row offset, the stride a row step) under a dynamic power-of-two scale PER SAMPLE AND CHUNK from the exact
//     maximum of the staged values (dyn_scale; the maximum is taken on the registers the loads land in, through LDS atomics -- no second
//     pass over the input).  The accumulators live across the chunks in units of the current chunk's scale: a new chunk multiplies them
//     by the ratio of the two scales (a power of two: exact);
//   * weights: host-packed per layer and chunk in B-fragment order [chunk][n-tile][tap][kc][piece][lane] x 16 bytes with per-column
//     power-of-two scales (pack_rd / rd_col_scales of f16x2.h), streamed from L2 through a register ring of RD steps (RD divides a chunk's
//     steps: 5 taps, 3 taps, or the k1 conv's chunk always padded to four K chunks);
//   * a wave takes NTW n-tiles x MTW M tiles (cs <= 32: the waves that share an n-tile split the M tiles);
//   * KIND 0: the accumulators go -- scaled back per channel and sample, bias added -- to an LDS tile that re-uses the slab; a thread then
//     owns 4 rows x cs / 16 adjacent channels of it in registers: the GroupNorm statistics (two passes: mean, then squared deviations) are
//     sums over those and a FIXED balanced tree over the channel index, then over the rows (lanes, then waves), so the bits do not depend
//     on the slicing; normalise + Mish + addend as the fused kernel (gn_mish.h).  The plain convs store from the accumulators.
// A launch has as many workgroups as the chip holds at once; a workgroup (4 waves) takes the items blockIdx.x, + gridDim.x, ... of its slice
// as a sequence of STAGES (item, chunk), and requests the next stage's rows right behind the current stage's conversion, so that they land
// under its GEMM and tail (the widest slice, whose registers are the weight ring's, requests them at the stage's start).
// A sample's arithmetic does not depend on the batch it sits in, on its place in the workgroup or on the slicing.
constexpr int MCONV_KCH = 128;                     // input channels staged per chunk (at most)
constexpr int MCONV_MAX_CHUNKS = 8;                // <= 1024 input channels
constexpr int OST = 68;                            // row stride of the LDS output tile [channel][64 (sample, position) rows]
struct MPack { size_t w = 0, isc = 0; int n_chunks = 0, kch = 0; unsigned stride = 0; };   // f16x2 packs of a conv in the blob (w = 0: none)
struct MConvArgs {
  ConvArgs c;                         // x1 / x2 / c1 / c2 / l_in / l_out / c_out / cs (columns of the slice) / in_cl / bias / gamma / beta / add_c / add_t / y
  const uint4* wpk;                   // the layer's packs; chunk j starts at wpk + j stride: n-tiles x taps x kcj chunks x 2 pieces x 64 lanes
  const float* isc;                   // [columns] inverse weight scales
  unsigned stride;                    // (every chunk but the last is kch channels wide: one stride)
  int n_chunks, kch, n;
  // KIND 4: the block's second Conv1dBlock (c_out -> c_out, one chunk): its packs and GroupNorm parameters; c.add_c (the time bias)
  // follows the first conv, c.add_t (the residual) the second
  const uint4* wpk2;
  const float* isc2;
  const float* bias2; const float* gamma2; const float* beta2;
};
// -DMCONV_TIMING (tools/dbg/mconv_phases.py builds it): wave 0 of every workgroup adds its clock64() phase times to a table indexed by
// (KIND, log2 l_in - 3, NTW * MTW): staging loads + maxima, conversion, GEMM, exchange, statistics, tail, whole; read and cleared by
// mmd_debug_mconv_clocks
#ifdef MCONV_TIMING
__device__ unsigned long long g_mconv_clk[5][4][9][8];
#define MCONV_T(i) if (tid == 0) { const long long now_ = clock64(); clk_[i] += now_ - last_; last_ = now_; }
#else
#define MCONV_T(i)
#endif
template <int KIND> struct MKind;
template <> struct MKind<0> { static constexpr int K = 5, S = 1, RD = 5, NR = 3; };
template <> struct MKind<1> { static constexpr int K = 1, S = 1, RD = 4, NR = 3; };
template <> struct MKind<2> { static constexpr int K = 3, S = 2, RD = 3, NR = 5; };
template <> struct MKind<3> { static constexpr int K = 3, S = 1, RD = 3, NR = 3; };
template <> struct MKind<4> { static constexpr int K = 5, S = 1, RD = 5, NR = 3; };   // a whole ResidualTemporalBlock: two KIND 0 convs
// V adjacent floats (V = 1, 2, 4 or 8) of a blocked tensor from / to global memory (a run inside one 32-byte block: 4 V-byte aligned)
template <int V> __device__ __forceinline__ void ld_run(float (&d)[V], const float* p) {
  if constexpr (V == 1) d[0] = p[0];
  else if constexpr (V == 2) { const float2 q = *reinterpret_cast<const float2*>(p); d[0] = q.x; d[1] = q.y; }
  else {
#pragma unroll
    for (int j = 0; j < V; j += 4) { const float4 q = *reinterpret_cast<const float4*>(p + j); d[j] = q.x; d[j + 1] = q.y; d[j + 2] = q.z; d[j + 3] = q.w; }
  }
}
template <int V> __device__ __forceinline__ void st_run(float* p, const float (&d)[V]) {
  if constexpr (V == 1) p[0] = d[0];
  else if constexpr (V == 2) *reinterpret_cast<float2*>(p) = make_float2(d[0], d[1]);
  else {
#pragma unroll
    for (int j = 0; j < V; j += 4) *reinterpret_cast<float4*>(p + j) = make_float4(d[j], d[j + 1], d[j + 2], d[j + 3]);
  }
}
// NTW n-tiles x MTW M tiles per wave: (2, 4) for a slice of 128 columns, (1, 4) for 64, (1, 2) for 32, (1, 1) for 16 -- compile-time, so
// that the weight ring's slots are registers with exact s_waitcnt counts (with run-time tile counts the compiler drained every load)
// NBH: channel blocks per staging thread (1: chunks of <= 64 channels, three workgroups per CU; 2: up to 128)
template <int NTW, int MTW, int KIND, int NBH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NTW == 1 && NBH == 1 && KIND != 4 ? 3 : 2))) void mconv_kernel(MConvArgs m) {
  constexpr int K = MKind<KIND>::K, S = MKind<KIND>::S, RD = MKind<KIND>::RD, NR = MKind<KIND>::NR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ConvArgs& a = m.c;
  const int cin = a.c1 + a.c2, L = a.l_in, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lsh_in = 31 - __builtin_clz(L);                // l_in is 8, 16, 32 or 64
  const int lsh = lsh_in - (S - 1), LR = 1 << lsh;         // GEMM rows of a sample
  const int SPW = 64 >> lsh, RS = L + 4, SROWS = SPW * RS;  // samples per workgroup, slab rows per sample / in all
  const int c0 = blockIdx.y * a.cs, n_items = (m.n + SPW - 1) / SPW;   // an item = SPW consecutive samples; the workgroup takes items blockIdx.x, + gridDim.x, ...
#ifdef MCONV_TIMING
  long long clk_[6] = {}, last_ = clock64();
  const long long first_ = last_;
#endif
  char* const slab = smem;
  // LDS: [slab: 2 pieces x (kch / 8) blocks x SROWS x 16 B | KIND 0: the output tile [cs][OST] aliases it] [red: 4 waves x 16 partial sums]
  // [smax: 2 x 8 per-sample maxima (as uint: non-negative floats order like their bits), alternating between the stages; 8 more for the
  // hidden tensor of KIND 4]
  const int slab_ch = KIND == 4 && a.c_out > m.kch ? a.c_out : m.kch;   // (KIND 4: the hidden tensor's c_out channels are a slab too)
  const int slab_bytes = 2 * (slab_ch / 8) * SROWS * 16, outs_bytes = KIND == 0 || KIND == 4 ? a.cs * OST * 4 : 0;
  float* const red = reinterpret_cast<float*>(smem + (slab_bytes > outs_bytes ? slab_bytes : outs_bytes));
  unsigned* const smax = reinterpret_cast<unsigned*>(red + 64);
  // (the per-sample maxima, the `red` partials and the slab bookkeeping hold at most 8 samples per item: a sample has >= 8 GEMM rows --
  // H = 64 over at most MAX_LEVELS = 4 levels, the stride-2 conv never runs at the last level -- which mconv() on the host checks
  // (`rows >= 8`); a launch that breaks it must not scribble past smax)
  if (SPW > 8) __builtin_trap();
  if (tid < 24) smax[tid] = 0u;
  // ---- this thread's staging rows of an item: rows tid % 32 + 32 i of the slab -> (sample, position); a padding row, a row past the slab
  // or past the batch reads the first element of the batch and is not used
  int st_smp[NR], st_l[NR], st_sm[NR];
  bool st_row[NR], st_ok[NR];
  auto set_rows = [&](int s0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = (tid & 31) + 32 * i, sm = row / RS, l = row - sm * RS - 2, smp = s0 + sm;
      st_row[i] = row < SROWS;
      st_ok[i] = st_row[i] && smp < m.n && l >= 0 && l < L;
      st_sm[i] = st_row[i] ? sm : 0;
      st_smp[i] = st_ok[i] ? smp : 0;
      st_l[i] = st_ok[i] ? l : 0;
    }
  };
  // the raw rows of (item, chunk): item = (channel block, slab row): the 8 channels of one position (32 contiguous bytes of a blocked tensor);
  // a thread has <= 2 blocks x NR rows.  They are REQUESTED one stage ahead -- behind the previous stage's conversion, so that their
  // latency passes under its GEMM and tail (the weight ring's loads of a stage are issued before, and loads return in order)
  float4 v[NBH][NR][2];
  auto chunk_blocks = [&](int ch) {                       // channel blocks of chunk ch (the k1 conv's chunk is always four K chunks wide)
    const int cc = K == 1 ? m.kch : min(m.kch, cin - ch * m.kch);
    return 4 * ((cc + 31) / 32);
  };
  auto request_rows = [&](int ch) {
    const int c_lo = ch * m.kch, NB = chunk_blocks(ch);
#pragma unroll
    for (int h = 0; h < NBH; ++h) {
      const int blk = (tid >> 5) + 8 * h, c = c_lo + 8 * blk;   // (c1 is a multiple of 8: the block lies in one source)
      if (blk >= NB) continue;
      const bool live = a.in_cl ? c == 0 : c < cin, first = c < a.c1;
      const float* const src = a.in_cl ? a.x1 : first ? a.x1 + (size_t)c * L : a.x2 + (size_t)(c - a.c1) * L;
      const int cw = first ? a.c1 : a.c2;                  // channels of the source: a sample is cw x L elements
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        // blocked: ((smp cw / 8 + c / 8) L + l) 8; the trajectory [n][L][4]: channels 0 .. 3 are one 16-byte load
        const float* const q = !live ? a.x1 : a.in_cl ? src + ((size_t)st_smp[i] * L + st_l[i]) * 4 : src + (size_t)st_smp[i] * cw * L + st_l[i] * 8;
        v[h][i][0] = live ? *reinterpret_cast<const float4*>(q) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[h][i][1] = live && !a.in_cl ? *reinterpret_cast<const float4*>(q + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  // ---- GEMM over the channel chunks
  constexpr int WPT = 4 / MTW;                             // waves that share an n-tile (MTW < 4: they split its M tiles)
  const int nt0 = (wave / WPT) * NTW;                      // first n-tile of the wave
  const int mt0 = wave % WPT;                              // its M tiles: mt0, mt0 + WPT, ...
  const int row = lane & 15, g = lane >> 4;
  int arow[MTW];                                           // slab row of the lane's A row in its M tile i at tap 0: position S lo - K / 2
  int csm[MTW];                                            // sample (of the workgroup) of the lane's four C/D rows 4 g .. 4 g + 3 of M tile i
  float inv_prev[MTW];                                     // 1 / (the current chunk's scale) of that sample
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int r64 = 16 * (mt0 + i * WPT) + row;            // (sample, output row) of the 64
    arow[i] = (r64 >> lsh) * RS + S * (r64 & (LR - 1)) + 2 - K / 2;
    csm[i] = (16 * (mt0 + i * WPT) + 4 * g) >> lsh;
    inv_prev[i] = 1.f;
  }
  f32x4 acc[NTW][MTW];
  // the weight ring: B fragments of step st of the wave's n-tiles nt0 + t (wp: the wave's first fragment, tstride: one n-tile further)
  u32x4 b[RD][NTW][2];
  auto load_b = [&](const u32x4* wp, size_t tstride, int slot, int st) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      b[slot][t][0] = wp[t * tstride + (size_t)st * 128];
      b[slot][t][1] = wp[t * tstride + (size_t)st * 128 + 64];
    }
  };
  auto ring_start = [&](const u32x4* wp, size_t tstride) {
#pragma unroll
    for (int j = 0; j < RD; ++j) load_b(wp, tstride, j, j);
  };
  // the GEMM of one chunk: `steps` = K x KCj steps (tap, kc), tap-major, over the slab's KCj x 4 channel blocks (PS: bytes of a piece)
  auto gemm = [&](const u32x4* wp, size_t tstride, int steps, int KCj, int PS) {
    auto group = [&](int base, auto refill) {              // RD steps: slot j holds step base + j and is refilled with step base + j + RD
#pragma unroll
      for (int j = 0; j < RD; ++j) {
        const int st = base + j, tap = st / KCj, kc = st - tap * KCj;
        const char* const ablk = slab + ((size_t)(KCj * g + kc) * SROWS + tap) * 16;
        u32x4 af[MTW][2];
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          af[i][0] = *reinterpret_cast<const u32x4*>(ablk + arow[i] * 16);
          af[i][1] = *reinterpret_cast<const u32x4*>(ablk + PS + arow[i] * 16);
        }
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int t = 0; t < NTW; ++t) vb_three<false>(acc[t][i], af[i], b[j][t]);
        if (decltype(refill)::value) load_b(wp, tstride, j, st + RD);
      }
    };
    // (the last group peeled: every load of the loop is unconditional, so the waits count loads instead of draining them)
    for (int base = 0; base < steps - RD; base += RD) group(base, std::true_type{});
    group(steps - RD, std::false_type{});
  };
  // (the widest slice keeps no rows in flight across its GEMM: the registers are the weight ring's -- its stages are GEMM-bound)
  constexpr bool AHEAD = NTW == 1 && KIND != 4;          // (nor does the two-conv block: its rows would be held through both tails)
  int item = blockIdx.x, ch = 0, s0 = item * SPW;
  set_rows(s0);
  if (AHEAD) request_rows(0);
  __syncthreads();                                         // smax is zero
  for (unsigned seq = 0;; ++seq) {                         // the stages (item, chunk) of this workgroup
    const int NB = chunk_blocks(ch), KCj = NB / 4;
    const int PS = NB * SROWS * 16;
    unsigned* const mxs = smax + 8 * (seq & 1);
    if (seq) __syncthreads();                              // every wave is done reading the previous stage's slab / output tile
    if (ch == 0) {
      if (KIND == 4 && tid < 8) smax[16 + tid] = 0u;       // (the maxima of the block's hidden tensor: the previous item is done with them)
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        inv_prev[i] = 1.f;
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // the weight ring's first steps are requested before the staging: their latency passes under it
    const int steps = K * KCj;                             // (tap, kc), tap-major: a multiple of the ring's RD slots
    // the wave's B fragments of step st: n-tile nt0 + t, pieces q
    const u32x4* const wp = reinterpret_cast<const u32x4*>(m.wpk) + (size_t)ch * m.stride + (size_t)(c0 / 16 + nt0) * steps * 128 + lane;
    const size_t tstride = (size_t)steps * 128;            // one n-tile further
    ring_start(wp, tstride);
    if (!AHEAD) request_rows(ch);
    {
      // the samples' maxima over the chunk
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        float mx = 0.f;
#pragma unroll
        for (int h = 0; h < NBH; ++h) {
          if ((tid >> 5) + 8 * h >= NB) continue;
          const float4 p = v[h][i][0], q = v[h][i][1];
          mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fmaxf(fabsf(p.z), fabsf(p.w))),
                               fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w)))));
        }
        if (st_ok[i] && mx > 0.f) atomicMax(mxs + st_sm[i], __float_as_uint(mx));
      }
      __syncthreads();
      MCONV_T(0)
      if (tid < 8) smax[8 * ((seq + 1) & 1) + tid] = 0u;   // (the other stage's maxima: every thread has used them by now)
#pragma unroll
      for (int h = 0; h < NBH; ++h) {
        const int blk = (tid >> 5) + 8 * h;
        if (blk >= NB) continue;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          if (!st_row[i]) continue;
          const float sc = st_ok[i] ? dyn_scale(__uint_as_float(mxs[st_sm[i]])).s : 0.f;
          const float4 p = st_ok[i] ? v[h][i][0] : make_float4(0.f, 0.f, 0.f, 0.f), q = st_ok[i] ? v[h][i][1] : make_float4(0.f, 0.f, 0.f, 0.f);
          const F16Pair p0 = f16_split2(p.x * sc, p.y * sc), p1 = f16_split2(p.z * sc, p.w * sc), p2 = f16_split2(q.x * sc, q.y * sc),
                        p3 = f16_split2(q.z * sc, q.w * sc);
          const size_t o = ((size_t)blk * SROWS + (tid & 31) + 32 * i) * 16;
          *reinterpret_cast<uint4*>(slab + o) = make_uint4(p0.hi, p1.hi, p2.hi, p3.hi);
          *reinterpret_cast<uint4*>(slab + PS + o) = make_uint4(p0.lo, p1.lo, p2.lo, p3.lo);
        }
      }
    }
    // the accumulators pass into the new chunk's units
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const DynScale ds = dyn_scale(__uint_as_float(mxs[csm[i]]));
      if (ch) {
        const float r = ds.s * inv_prev[i];
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t][i] *= r;
      }
      inv_prev[i] = ds.inv;
    }
    // the next stage's rows are requested now (v is free), to land under this stage's GEMM and tail
    const bool last_chunk = ch + 1 == m.n_chunks;
    const int n_item = last_chunk ? item + (int)gridDim.x : item, n_ch = last_chunk ? 0 : ch + 1;
    const bool more = n_item < n_items;
    if (more) {
      if (last_chunk) set_rows(n_item * SPW);
      if (AHEAD) request_rows(n_ch);
    }
    __syncthreads();
    MCONV_T(1)
    gemm(wp, tstride, steps, KCj, PS);
    MCONV_T(2)
    if (last_chunk) {
      if (KIND != 0 && KIND != 4) {
        // ---- plain convs: column scale x sample scale, + bias, stored from the accumulators (C/D layout: lane = column lane & 15, rows 4 g ..
        // 4 g + 3 of the M tile = four consecutive output rows of one sample) into the blocked output: the 16 lanes of a row write 64
        // contiguous bytes (two channel blocks; the transposed conv: one block at the positions 2 m and 2 m + 1)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const int col = c0 + 16 * (nt0 + t) + row, co = KIND == 3 ? col >> 1 : col;
          const float kc_ = m.isc[col], bs = a.bias[co];
#pragma unroll
          for (int i = 0; i < MTW; ++i) {
            const int r64 = 16 * (mt0 + i * WPT) + 4 * g, lo = r64 & (LR - 1), smp = s0 + csm[i];
            const float k = kc_ * inv_prev[i];
            if (smp >= m.n) continue;
            float* const dst = a.y + bl_off(smp, a.c_out, a.l_out, co, KIND == 3 ? 2 * lo + (col & 1) : lo);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(KIND == 3 ? 16 : 8) * e] = fmaf(acc[t][i][e], k, bs);
          }
        }
        MCONV_T(5)
      } else {
        // ---- Conv1dBlock tail.  A thread owns rows 4 rq .. 4 rq + 3 (one sample: rq = tid % 16) x the NIT adjacent channels cl NIT .. of the
        // slice (cl = tid / 16)
        float* const outs = reinterpret_cast<float*>(smem);      // [cs][OST] conv output (64 rows + 4 of padding: conflict-free column writes)
        constexpr int NIT = NTW * MTW;                           // = cs / 16
        const int rq = tid & 15, cl = tid >> 4, r0 = 4 * rq, tsm = r0 >> lsh, tl = r0 & (L - 1), tsmp = s0 + tsm, tc = c0 + cl * NIT;
        const int cpg = a.c_out / N_GROUPS, clu = cpg / NIT;     // channel lanes (16 apart) per group: 2, 4, 8 or 16
        // the sum of a (sample, group) -- cpg channels x L positions -- is DEFINED as: per channel the four rows of a quad ((x0 + x1) + (x2 +
        // x3)); a balanced tree over the channel index (within the thread, then lanes 16 and 32 apart, then waves); then a balanced tree
        // over the sample's row quads (lanes 1, 2, 4, 8 apart) -- whatever the slice width
        auto group_sum = [&](float (&s)[NIT]) -> float {
#pragma unroll
          for (int w = 1; w < NIT; w *= 2)
#pragma unroll
            for (int k = 0; k < NIT; k += 2 * w) s[k] += s[k + w];
          float v = s[0];
          v += __shfl_xor(v, 16);
          if (clu >= 4) v += __shfl_xor(v, 32);
          if (clu >= 8) {                                        // the group spans 2 or 4 waves
            if (lane < 16) red[16 * wave + lane] = v;
            __syncthreads();
            const int w0 = clu >= 16 ? 0 : wave & 2;
            v = red[16 * w0 + rq] + red[16 * (w0 + 1) + rq];
            if (clu >= 16) v = v + (red[32 + rq] + red[48 + rq]);
            __syncthreads();
          }
          v += __shfl_xor(v, 1);
          if (L >= 16) v += __shfl_xor(v, 2);
          if (L >= 32) v += __shfl_xor(v, 4);
          if (L >= 64) v += __shfl_xor(v, 8);
          return v;
        };
        // accumulators (in units of 1 / inv[i]) -> conv output + bias -> GroupNorm -> Mish -> + addend (per channel, or per element from a
        // blocked tensor) -> o[row][channel] of the thread's 4 x NIT patch.  The global operands are requested first, so that their latency
        // passes under the exchange
        auto block_tail = [&](const float* bias, const float* gamma, const float* beta, const float* isc, const float* add_c,
                              const float* add_t, const float (&inv)[MTW], float (&o)[4][NIT]) {
          float gm[NIT], bt[NIT], ad[4][NIT];                    // gamma, beta; the addend per (row, channel)
#pragma unroll
          for (int k = 0; k < NIT; ++k) {
            gm[k] = gamma[tc + k];
            bt[k] = beta[tc + k];
          }
          if (add_c) {
#pragma unroll
            for (int k = 0; k < NIT; ++k) ad[0][k] = ad[1][k] = ad[2][k] = ad[3][k] = add_c[tc + k];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (add_t && tsmp < m.n) {
                ld_run<NIT>(ad[e], add_t + bl_off(tsmp, a.c_out, L, tc, tl + e));
              } else {
#pragma unroll
                for (int k = 0; k < NIT; ++k) ad[e][k] = 0.f;
              }
            }
          }
          float kc_[NTW], bs_[NTW];
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            kc_[t] = isc[c0 + 16 * (nt0 + t) + row];
            bs_[t] = bias[c0 + 16 * (nt0 + t) + row];
          }
          __syncthreads();                                       // the slab is dead: its memory becomes the output tile
          // accumulators -> outs[c][r64] (C/D layout: lane = column lane & 15, rows 4 g .. 4 g + 3 of the M tile), scaled back, + bias
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            const int ocl = 16 * (nt0 + t) + row;
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
              const int r64 = 16 * (mt0 + i * WPT) + 4 * g;
              const float k = kc_[t] * inv[i], bs = bs_[t];
              *reinterpret_cast<float4*>(outs + ocl * OST + r64) =
                  make_float4(fmaf(acc[t][i][0], k, bs), fmaf(acc[t][i][1], k, bs), fmaf(acc[t][i][2], k, bs), fmaf(acc[t][i][3], k, bs));
            }
          }
          __syncthreads();
          MCONV_T(3)
          float4 x[NIT];
#pragma unroll
          for (int k = 0; k < NIT; ++k) x[k] = *reinterpret_cast<const float4*>(outs + (cl * NIT + k) * OST + r0);
          const float per = (float)(cpg << lsh);
          float s[NIT];
#pragma unroll
          for (int k = 0; k < NIT; ++k) s[k] = (x[k].x + x[k].y) + (x[k].z + x[k].w);
          const float mean = group_sum(s) / per;                 // two passes, as the reference: the mean, then the squared deviations
#pragma unroll
          for (int k = 0; k < NIT; ++k) {
            const float d0 = x[k].x - mean, d1 = x[k].y - mean, d2 = x[k].z - mean, d3 = x[k].w - mean;
            s[k] = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          }
          const float rstd = 1.f / sqrtf(group_sum(s) / per + 1e-5f);
          MCONV_T(4)
          // normalise + Mish + addend (the fused kernel's arithmetic, gn_mish.h)
#pragma unroll
          for (int k = 0; k < NIT; ++k) {
            const GnCoef cf = gn_coef(mean, rstd, gm[k], bt[k]);
            const f32x2_t lo = gn_mish2(f32x2_t{x[k].x, x[k].y}, cf, f32x2_t{ad[0][k], ad[1][k]});
            const f32x2_t hi = gn_mish2(f32x2_t{x[k].z, x[k].w}, cf, f32x2_t{ad[2][k], ad[3][k]});
            o[0][k] = lo.x; o[1][k] = lo.y; o[2][k] = hi.x; o[3][k] = hi.y;
          }
        };
        float o[4][NIT];
        if (KIND == 0) {
          block_tail(a.bias, a.gamma, a.beta, m.isc, a.add_c, a.add_t, inv_prev, o);
        } else {
          // ---- KIND 4, a whole ResidualTemporalBlock (cs = c_out: one slice).  The first Conv1dBlock's output + time bias stays in the
          // workgroup: it becomes the second conv's A slab (fp16 pieces under its own per-sample scale: the maximum over the thread patches,
          // LDS atomics as in the staging) -- the arithmetic of the two-launch form, without the tensor's trip through memory
          block_tail(a.bias, a.gamma, a.beta, m.isc, a.add_c, nullptr, inv_prev, o);
          const int KC2 = a.c_out / 32, NB2 = 4 * KC2, PS2 = NB2 * SROWS * 16, steps2 = K * KC2;
          const u32x4* const wp2 = reinterpret_cast<const u32x4*>(m.wpk2) + (size_t)nt0 * steps2 * 128 + lane;
          if (NTW == 1) ring_start(wp2, (size_t)steps2 * 128);   // (the widest slice: behind the conversion, its patch is 32 registers)
          unsigned* const mx2 = smax + 16;
          float mx = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
              if (tsmp >= m.n) o[e][k] = 0.f;                    // (a sample past the batch: zeros, as the staging gives)
              mx = fmaxf(mx, fabsf(o[e][k]));
            }
          if (mx > 0.f) atomicMax(mx2 + tsm, __float_as_uint(mx));
          __syncthreads();                                       // the maxima are complete, and every thread has read its patch of the tile
          {
            const float sc = dyn_scale(__uint_as_float(mx2[tsm])).s;
            // the thread's NIT channels of row e: 2 NIT bytes at channel offset tc % 8 of block tc / 8 (tc = cl NIT: c0 = 0)
            char* const dst = slab + ((size_t)(tc >> 3) * SROWS + tsm * RS + 2 + tl) * 16 + (tc & 7) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              unsigned hi[NIT / 2 > 0 ? NIT / 2 : 1], lo[NIT / 2 > 0 ? NIT / 2 : 1];
#pragma unroll
              for (int k = 0; k + 1 < NIT; k += 2) {
                const F16Pair pr = f16_split2(o[e][k] * sc, o[e][k + 1] * sc);
                hi[k / 2] = pr.hi;
                lo[k / 2] = pr.lo;
              }
              if constexpr (NIT == 8) {
                *reinterpret_cast<uint4*>(dst + e * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                *reinterpret_cast<uint4*>(dst + PS2 + e * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
              } else if constexpr (NIT == 4) {
                *reinterpret_cast<uint2*>(dst + e * 16) = make_uint2(hi[0], hi[1]);
                *reinterpret_cast<uint2*>(dst + PS2 + e * 16) = make_uint2(lo[0], lo[1]);
              } else {
                *reinterpret_cast<unsigned*>(dst + e * 16) = hi[0];
                *reinterpret_cast<unsigned*>(dst + PS2 + e * 16) = lo[0];
              }
            }
            // the padding rows (two either side of a sample) of every block, both pieces
            for (int i = tid; i < SPW * 4 * NB2 * 2; i += 256) {
              const int h = i & 3, sm = (i >> 2) % SPW, bq = (i >> 2) / SPW;        // bq = piece * NB2 + block
              *reinterpret_cast<uint4*>(slab + ((size_t)bq * SROWS + sm * RS + (h < 2 ? h : L + h)) * 16) = make_uint4(0u, 0u, 0u, 0u);
            }
          }
          if (NTW != 1) ring_start(wp2, (size_t)steps2 * 128);
          float inv2[MTW];
#pragma unroll
          for (int i = 0; i < MTW; ++i) {
            inv2[i] = dyn_scale(__uint_as_float(mx2[csm[i]])).inv;
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          __syncthreads();                                       // the second conv's slab is complete
          gemm(wp2, (size_t)steps2 * 128, steps2, KC2, PS2);
          block_tail(m.bias2, m.gamma2, m.beta2, m.isc2, nullptr, a.add_t, inv2, o);
        }
        // -> y, blocked: a row's NIT channels are one run
        if (tsmp < m.n) {
#pragma unroll
          for (int e = 0; e < 4; ++e) st_run<NIT>(a.y + bl_off(tsmp, a.c_out, L, tc, tl + e), o[e]);
        }
        MCONV_T(5)
      }
    }
    if (!more) break;
    item = n_item;
    ch = n_ch;
    s0 = item * SPW;
  }
#ifdef MCONV_TIMING
  if (tid == 0) {
    unsigned long long* t = g_mconv_clk[KIND][lsh_in - 3][NTW * MTW];
    for (int i = 0; i < 6; ++i) atomicAdd(t + i, (unsigned long long)clk_[i]);
    atomicAdd(t + 6, (unsigned long long)(clock64() - first_));
    atomicAdd(t + 7, (unsigned long long)((n_items - 1 - (int)blockIdx.x) / (int)gridDim.x + 1));
  }
#endif
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
// smallest slice of a Conv1dBlock on the matrix pipe: 16, 32 or 64 output channels = whole 16-column n-tiles AND whole GroupNorm groups
// (0: none fits -- unet_input_dim 8's first level, the 24 / 40 / 56 ladders: those blocks stay on conv5_block_kernel)
int mfma_slice(int c_out) {
  if (c_out % 16) return 0;
  const int cpg = c_out / N_GROUPS;
  for (int cs : {16, 32, 64})
    if (c_out % cs == 0 && cs % cpg == 0) return cs;
  return 0;
}
// f16x2 packs of a conv given as wk [cols][cin][K] (K taps in slab-row order), one per chunk of kch input channels: kch = the input
// channels rounded up to whole K chunks of 32, at most MCONV_KCH (the k1 conv: always MCONV_KCH, its four steps are the weight ring);
// the last chunk zero-padded
MPack push_mfma(std::vector<float>& blob, const float* wk, int cols, int cin, int K) {
  MPack p{};
  const int kch = K == 1 ? MCONV_KCH : std::min(MCONV_KCH, (cin + 31) / 32 * 32);
  const int n_chunks = (cin + kch - 1) / kch;
  if (cols % 16 || n_chunks > MCONV_MAX_CHUNKS) return p;
  const int cp = K == 1 ? n_chunks * kch : (cin + 31) / 32 * 32;
  std::vector<float> wp((size_t)cols * cp * K, 0.f);
  for (int co = 0; co < cols; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int k = 0; k < K; ++k) wp[((size_t)co * cp + ci) * K + k] = wk[((size_t)co * cin + ci) * K + k];
  std::vector<int> taps(K);
  for (int k = 0; k < K; ++k) taps[k] = k;
  const std::vector<float> sc = rd_col_scales(wp.data(), cols, cp, K, taps, false);
  p.isc = push_inverse(blob, sc);
  p.n_chunks = n_chunks;
  p.kch = kch;
  for (int j = 0; j < n_chunks; ++j) {
    const int c_lo = j * kch, cc = std::min(kch, cp - c_lo);
    const size_t o = pack_rd(blob, wp.data(), cols, cp, c_lo, cc, K, taps, false, false, sc);
    if (j == 0) p.w = o;
    if (j == 1) p.stride = (unsigned)((o - p.w) / 4);      // in 16-byte units; every chunk but the last has the same size
    if (j > 1 && (o - p.w) / 4 != (size_t)j * p.stride) return MPack{};
  }
  return p;
}
// Conv1dBlock: the slice must be whole GroupNorm groups
MPack push_mfma5(std::vector<float>& blob, const float* w, int cout, int cin) { return mfma_slice(cout) ? push_mfma(blob, w, cout, cin, 5) : MPack{}; }
// ConvTranspose1d(k4, s2, p1) [cin][cout][4] as a k3 conv with columns 2 co + parity over rows (m - 1, m, m + 1):
// out[2 m] = in[m - 1] W3 + in[m] W1, out[2 m + 1] = in[m] W2 + in[m + 1] W0
MPack push_mfma_up(std::vector<float>& blob, const float* w, int cout, int cin) {
  std::vector<float> wk((size_t)2 * cout * cin * 3, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci) {
      const float* k = w + ((size_t)ci * cout + co) * 4;
      float* e = wk.data() + ((size_t)(2 * co) * cin + ci) * 3;
      float* o = wk.data() + ((size_t)(2 * co + 1) * cin + ci) * 3;
      e[0] = k[3]; e[1] = k[1];
      o[1] = k[2]; o[2] = k[0];
    }
  return push_mfma(blob, wk.data(), 2 * cout, cin, 3);
}
template <int KIND, int NBH>
const void* mconv_fn(int cs) {
  return cs == 128 ? (const void*)mconv_kernel<2, 4, KIND, NBH> : cs == 64 ? (const void*)mconv_kernel<1, 4, KIND, NBH>
       : cs == 32 ? (const void*)mconv_kernel<1, 2, KIND, NBH> : (const void*)mconv_kernel<1, 1, KIND, NBH>;
}
template <int KIND>
const void* mconv_fn(int cs, int kch) { return kch <= 64 ? mconv_fn<KIND, 1>(cs) : mconv_fn<KIND, 2>(cs); }
constexpr int kNumCUs = 256;                       // MI355X
// workgroups of `fn` with `shm` bytes of LDS that one CU holds (cached: the launches of a forward ask 47 times)
int mconv_resident(const void* fn, size_t shm) {
  static std::mutex mu;
  static std::map<std::pair<const void*, size_t>, int> cache;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find({fn, shm});
  if (it != cache.end()) return it->second;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, shm) != hipSuccess || nb < 1) nb = 1;
  cache[{fn, shm}] = nb;
  return nb;
}
template <int KIND>
int mconv_set_lds() {
  for (int cs : {16, 32, 64, 128})
    for (int kch : {64, 128}) MMD_HIP_CHECK(hipFuncSetAttribute(mconv_fn<KIND>(cs, kch), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  return 0;
}
size_t push_v(std::vector<float>& blob, const float* p, int64_t n) {
  const size_t off = blob.size();
  blob.insert(blob.end(), p, p + n);
  return off;
}

struct LRtb { int cin, cout; size_t wa, ba, ga, bea, wb, bb, gb, beb, wr, br; bool res; int tb_off; MPack ma, mb, mr; };

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
  MPack fin_m, down_m[MAX_LEVELS - 1], up_m[MAX_LEVELS - 1];
  int rtb_fused = 64;                 // mmd_unet_options.rtb_fused (A/B): the widest ResidualTemporalBlock that runs as ONE launch
                                      // (0: none -- two Conv1dBlock launches each; default 64: with 128 channels the one-launch form spills)
  int mconv_max_cs = 128;             // mmd_unet_options.mconv_max_cs (A/B): the widest column slice mconv_kernel is launched with
  bool mfma = false;                  // every layer has f16x2 packs: the matrix-pipe kernels with blocked activations; else the vector-ALU kernels
  int per_sample = 0;                 // floats of the largest activation tensor of a sample (64 x unet_input_dim)
};

int layered_create(LayeredUnet** out, const Spec& s, int T, const float* const* tensors, const mmd_unet_options* opt, hipStream_t st) {
  // (owned until the last step succeeded: no error path below leaks the device allocations or the host object)
  std::unique_ptr<LayeredUnet, void (*)(LayeredUnet*)> owner(new LayeredUnet(), layered_destroy);
  LayeredUnet* u = owner.get();
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
    if (R.res) {
      L.wr = push_wt(blob, tensors[R.t_rw], R.cout, R.cin, 1, false); L.br = push_v(blob, tensors[R.t_rb], R.cout);
      L.mr = push_mfma(blob, tensors[R.t_rw], R.cout, R.cin, 1);
    }
    L.ma = push_mfma5(blob, tensors[R.t_w0], R.cout, R.cin);
    L.mb = push_mfma5(blob, tensors[R.t_w1], R.cout, R.cout);
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
    u->down_m[i] = push_mfma(blob, tensors[s.t_down[i][0]], c, c, 3);
    const int cu = s.dims[s.n_levels - 1 - i];
    u->up_w[i] = push_wt(blob, tensors[s.t_up[i][0]], cu, cu, 4, true);
    u->up_b[i] = push_v(blob, tensors[s.t_up[i][1]], cu);
    u->up_m[i] = push_mfma_up(blob, tensors[s.t_up[i][0]], cu, cu);
  }
  u->fin_m = push_mfma5(blob, tensors[s.t_final[0]], s.uid, s.uid);
  if (opt && opt->rtb_fused >= 0) u->rtb_fused = opt->rtb_fused;
  if (opt && opt->mconv_max_cs >= 16) u->mconv_max_cs = opt->mconv_max_cs;
  const bool valu_only = opt && (opt->flags & MMD_UNET_LAYERED_VALU);     // every layer on the vector-ALU kernels (A/B of mconv_kernel)
  u->mfma = !valu_only && u->fin_m.w && s.uid % 8 == 0;
  for (size_t r = 0; r < u->rtb.size(); ++r) {
    const LRtb& R = u->rtb[r];
    u->mfma = u->mfma && R.ma.w && R.mb.w && (!R.res || R.mr.w) && (r == 0 || R.cin % 8 == 0);
  }
  for (int i = 0; i < s.n_levels - 1; ++i) u->mfma = u->mfma && u->down_m[i].w && u->up_m[i].w;
  u->fin_w5 = push_wt(blob, tensors[s.t_final[0]], s.uid, s.uid, 5, false);
  u->fin_b5 = push_v(blob, tensors[s.t_final[1]], s.uid);
  u->fin_g = push_v(blob, tensors[s.t_final[2]], s.uid);
  u->fin_be = push_v(blob, tensors[s.t_final[3]], s.uid);
  u->fin_w1 = push_wt(blob, tensors[s.t_final[4]], 4, s.uid, 1, false);
  u->fin_b1 = push_v(blob, tensors[s.t_final[5]], 4);
  if (hipMalloc(&u->blob, blob.size() * sizeof(float)) != hipSuccess ||
      hipMalloc(&u->ttable, (size_t)T * u->tb_total * sizeof(float)) != hipSuccess) {
    set_error("mmd_unet_create: hipMalloc failed");
    return 1;
  }
  // widest Conv1dBlock input: ups.0.0 of a four-level net stages 2 x 8 uid x 8 channels x (8 + 8) positions (64 KB at uid 64)
  for (const void* f : {(const void*)conv5_block_kernel<1>, (const void*)conv5_block_kernel<2>, (const void*)conv5_block_kernel<4>})
    MMD_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  if (mconv_set_lds<0>() || mconv_set_lds<1>() || mconv_set_lds<2>() || mconv_set_lds<3>() || mconv_set_lds<4>()) return 1;
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
  *out = owner.release();
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
  // a layer on the matrix pipe (mconv_kernel): a workgroup = 64 / (rows of a sample) samples x a slice of <= 128 of the GEMM's columns;
  // narrower slices (down to `unit`: whole n-tiles, for a Conv1dBlock whole GroupNorm groups) while the launch has fewer than 3
  // workgroups per CU
  auto mconv_ok = [&](const MPack& mp, int c1, int c2, int in_cl) { return u->mfma && mp.w && (in_cl ? c1 == 4 && !c2 : c1 % 8 == 0 && c2 % 8 == 0); };
  auto mconv = [&](int kind, const MPack& mp, ConvArgs c, int cols, int unit, const MPack* mp2 = nullptr, size_t b2 = 0, size_t g2 = 0,
                   size_t be2 = 0) -> int {
    const int rows = kind == 2 ? c.l_in / 2 : c.l_in, spw = 64 / rows, n_wg = (n + spw - 1) / spw;
    int cs = 0;                                            // the widest slice that leaves >= 768 workgroups, else the narrowest there is
    for (int w : {128, 64, 32, 16})
      if (w <= u->mconv_max_cs && cols % w == 0 && w % unit == 0 && (!cs || (long long)n_wg * (cols / cs) < 768)) cs = w;
    if (kind == 4) cs = cols;                              // a whole ResidualTemporalBlock: one slice (32, 64 or 128 channels)
    MMD_REQUIRE(cs && rows >= 8 && rows <= 64, "layered_forward: no slice for a layer of %d columns", cols);
    MConvArgs ma{};
    ma.c = c;
    ma.c.cs = cs;
    ma.wpk = reinterpret_cast<const uint4*>(B + mp.w);
    ma.isc = B + mp.isc;
    ma.stride = mp.stride;
    ma.n_chunks = mp.n_chunks;
    ma.kch = mp.kch;
    ma.n = n;
    if (kind == 4) {
      MMD_REQUIRE(mp2 && mp2->w && mp2->n_chunks == 1 && (cols == 32 || cols == 64 || cols == 128), "layered_forward: no fused block of %d channels", cols);
      ma.wpk2 = reinterpret_cast<const uint4*>(B + mp2->w);
      ma.isc2 = B + mp2->isc;
      ma.bias2 = B + b2; ma.gamma2 = B + g2; ma.beta2 = B + be2;
    }
    // LDS: the widest slab (the staged chunk; KIND 4: also the hidden tensor's c_out channels), or the output tile over it
    const size_t slab = (size_t)2 * (std::max(mp.kch, kind == 4 ? cols : 0) / 8) * spw * (c.l_in + 4) * 16,
                 outs = kind == 0 || kind == 4 ? (size_t)cs * OST * 4 : 0;
    const size_t shm = std::max(slab, outs) + (64 + 24) * sizeof(float);   // + the statistics' exchange + the samples' maxima
    const void* const fn = kind == 0 ? mconv_fn<0>(cs, mp.kch) : kind == 1 ? mconv_fn<1>(cs, mp.kch) : kind == 2 ? mconv_fn<2>(cs, mp.kch)
                         : kind == 3 ? mconv_fn<3>(cs, mp.kch) : mconv_fn<4>(cs, mp.kch);
    // the items (64 / rows samples each) of a slice go to as many workgroups as the chip holds at once, the same number each (+- 1): a
    // workgroup requests its next item's rows under the current one's GEMM and tail
    const int ny = cols / cs, cap = std::max(1, mconv_resident(fn, shm) * kNumCUs / ny), per_wg = (n_wg + cap - 1) / cap;
    const dim3 grid((n_wg + per_wg - 1) / per_wg, ny);
    void* args[] = {&ma};
    MMD_HIP_CHECK(hipLaunchKernel(fn, grid, dim3(256), args, shm, st));
    return 0;
  };
  // Conv1dBlock: (x1 | x2) [c][L] -> Mish(GN(conv5)) + add_c[c] + add_t -> y.  A workgroup of KS x nconv threads, nconv =
  // (L / 4) * (cs / CT) in [16, 64]: the slice width cs (whole GroupNorm groups) follows from that, CT = 2 for the big launches
  auto block5 = [&](const float* x1, int c1, const float* x2, int c2, int in_cl, int L, int c_out, size_t w, size_t b, size_t gamma,
                    size_t beta, const float* add_c, const float* add_t, float* y, const MPack& mp) -> int {
    const int cpg = c_out / N_GROUPS;
    if (u->mfma) {
      MMD_REQUIRE(mconv_ok(mp, c1, c2, in_cl), "layered_forward: a Conv1dBlock of %d -> %d channels has no matrix-pipe form", c1 + c2, c_out);
      return mconv(0, mp, ConvArgs{x1, x2, c1, c2, L, L, c_out, 0, in_cl, 0, nullptr, B + b, B + gamma, B + beta, add_c, add_t, y}, c_out,
                   mfma_slice(c_out));
    }
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
                   int out_cl, size_t w, size_t b, float* y, const MPack& mp) -> int {
    if (u->mfma && !out_cl) {
      MMD_REQUIRE(mconv_ok(mp, c1, c2, in_cl), "layered_forward: a conv of %d -> %d channels has no matrix-pipe form", c1 + c2, c_out);
      return mconv(mode == 1 ? 3 : k == 3 ? 2 : 1, mp, ConvArgs{x1, x2, c1, c2, l_in, l_out, c_out, 0, in_cl, 0, nullptr, B + b, nullptr,
                                                                 nullptr, nullptr, nullptr, y}, mode == 1 ? 2 * c_out : c_out, 16);
    }
    const int ns = c_out >= 32 ? slices(c_out, 8, 1 << 30) : 1;
    ConvArgs a{x1, x2, c1, c2, l_in, l_out, c_out, c_out / ns, in_cl, out_cl, B + w, B + b, nullptr, nullptr, nullptr, nullptr, y, u->mfma && !in_cl};
    MMD_REQUIRE(!a.in_bl || !x2, "layered_forward: a blocked input is one tensor");
    const size_t shm = (size_t)l_in * (c1 + c2) * sizeof(float);
    if (mode == 1) hipLaunchKernelGGL((conv_plain_kernel<1, 4, 2>), dim3(n, ns), dim3(256), shm, st, a);
    else if (k == 3) hipLaunchKernelGGL((conv_plain_kernel<0, 3, 2>), dim3(n, ns), dim3(256), shm, st, a);
    else hipLaunchKernelGGL((conv_plain_kernel<0, 1, 1>), dim3(n, ns), dim3(256), shm, st, a);
    return 0;
  };
  // one ResidualTemporalBlock (layers.py:346-358): (x1 | x2) [cin][L] -> out [cout][L]
  auto rtb = [&](const LRtb& R, const float* x1, int c1, const float* x2, int c2, int in_cl, int L, float* out) -> int {
    // matrix pipe, 32 / 64 / 128 output channels: the two Conv1dBlocks in ONE launch (mconv_kernel KIND 4: the hidden tensor stays in the
    // workgroup; same bits as the two launches)
    const bool fused = u->mfma && R.mb.n_chunks == 1 && (R.cout == 32 || R.cout == 64 || R.cout == 128) && R.cout <= u->rtb_fused;
    const float* res = x1;                                   // identity residual (cin == cout: never a concatenated input)
    if (fused && R.res) {
      if (int rc = plain(0, 1, x1, c1, x2, c2, in_cl, L, L, R.cout, 0, R.wr, R.br, resb, R.mr)) return rc;
      res = resb;
    }
    if (fused) {
      MMD_REQUIRE(mconv_ok(R.ma, c1, c2, in_cl), "layered_forward: a block of %d -> %d channels has no matrix-pipe form", c1 + c2, R.cout);
      return mconv(4, R.ma, ConvArgs{x1, x2, c1, c2, L, L, R.cout, 0, in_cl, 0, nullptr, B + R.ba, B + R.ga, B + R.bea, tt + R.tb_off, res, out},
                   R.cout, R.cout, &R.mb, R.bb, R.gb, R.beb);
    }
    if (int rc = block5(x1, c1, x2, c2, in_cl, L, R.cout, R.wa, R.ba, R.ga, R.bea, tt + R.tb_off, nullptr, tmp, R.ma)) return rc;
    if (R.res) {
      if (int rc = plain(0, 1, x1, c1, x2, c2, in_cl, L, L, R.cout, 0, R.wr, R.br, resb, R.mr)) return rc;
      res = resb;
    }
    return block5(tmp, R.cout, nullptr, 0, 0, L, R.cout, R.wb, R.bb, R.gb, R.beb, nullptr, res, out, R.mb);
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
      if (int rc = plain(0, 3, level_out, c, nullptr, 0, 0, L, L / 2, c, 0, u->down_w[i], u->down_b[i], in[i & 1], u->down_m[i])) return rc;
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
    if (int rc = plain(1, 4, in[1], din, nullptr, 0, 0, L, 2 * L, din, 0, u->up_w[i], u->up_b[i], in[0], u->up_m[i])) return rc;
    L *= 2;
  }
  // final_conv (temporal_unet.py:104-110); with one level there are no ups and L is still 64
  if (int rc = block5(in[0], s.uid, nullptr, 0, 0, L, s.uid, u->fin_w5, u->fin_b5, u->fin_g, u->fin_be, nullptr, nullptr, mid, u->fin_m)) return rc;
  if (int rc = plain(0, 1, mid, s.uid, nullptr, 0, 0, L, L, 4, 1, u->fin_w1, u->fin_b1, eps, MPack{})) return rc;
  MMD_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace mmd

#ifdef MCONV_TIMING
extern "C" __attribute__((visibility("default"))) int mmd_debug_mconv_clocks(unsigned long long* out) {   // [5][4][9][8], cleared after the read
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mmd::g_mconv_clk), sizeof(mmd::g_mconv_clk)) != hipSuccess) return 1;
  static unsigned long long zero[5 * 4 * 9 * 8];
  return hipMemcpyToSymbol(HIP_SYMBOL(mmd::g_mconv_clk), zero, sizeof(zero)) != hipSuccess;
}
#endif
