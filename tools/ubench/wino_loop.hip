// What bounds the Winograd k-step (6 LDS reads + 12 VALU + 6 MFMA + 24 B/lane of weights)?  Variants switch the LDS
// reads, the global weight loads and the VALU transform off one at a time.  Usage: wino_loop [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(8)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(8)));
struct B6 { f32x4u lo; f32x2u hi; };
typedef int i4_ __attribute__((ext_vector_type(4)));
typedef int i2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ B6 buf_b6(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  B6 b;
  i4_ x = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  i2_ y = __builtin_amdgcn_raw_buffer_load_b64(r, voff + 16, soff, 0);
  b.lo = __builtin_bit_cast(f32x4u, x); b.hi = __builtin_bit_cast(f32x2u, y);
  return b;
}
#define PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
constexpr int STR = 129, KSTRIDE = 64 * 6, KS = 64;

__device__ __forceinline__ B6 load_b6(const float* __restrict__ p) {
  B6 b; b.lo = *reinterpret_cast<const f32x4u*>(p); b.hi = *reinterpret_cast<const f32x2u*>(p + 4); return b;
}
template <bool VALU>
__device__ __forceinline__ void step(f32x16 (&m)[6], const float (&d)[6], const B6& b) {
  float v0, v1, v2, v3, v4, v5;
  if (VALU) {
    const float a = fmaf(-4.f, d[2], d[4]), bb = fmaf(-4.f, d[1], d[3]);
    const float c = d[4] - d[2], e = d[3] - d[1];
    v0 = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
    v1 = a + bb; v2 = a - bb; v3 = fmaf(2.f, e, c); v4 = fmaf(-2.f, e, c);
    v5 = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
  } else { v0 = d[0]; v1 = d[1]; v2 = d[2]; v3 = d[3]; v4 = d[4]; v5 = d[5]; }
  m[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, b.lo[0], m[0], 0, 0, 0);
  m[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, b.lo[1], m[1], 0, 0, 0);
  m[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2, b.lo[2], m[2], 0, 0, 0);
  m[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3, b.lo[3], m[3], 0, 0, 0);
  m[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(v4, b.hi[0], m[4], 0, 0, 0);
  m[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(v5, b.hi[1], m[5], 0, 0, 0);
}

template <int LDS, int GLD, bool VALU>
__global__ __launch_bounds__(256) void wino(const float* __restrict__ w, float* __restrict__ out, int reps) {
  const long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  __shared__ float slab[4 * 20 * STR + 64];
  for (int i = threadIdx.x; i < 4 * 20 * STR + 64; i += 256) slab[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int srow = (lane & 31) / 8, tile = (lane & 31) % 8;
  // LDS == 2: lane stride STR (odd) instead of 2 * STR: bank-conflict free
  const float* s0 = slab + srow * 20 * STR + (LDS == 2 ? 1 : 2) * tile * STR + (lane >> 5);
  const float* wp0 = w + ((size_t)wave * KS * KSTRIDE) + lane * 6;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(w + (size_t)wave * KS * KSTRIDE), 0, (KS + 16) * KSTRIDE * 4, 0x00020000);
  f32x16 m[6];
  float dummy = 0.f;
  for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) m[p][r] = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    const float* p = wp0; const float* s = s0;
    B6 b[4], bx[4];
    for (int j = 0; j < 4; ++j) bx[j] = b[j] = load_b6(p + j * KSTRIDE);
    float d[2][6], dx[2][6];
    for (int j = 0; j < 6; ++j) dx[0][j] = dx[1][j] = d[0][j] = d[1][j] = s[j * STR];
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += 4) {
      p += 4 * KSTRIDE;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (LDS == 1 || LDS == 2) { for (int q = 0; q < 6; ++q) d[(j + 1) & 1][q] = s[2 * (j + 1) + q * STR]; }
        if (LDS == 3) { for (int q = 0; q < 6; ++q) dx[(j + 1) & 1][q] = s[2 * (j + 1) + q * STR]; }
        if (LDS == 4) { for (int q = 0; q < 6; ++q) d[(j + 1) & 1][q] = d[j & 1][q] * 1.0001f; }   // register-only, not loop invariant
        PIN();
        if (LDS == 3) dummy += (dx[j & 1][0] + dx[j & 1][1]) + (dx[j & 1][2] + dx[j & 1][3]) + (dx[j & 1][4] + dx[j & 1][5]);
        if (GLD == 3) dummy += bx[j].lo[0] + bx[j].lo[1] + bx[j].lo[2] + bx[j].lo[3] + bx[j].hi[0] + bx[j].hi[1];
        step<VALU>(m, d[j & 1], b[j]);
        if (GLD == 3) bx[j] = load_b6(p + j * KSTRIDE);
        if (GLD == 1) b[j] = load_b6(p + j * KSTRIDE);
        if (GLD == 4) b[j] = buf_b6(rsrc, lane * 24, (int)((p - wp0) + j * KSTRIDE) * 4);
        if (GLD == 2) {   // dense: three 16-byte loads per TWO k-steps (same bytes, lanes contiguous)
          const float4* q = reinterpret_cast<const float4*>(w + (size_t)(threadIdx.x >> 6) * KS * KSTRIDE) + ((ks + j) / 2 * 3) * 64 + lane;
          if ((j & 1) == 0) { float4 x = q[0], y = q[64]; b[j].lo = {x.x, x.y, x.z, x.w}; b[j].hi = {y.x, y.y}; b[j + 1].lo[0] = y.z; b[j + 1].lo[1] = y.w; }
          else { float4 z = q[128]; b[j].lo[2] = z.x; b[j].lo[3] = z.y; b[j].hi = {z.z, z.w}; }
        }
        PIN();
      }
      s += 8;
    }
  }
  float acc = dummy;
  for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) acc += m[p][r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (blockIdx.x == 7 && threadIdx.x == 0) { long long* c = reinterpret_cast<long long*>(out + (size_t)gridDim.x * 256 + 2); c[0] = wall_clock64() - w0; c[1] = __builtin_readcyclecounter() - c0; }
}

__device__ __forceinline__ void xform(float (&v)[6], const float (&d)[6]) {
  const float a = fmaf(-4.f, d[2], d[4]), bb = fmaf(-4.f, d[1], d[3]);
  const float c = d[4] - d[2], e = d[3] - d[1];
  v[0] = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
  v[1] = a + bb; v[2] = a - bb; v[3] = fmaf(2.f, e, c); v[4] = fmaf(-2.f, e, c);
  v[5] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
}
__device__ __forceinline__ void mfma6(f32x16 (&m)[6], const float (&v)[6], const B6& b) {
  m[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], b.lo[0], m[0], 0, 0, 0);
  m[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[1], b.lo[1], m[1], 0, 0, 0);
  m[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2], b.lo[2], m[2], 0, 0, 0);
  m[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[3], b.lo[3], m[3], 0, 0, 0);
  m[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[4], b.hi[0], m[4], 0, 0, 0);
  m[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[5], b.hi[1], m[5], 0, 0, 0);
}
// software-pipelined: the transform of k-step i+1 is issued between the MFMAs of k-step i, the LDS reads run two ahead
template <int LDSM>
__global__ __launch_bounds__(256) void wino_pipe(const float* __restrict__ w, float* __restrict__ out, int reps) {
  __shared__ float slab[4 * 20 * STR + 64];
  for (int i = threadIdx.x; i < 4 * 20 * STR + 64; i += 256) slab[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int srow = (lane & 31) / 8, tile = (lane & 31) % 8;
  const float* s0 = slab + srow * 20 * STR + (LDSM == 2 ? 1 : 2) * tile * STR + (lane >> 5);
  const float* wp0 = w + ((size_t)wave * KS * KSTRIDE) + lane * 6;
  f32x16 m[6];
  for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) m[p][r] = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    const float* p = wp0; const float* s = s0;
    B6 b[4];
    for (int j = 0; j < 4; ++j) b[j] = load_b6(p + j * KSTRIDE);
    float d[6], v[2][6];
    for (int q = 0; q < 6; ++q) d[q] = s[q * STR];
    xform(v[0], d);
    for (int q = 0; q < 6; ++q) d[q] = s[2 + q * STR];
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += 4) {
      p += 4 * KSTRIDE;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xform(v[(j + 1) & 1], d);                                         // k-step j+1 from d (loaded one step ago)
        for (int q = 0; q < 6; ++q) d[q] = s[2 * (j + 2) + q * STR];      // reads for k-step j+2
        mfma6(m, v[j & 1], b[j]);
        b[j] = load_b6(p + j * KSTRIDE);
        // interleave: 1 MFMA, then 2 VALU / 1 LDS read in its shadow
        for (int g = 0; g < 6; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);     // VMEM read
        PIN();
      }
      s += 8;
    }
  }
  float acc = 0;
  for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) acc += m[p][r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

#define SB() __builtin_amdgcn_sched_barrier(0)
// hand-interleaved: MFMA p of k-step j, then the two VALU ops that build part of k-step j+1's A operands (from d loaded
// during k-step j-1); LDS reads for k-step j+2 and the weight refill at the end.  sched_barrier after every group.
template <int LDSM>
__global__ __launch_bounds__(256) void wino_pipe2(const float* __restrict__ w, float* __restrict__ out, int reps) {
  __shared__ float slab[4 * 20 * STR + 64];
  for (int i = threadIdx.x; i < 4 * 20 * STR + 64; i += 256) slab[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int srow = (lane & 31) / 8, tile = (lane & 31) % 8;
  const float* s0 = slab + srow * 20 * STR + (LDSM == 2 ? 1 : 2) * tile * STR + (lane >> 5);
  const float* wp0 = w + ((size_t)wave * KS * KSTRIDE) + lane * 6;
  f32x16 m[6];
  for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) m[p][r] = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    const float* p = wp0; const float* s = s0;
    B6 b[4];
    for (int j = 0; j < 4; ++j) b[j] = load_b6(p + j * KSTRIDE);
    float d[2][6], v[2][6];
    for (int q = 0; q < 6; ++q) d[1][q] = s[q * STR];
    xform(v[0], d[1]);                                     // k-step 0
    for (int q = 0; q < 6; ++q) d[1][q] = s[2 + q * STR];  // k-step 1 (consumed during step 0)
    SB();
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += 4) {
      p += 4 * KSTRIDE;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float (&dc)[6] = d[(j + 1) & 1];     // data of k-step j+1 (loaded one step ago)
        float (&dn)[6] = d[j & 1];           // buffer for k-step j+2
        float (&vc)[6] = v[j & 1];
        float (&vn)[6] = v[(j + 1) & 1];
        for (int q = 0; q < 6; ++q) dn[q] = s[2 * (j + 2) + q * STR];
        SB();
        m[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[0], b[j].lo[0], m[0], 0, 0, 0);
        const float a = fmaf(-4.f, dc[2], dc[4]), bb = fmaf(-4.f, dc[1], dc[3]);
        SB();
        m[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[1], b[j].lo[1], m[1], 0, 0, 0);
        vn[1] = a + bb; vn[2] = a - bb;
        SB();
        m[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[2], b[j].lo[2], m[2], 0, 0, 0);
        const float c = dc[4] - dc[2], e = dc[3] - dc[1];
        SB();
        m[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[3], b[j].lo[3], m[3], 0, 0, 0);
        vn[3] = fmaf(2.f, e, c); vn[4] = fmaf(-2.f, e, c);
        SB();
        m[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[4], b[j].hi[0], m[4], 0, 0, 0);
        vn[0] = fmaf(4.f, dc[0], fmaf(-5.f, dc[2], dc[4]));
        SB();
        m[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[5], b[j].hi[1], m[5], 0, 0, 0);
        vn[5] = fmaf(4.f, dc[1], fmaf(-5.f, dc[3], dc[5]));
        SB();
        b[j] = load_b6(p + j * KSTRIDE);
        PIN();
      }
      s += 8;
    }
  }
  float acc = 0;
  for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) acc += m[p][r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
struct B12 { f4 a, b, c; };
__device__ __forceinline__ B12 load_b12(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  B12 x;
  x.a = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  x.b = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 1024, soff, 0));
  x.c = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 2048, soff, 0));
  return x;
}
// two k-steps per unit: ds_read2_b32 fetches (channel c, c + 2) pairs, the transform runs on packed pairs (v_pk_*),
// weights come through buffer loads with an SGPR running offset (no per-iteration VGPR address arithmetic)
template <int RING>
__global__ __launch_bounds__(256) void wino_packed(const float* __restrict__ w, float* __restrict__ out, int reps) {
  const long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  __shared__ float slab[4 * 20 * STR + 64];
  for (int i = threadIdx.x; i < 4 * 20 * STR + 64; i += 256) slab[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int srow = (lane & 31) / 8, tile = (lane & 31) % 8;
  const float* s0 = slab + srow * 20 * STR + 2 * tile * STR + (lane >> 5);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(w + (size_t)wave * KS * KSTRIDE), 0, (KS + 16) * KSTRIDE * 4, 0x00020000);
  const int voff = lane * 16;
  f32x16 m[6];
  for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) m[p][r] = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    const float* s = s0;
    int soff = 0;
    B12 b[RING];
    for (int j = 0; j < RING; ++j) b[j] = load_b12(rs, voff, soff + j * 3072);
    f2 d[2][6];
    for (int q = 0; q < 6; ++q) d[0][q] = f2{s[q * STR], s[q * STR + 2]};
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += 2 * RING) {
      soff += RING * 3072;
#pragma unroll
      for (int j = 0; j < RING; ++j) {
        for (int q = 0; q < 6; ++q) d[(j + 1) & 1][q] = f2{s[4 * (j + 1) + q * STR], s[4 * (j + 1) + q * STR + 2]};
        PIN();
        const f2 (&D)[6] = d[j & 1];
        const f2 a = -4.f * D[2] + D[4], bb = -4.f * D[1] + D[3];
        const f2 c = D[4] - D[2], e = D[3] - D[1];
        const f2 V0 = 4.f * D[0] + (-5.f * D[2] + D[4]);
        const f2 V1 = a + bb, V2 = a - bb, V3 = 2.f * e + c, V4 = -2.f * e + c;
        const f2 V5 = 4.f * D[1] + (-5.f * D[3] + D[5]);
        const B12& B = b[j];
        m[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(V0.x, B.a[0], m[0], 0, 0, 0);
        m[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V1.x, B.a[1], m[1], 0, 0, 0);
        m[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(V2.x, B.a[2], m[2], 0, 0, 0);
        m[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(V3.x, B.a[3], m[3], 0, 0, 0);
        m[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(V4.x, B.b[0], m[4], 0, 0, 0);
        m[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(V5.x, B.b[1], m[5], 0, 0, 0);
        m[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(V0.y, B.b[2], m[0], 0, 0, 0);
        m[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V1.y, B.b[3], m[1], 0, 0, 0);
        m[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(V2.y, B.c[0], m[2], 0, 0, 0);
        m[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(V3.y, B.c[1], m[3], 0, 0, 0);
        m[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(V4.y, B.c[2], m[4], 0, 0, 0);
        m[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(V5.y, B.c[3], m[5], 0, 0, 0);
        b[j] = load_b12(rs, voff, soff + j * 3072);
        PIN();
      }
      s += 4 * RING;
    }
  }
  float acc = 0;
  for (int p = 0; p < 6; ++p) for (int r = 0; r < 16; ++r) acc += m[p][r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (blockIdx.x == 7 && threadIdx.x == 0) { long long* c = reinterpret_cast<long long*>(out + (size_t)gridDim.x * 256 + 2); c[0] = wall_clock64() - w0; c[1] = __builtin_readcyclecounter() - c0; }
}

template <int RING>
void run_packed(const char* name, int wps, const float* w, float* out) {
  const int blocks = 256 * wps, reps = 200;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((wino_packed<RING>), dim3(blocks), dim3(256), 0, 0, w, out, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double ns_per_kstep = best * 1e6 / ((double)reps * KS);
  long long clk[2];
  (void)hipMemcpy(clk, out + (size_t)blocks * 256 + 2, sizeof(clk), hipMemcpyDeviceToHost);
  const double ghz = (double)clk[1] / ((double)clk[0] * 10.0);
  printf("%-28s waves/SIMD %d: %8.3f ms  %6.1f ns / k-step  %6.1f TF executed  clock %.2f GHz -> %.0f cycles / k-step / SIMD\n", name, wps, best,
         ns_per_kstep, (double)blocks * 4 * reps * KS * 6 * 4096.0 / best / 1e9, ghz, ns_per_kstep * ghz / wps);
}

template <int LDSM, bool V2>
void run_pipe(const char* name, int wps, const float* w, float* out) {
  const int blocks = 256 * wps, reps = 200;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0);
    if (V2) hipLaunchKernelGGL((wino_pipe2<LDSM>), dim3(blocks), dim3(256), 0, 0, w, out, reps);
    else hipLaunchKernelGGL((wino_pipe<LDSM>), dim3(blocks), dim3(256), 0, 0, w, out, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%-28s waves/SIMD %d: %8.3f ms  %6.1f ns / k-step  %6.1f TF executed\n", name, wps, best,
         best * 1e6 / ((double)reps * KS), (double)blocks * 4 * reps * KS * 6 * 4096.0 / best / 1e9);
}

template <int LDS, int GLD, bool VALU>
void run(const char* name, int wps, const float* w, float* out) {
  const int blocks = 256 * wps, reps = 200;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((wino<LDS, GLD, VALU>), dim3(blocks), dim3(256), 0, 0, w, out, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double ns_per_kstep = best * 1e6 / ((double)reps * KS);
  long long clk[2];
  (void)hipMemcpy(clk, out + (size_t)blocks * 256 + 2, sizeof(clk), hipMemcpyDeviceToHost);
  const double ghz = (double)clk[1] / ((double)clk[0] * 10.0);
  printf("%-28s waves/SIMD %d: %8.3f ms  %6.1f ns / k-step  %6.1f TF executed  clock %.2f GHz -> %.0f cycles / k-step / SIMD\n", name, wps, best,
         ns_per_kstep, (double)blocks * 4 * reps * KS * 6 * 4096.0 / best / 1e9, ghz, ns_per_kstep * ghz / wps);
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;
  const size_t nw = (size_t)(4 * KS + 16) * KSTRIDE;
  std::vector<float> h(nw);
  for (auto& v : h) v = (float)(rand() % 1000) / 1000.f - 0.5f;
  float *w, *out;
  (void)hipMalloc(&w, nw * 4); (void)hipMalloc(&out, (size_t)256 * wps * 256 * 4 + 64);
  (void)hipMemcpy(w, h.data(), nw * 4, hipMemcpyHostToDevice);
  run_packed<2>("packed, ring 2 pairs", wps, w, out);
  run_packed<4>("packed, ring 4 pairs", wps, w, out);
  run_pipe<1, true>("hand-interleaved", wps, w, out);
  run_pipe<2, true>("hand-interleaved, conflict-free", wps, w, out);
  run<1, 1, true>("full", wps, w, out);
  run<1, 4, true>("buffer loads (SGPR offset)", wps, w, out);
  run<2, 1, true>("conflict-free LDS", wps, w, out);
  run<1, 2, true>("dense weight loads", wps, w, out);
  run<2, 2, true>("conflict-free + dense", wps, w, out);
  run<0, 1, true>("no LDS reads", wps, w, out);
  run<1, 0, true>("no weight loads", wps, w, out);
  run<1, 1, false>("no VALU transform", wps, w, out);
  run<3, 0, true>("LDS issued, side-consumed", wps, w, out);
  run<0, 3, true>("GLD issued, side-consumed", wps, w, out);
  run<3, 3, true>("both issued, side-consumed", wps, w, out);
  run<4, 0, true>("live VALU (18 ops) + MFMA", wps, w, out);
  run<4, 0, false>("live VALU (6 ops) + MFMA", wps, w, out);
  run<4, 1, true>("live VALU + GLD + MFMA", wps, w, out);
  run<0, 0, true>("VALU + MFMA only", wps, w, out);
  run<0, 0, false>("MFMA only", wps, w, out);
  return 0;
}
