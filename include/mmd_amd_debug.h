/*
 * mmd_amd_debug.h -- measurement hooks of libmmd_amd.so (bench.py / tools only; NOT part of the drop-in boundary of
 * include/mmd_amd.h).  The product handles (mmd_unet_t) are immutable: all profiling state lives in a caller-owned
 * mmd_profiler_t that is handed to the sampler through mmd_sampler_desc.profiler.
 */
#ifndef MMD_AMD_DEBUG_H
#define MMD_AMD_DEBUG_H

#include <stddef.h>
#include <stdint.h>

#include "mmd_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Event-pair pool: every `stride`-th UNet launch issued with this profiler attached is bracketed by a HIP event pair
 * recorded on the stream the kernel is launched on, up to max_launches pairs (an event pair costs ~10 us of host/stream
 * time, so a stride keeps the perturbation of a timed region below 0.5 %).  One profiler must not be shared by
 * concurrent calls. */
typedef struct mmd_profiler_s* mmd_profiler_t;
int mmd_profiler_create(mmd_profiler_t* out, int max_launches, int stride);
int mmd_profiler_destroy(mmd_profiler_t p);
/* Windowed form: launches are counted modulo `period` (0 = never wrap; e.g. the launches of one sampling call) and launch i
 * of a period is bracketed iff (i / window) % stride == 0 -- `window` consecutive launches (e.g. every stream chunk of two
 * consecutive steps) out of every window * stride, so that the overlap of concurrent launches can be read off the intervals. */
int mmd_profiler_create_windowed(mmd_profiler_t* out, int max_launches, int stride, int window, int period);
/* Kinds of bracketed launches: the UNet forward alone (MMD_PROF_UNET), the UNet forward with an unguided DDPM step fused into
 * its tail (MMD_PROF_UNET_FUSED: posterior mean + Philox noise + hard conditioning ride in the same launch), and the DDPM-step
 * kernel of a guided / an unguided step (the step kernels of the same steps as the UNet launches are bracketed). */
enum { MMD_PROF_UNET = 0, MMD_PROF_STEP_GUIDED = 1, MMD_PROF_STEP_PLAIN = 2, MMD_PROF_UNET_FUSED = 3 };
/* After synchronising the stream(s), BEFORE mmd_profiler_read: [start, end] of every bracketed launch of `kind` in ms since
 * the first bracket (one clock across streams), up to `cap` intervals. */
int mmd_profiler_intervals(mmd_profiler_t p, int kind, double* start_ms, double* end_ms, int cap, int* n_out);
/* After synchronising the stream(s): mean duration [ms] and number of bracketed UNet launches; rearms the pool. */
int mmd_profiler_read(mmd_profiler_t p, double* mean_ms, int* n_launches);

/* One TemporalUnet forward = one launch of unet_kernel.  Algorithmic FLOPs per trajectory (direct-convolution count:
 * 2 * C_out * taps * C_in * L_out over its convs); the fp32 GEMM FLOPs the kernel actually runs on the matrix
 * pipe per trajectory (every conv a direct GEMM; channel / N padding and the strided tails' discarded positions included); and
 * the part of the latter that runs as f16x2 on the fp16 pipe (two fp16 pieces per operand, 3 fp16 MFMA FLOPs per fp32
 * FLOP) -- all of it since round 3. */
double mmd_unet_flops_per_trajectory(void);
double mmd_unet_mfma_flops_per_trajectory(void);
double mmd_unet_f16x2_flops_per_trajectory(void);

/* How many concurrent stream chunks mmd_p_sample_loop splits a batch into for this n_streams setting (0 = the
 * library's automatic choice): what a measurement needs to know the launch shape. */
int mmd_sampler_stream_chunks(int n_streams, int n_robots, int samples_per_robot);

/* mmd_unet_forward with the profiler attached (what mmd_p_sample_loop does internally when desc.profiler is set). */
int mmd_unet_forward_profiled(mmd_unet_t unet, const float* x_dev, int t, float* eps_dev, int n_traj,
                              void* workspace_dev, size_t workspace_bytes, mmd_profiler_t profiler, void* stream);

/* Bytes of the packed weight / parameter block a TemporalUnet forward reads (the fused kernel's f16x2 packs + biases + GroupNorm
 * affines + scales; the layer-by-layer path: its fp32 blob): with 2 KiB per trajectory the ALGORITHMIC HBM-side bytes of one
 * launch, the denominator of bench.py's roofline.traffic / wasted ratio. */
size_t mmd_unet_weight_bytes(mmd_unet_t unet);

/* Decision trace of ONE guided ddpm_sample_fn step (sample_functions.py:40-107): mmd_ddpm_step on the one-wave step kernel with
 * the dump compiled in -- the same arithmetic, bit for bit, as every production launch shape -- which additionally writes
 *   mu_dev          [n_traj][64][4]  (optional) the posterior mean as the first guide iteration sees it (hard rows are pinned after an iteration, not before the first)
 *   guide_chain_dev [n_guide_steps][n_traj][64][4]  the state after every guide iteration (before the step's noise)
 *   trace_dev       [n_guide_steps][n_traj][64][MMD_TRACE_WORDS] uint32: the discrete decisions of the iteration at that support
 *                   point, i.e. everything in GuideManagerTrajectoriesWithVelocity.forward that is not continuous in x:
 *     word 0      SDF cell index ix * ny + iy of the nearest-cell lookup (grid_map_sdf.py:84-114)
 *     word 1      bit 0: the object-collision hinge is active (margin - sdf > 0, distance_fields.py:110-135), bits 1-3: the field
 *                 that wins the max (grid index; 7 = the env's extra objects); bit 4: the workspace-boundary hinge is active,
 *                 bits 5-6: its arg max (x-min, y-min, x-max, y-max; distance_fields.py:354-367); bits 8 / 9 / 10: the gradient
 *                 clip (guides.py:228-259) is active on the object / boundary / GP term, bits 11-14: on constraint group 0..3 of the
 *                 robot; bits 16-19: the un-normalisation clips dimension 0..3 (normalization.py:161-163); bits 20-23: the
 *                 robot's number of constraint groups
 *     words 2-9   four 64-bit masks (lo, hi): slot s of constraint group 0..3 is ACTIVE at this support point (||p - q|| <= R
 *                 inside its time range, cost_functions.py:305-312); slot = the point's rank among the group's points that cover
 *                 this support point, in list order (mmd_pack_constraints) / the other robot's index (mmd_soft_constraints_from_paths)
 *     word 10     number of active slots over ALL groups and slots (also those beyond 4 groups x 64 slots)
 *     word 11     0
 * tests/test_gpu_flips.py holds the oracle's decisions against these. */
#define MMD_TRACE_WORDS 12
int mmd_debug_ddpm_step_trace(mmd_unet_t unet, const mmd_sampler_desc* s, const mmd_guide_desc* guide, float* x_dev,
                              const float* hard_dev, int n_robots, int samples_per_robot, int i, const float* noise_dev,
                              uint64_t seed, uint32_t draw_index, void* workspace_dev, size_t workspace_bytes,
                              float* mu_dev, float* guide_chain_dev, uint32_t* trace_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMD_AMD_DEBUG_H */
