/*
 * vkvolume_amd_debug.h - diagnostic entry points of libvkvolume_amd.so (NOT part of the drop-in boundary; nothing a renderer binds).
 * They exist for the measurement tools under tools/ and for the exhaustive numerics checks of tests/: per-wave timelines of the
 * ray-march launches, start-order experiments, and device-side proofs that a short-cut of a kernel equals the plain IEEE form.
 */
#ifndef VKVOLUME_AMD_DEBUG_H
#define VKVOLUME_AMD_DEBUG_H

#include "vkvolume_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-wave timeline of the next vkv_render / vkv_render_batch launches into d_buffer (10 x u64 per wave of the grid: start, end
 * [100 MHz clock], iterations, hardware id | unit, phase sums); NULL switches it off. */
int vkv_debug_trace(vkv_ctx *ctx, void *d_buffer);

/* The next vkv_render_batch launches take frame i's tile start order from d_orders + i * count (device array of schedule-entry indices,
 * a permutation of 0 .. count - 1) instead of the centre-first order, when count equals the schedule's tile count; NULL switches it off. */
int vkv_debug_tile_orders(vkv_ctx *ctx, const uint32_t *d_orders, uint32_t frames, uint32_t count);

/* Counts the floats with bit patterns [first_bits, first_bits + count) for which a short-cut of a kernel differs from the plain form;
 * *d_mismatches (device, zeroed by the caller) += that.
 *   what = 0  the gradient kernel's short correctly rounded sqrt against __builtin_sqrtf
 *   what = 1  its one-instruction clamped R8_UNORM store
 *   what = 2  the ray set-up's reciprocal (v_rcp_f32 + refinement) against the IEEE division 1.0f / x
 *   what = 3  the ray set-up's quotient a / b through that reciprocal against the IEEE division, b = the pattern, a = a hash of it
 *   what = 4  the dispatch of that quotient for the numerators +0 and -0: they take the IEEE division (the refinement loses the sign of -0 / b) */
int vkv_debug_check(vkv_ctx *ctx, int32_t what, uint32_t first_bits, uint64_t count, uint64_t *d_mismatches, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VKVOLUME_AMD_DEBUG_H */
