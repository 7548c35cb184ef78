/*
 * vkv_oracle.h — CPU restatement (plain C99) of the reference's volume ray-caster hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker
 * or as the timed CPU baseline.  The product (vkvolume_amd/) never links or calls it.
 *
 * PARITY UNPINNED: the reference (LDeakin/VkVolume) has no tests, golden vectors or fixtures for
 * this path, and it cannot be built or run in this environment (GLSL + Vulkan + an un-vendored
 * Vulkan-Samples submodule).  This oracle is therefore pinned by
 *   (1) brute-force mathematical known-answer tests for the distance maps (tests/test_oracle_*.py),
 *   (2) closed-form checks (sphere entry/exit, ESS invariance, monotonicity), and
 *   (3) self-generated golden vectors under tests/golden/ (regression only).
 *
 * All struct types come from the public C ABI header so that parameter blocks are byte-identical
 * between the oracle and the HIP path; here every pointer is a HOST pointer.
 */
#ifndef VKV_ORACLE_H
#define VKV_ORACLE_H

#include "../include/vkvolume_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* src/volume_component.cpp:226-240 */
void vkvo_transfer_function_uniform(const VkvVolumeOptions *options, VkvTransferFunctionUniform *out);
/* src/volume_component.cpp:242-261 */
void vkvo_transfer_function_texture(const VkvVolumeOptions *options, uint8_t *out_rgba8_256x256);
/* src/volume_render_subpass.cpp:221-249, evaluated in double precision (independent check of the product's
 * float implementation; compare with a tolerance). */
void vkvo_build_uniforms(const float *view, const float *proj, const float *node_transform, const float *image_transform,
                         float clip_distance, uint32_t image_width, uint32_t image_height,
                         VkvExtent3D volume_extent, VkvExtent3D map_extent,
                         VkvCameraUniform *out_camera, VkvRayCastUniform *out_ray_cast, VkvRayGen *out_ray_gen);

/* shaders/gradient_map.comp:35-41 + shaders/get_gradient_compute.glsl:5-23 */
void vkvo_gradient_map(const uint8_t *volume, uint8_t *gradient, VkvExtent3D extent, const VkvTransferFunctionUniform *tf);
/* shaders/occupancy_map.comp:45-73 + src/compute_distance_map.cpp:106-113 */
void vkvo_occupancy_map(const uint8_t *volume, const uint8_t *gradient_or_null, const uint8_t *tf_rgba8,
                        const VkvTransferFunctionUniform *tf, VkvExtent3D extent, uint8_t *map, VkvExtent3D map_extent);
/* shaders/occupied_voxel_count.comp:28-56: voxels whose ANALYTIC transfer-function alpha is > 0 */
uint64_t vkvo_occupied_voxel_count(const uint8_t *volume, const uint8_t *gradient_or_null, const VkvTransferFunctionUniform *tf, VkvExtent3D extent);
/* shaders/distance_map.comp:44-109 with the dispatch order / aliasing of src/compute_distance_map.cpp:154-172 */
void vkvo_distance_map(uint8_t *map, uint8_t *swap, VkvExtent3D map_extent);
/* shaders/distance_map_anisotropic.comp:31-92 with the schedule of src/compute_distance_map.cpp:201-252 */
void vkvo_distance_map_anisotropic(uint8_t *const maps[8], uint8_t *swap, VkvExtent3D map_extent);
/* src/compute_distance_map.cpp:65-101 */
void vkvo_compute_distance_map(const uint8_t *volume, const uint8_t *gradient_or_null, const uint8_t *tf_rgba8,
                               const VkvTransferFunctionUniform *tf, VkvExtent3D extent, uint8_t *const maps[8],
                               uint8_t *swap, VkvExtent3D map_extent, int32_t skipping_type);

/* shaders/volume_render.frag:117-336 for every pixel of the tile schedule; n_threads >= 1 scanline workers.
 * pixel_stride > 1 renders only pixels with x % stride == 0 and y % stride == 0 (CPU-baseline sampling);
 * skipped pixels are left untouched.  Returns the number of rays marched. */
uint64_t vkvo_render(const VkvRenderParams *params, int n_threads, uint32_t pixel_stride);

/* diagnostics: the event sequence of one ray ('P' skip probe, 'O' probe that hit an occupied cell, 'S' empty sample,
 * 'A' sample with alpha > 0); returns the number of events */
uint32_t vkvo_trace_ray(const VkvRenderParams *params, int px, int py, uint8_t *events, uint32_t cap);
/* the same, also recording the loop index i (frag:215) at which every event happened */
uint32_t vkvo_trace_ray_steps(const VkvRenderParams *params, int px, int py, uint8_t *events, int32_t *steps, uint32_t cap);

/* Deterministic synthetic volumes. */
void vkvo_synth_volume(uint8_t *volume, VkvExtent3D extent, uint32_t kind, uint32_t seed);

/* src/load_volume.cpp:33-86 / :112-172.  Return 0 on success, VKV_E_IO / VKV_E_INVALID_ARGUMENT otherwise. */
typedef struct VkvoHeader
{
	VkvExtent3D extent;
	float       voxel_size[3];
	float       normalisation_range[2];
	char        type[16];
	char        endianness[16];
	float       image_transform[16];
} VkvoHeader;
int vkvo_load_header(const char *filename_header, VkvoHeader *out);
int vkvo_load_data(const char *filename_data, const VkvoHeader *header, uint8_t *out_volume);

#ifdef __cplusplus
}
#endif
#endif
