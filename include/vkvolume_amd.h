/*
 * vkvolume_amd.h — C ABI of the MI355X-native volume ray-caster hot path.
 *
 * This is the drop-in boundary: every entry point below replaces one
 * Vulkan dispatcher of the reference (LDeakin/VkVolume) and is what a maintainer's FFI
 * would bind.  Plain C types only: device pointers, sizes, POD structs, an opaque
 * context and a `void *stream` that is a `hipStream_t`.
 *
 * Conventions
 *   - every function returns 0 on success, a negative VKV_E_* code for argument /
 *     capability errors, or a positive `hipError_t` value when the HIP runtime failed;
 *   - device functions ENQUEUE on `stream` and do not synchronise (the reference's
 *     compute_submit() fence wait, src/volume_render.cpp:301-327, is the caller's job).
 *     They neither allocate device memory nor wait for the device once the SET-UP calls below
 *     (vkv_create, vkv_prepare_render, vkv_register_target; marked "set-up call") have seen the
 *     shapes, streams and targets they are used with: the small device tables a launch needs
 *     (tile start order, address tables, per-stream scratch) come out of an arena allocated by
 *     vkv_create and are uploaded asynchronously on the launch's stream from a pinned host copy;
 *     nothing is ever freed or re-used while a launch could still read it (only vkv_trim,
 *     vkv_forget_target, vkv_release_stream and vkv_destroy give device memory back, and they
 *     say what they wait for).  GROWTH: every distinct (frame size, tile schedule) and every
 *     distinct volume extent a context has rendered keeps one cached table (32 KiB for a
 *     1920x1080 frame of 16x16 tiles); the arena's table region (default 6 MiB of 8) holds
 *     ~190 such sizes.  When it is full, launches run without the table (plain tile order,
 *     address arithmetic in registers: the same bits, a few per cent slower) until vkv_trim
 *     empties it; the scratch blocks of streams have a region of their own (16 streams) and
 *     are never starved by tables;
 *   - pointers named `d_*` are device pointers owned by the caller; POD structs are
 *     passed by const pointer and copied at call time;
 *   - volumes are dense uint8, x fastest: index = (z*height + y)*width + x
 *     (raw file order, src/volume_component.cpp:47-52); distance / occupancy maps the same;
 *   - matrices are column-major float[16] (glm::mat4 byte layout).
 */
#ifndef VKVOLUME_AMD_H
#define VKVOLUME_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VKV_OK 0
#define VKV_E_INVALID_ARGUMENT (-1) /* null pointer, zero extent, bad enum              */
#define VKV_E_UNSUPPORTED (-2)      /* e.g. map axis > 2048, volume too large for a launch */
#define VKV_E_NO_DEVICE (-3)        /* no HIP device / wrong architecture               */
#define VKV_E_IO (-4)               /* file errors of the loader                        */

#define VKV_TF_BITS_WORDS 2564      /* uint32 words of a vkv_transfer_function_tables() / _bits() buffer */

/* VkExtent3D stand-in (src/load_volume.h:31, src/volume_component.cpp:91-92). */
typedef struct VkvExtent3D
{
	uint32_t width, height, depth;
} VkvExtent3D;

/* TransferFunctionUniform, src/transfer_function.h:20-32 (32 bytes; the texture-path shaders read
 * the first 16, shaders/transfer_function.glsl:18-29). use_gradient is a VkBool32. */
typedef struct VkvTransferFunctionUniform
{
	float    sampling_factor;
	float    voxel_alpha_factor;
	float    grad_magnitude_modifier;
	uint32_t use_gradient;
	float    intensity_min;
	float    intensity_range_inv;
	float    gradient_min;
	float    gradient_range_inv;
} VkvTransferFunctionUniform;

/* Volume::Options, src/volume_component.h:45-56. */
typedef struct VkvVolumeOptions
{
	float    sampling_factor;          /* 1.0 */
	float    voxel_alpha_factor;       /* 1.0 */
	uint32_t use_precomputed_gradient; /* true */
	float    intensity_min;            /* 0.0 */
	float    intensity_max;            /* 1.0 */
	float    gradient_min;             /* 0.0 */
	float    gradient_max;             /* 1.0 */
} VkvVolumeOptions;

/* CameraUniform, src/volume_render_subpass.h:32-39 (320 bytes). */
typedef struct VkvCameraUniform
{
	float camera_view[16];
	float camera_proj[16];
	float camera_view_proj_inv[16];
	float model[16];
	float model_inv[16];
} VkvCameraUniform;

/* RayCastUniform, src/volume_render_subpass.h:46-53 (68 bytes). */
typedef struct VkvRayCastUniform
{
	float   plane[4];          /* clipping plane, global coordinates  */
	float   plane_tex[4];      /* clipping plane, texture coordinates */
	float   camera_pos_tex[4]; /* camera position, texture coordinates */
	float   block_size[4];     /* ceil(volume extent / map extent) per axis, as floats */
	int32_t front_index;
} VkvRayCastUniform;

/* Analytic pinhole ray generator.  It stands in for the rasteriser + the two vertex shaders
 * (shaders/volume_render_clipped.vert:50-65, shaders/volume_render_plane_intersection.vert:101-130):
 * the un-normalised texture-space direction of the ray through pixel (px, py) is
 *   dir = dir00 + (px + 0.5) * ddx + (py + 0.5) * ddy
 * which is exact for a pinhole camera.  vkv_build_uniforms() fills it from the matrices. */
typedef struct VkvRayGen
{
	float dir00[4];
	float ddx[4];
	float ddy[4];
} VkvRayGen;

/* VolumeRenderSubpass::SkippingType / Test / Options, src/volume_render_subpass.h:58-81. */
enum VkvSkippingType
{
	VKV_SKIP_NONE                 = 0,
	VKV_SKIP_BLOCK                = 1,
	VKV_SKIP_DISTANCE             = 2,
	VKV_SKIP_ANISOTROPIC_DISTANCE = 3
};

enum VkvTest
{
	VKV_TEST_NONE                = 0,
	VKV_TEST_RAY_ENTRY           = 1,
	VKV_TEST_RAY_EXIT            = 2,
	VKV_TEST_NUM_TEXTURE_SAMPLES = 3
};

typedef struct VkvRenderOptions
{
	int32_t skipping_type;         /* VKV_SKIP_DISTANCE */
	float   clip_distance;         /* 50.0 */
	int32_t early_ray_termination; /* true */
	int32_t depth_attachment;      /* false; true = DEPTH_ATTACHMENT variant: needs VkvRenderParams.d_in_depth */
	int32_t test;                  /* VKV_TEST_NONE */
} VkvRenderOptions;

/* A rectangle of screen tiles: x0, y0 = its first tile column / row, w x h tiles.  vkv_screen_tile_rect() derives the rectangle a frame's
 * fragments can lie in from the uniforms alone, so every rank of a multi-GPU frame arrives at the same one without talking. */
typedef struct VkvTileRect
{
	uint32_t x0, y0, w, h;
} VkvTileRect;

/* Screen tiling of one launch (single GPU: all tiles; multi GPU: every tile_stride-th tile).
 * The W×H image is cut into tiles of tile_width × tile_height pixels.  The SCHEDULED tiles are those of `rect` (rect.w == 0 or rect.h == 0:
 * every tile of the image), numbered row-major INSIDE the rectangle: tile t is column rect.x0 + t % rect.w, row rect.y0 + t / rect.w.
 * The launch renders tiles tile_first + k*tile_stride, k = 0 .. tile_count-1.  Pixels outside the rectangle are not touched by the launch
 * (vkv_scatter_tiles / vkv_assemble_frames clear them when they assemble a frame) unless fill_outside is set (single-GPU frames).
 * compact == 0: outputs are indexed by image pixel  (y*image_width + x);
 * compact != 0: outputs are indexed by (k*tile_height + ly)*tile_width + lx  (the per-rank
 *               buffer that is gathered over RCCL and de-interleaved by vkv_scatter_tiles); the
 *               slots of a partial tile's pixels beyond the image edge are never written.
 * The reference's rasteriser only shades fragments of the clipped box's faces (src/volume_render_subpass.cpp:262-293,
 * shaders/volume_render_clipped.vert:50-65): the rectangle is this build's counterpart - tiles no fragment can lie in are neither
 * scheduled nor exchanged. */
typedef struct VkvTileSchedule
{
	uint32_t    tile_width, tile_height; /* multiples of 16 (one 256-thread workgroup marches 16x16 pixels) */
	uint32_t    tile_first, tile_stride, tile_count;
	uint32_t    compact;
	VkvTileRect rect;                    /* all zero: the whole image */
	uint32_t    fill_outside;            /* single-GPU frames through a rectangle: != 0 = the launch ALSO writes the "no fragment" result (what a pixel whose ray
	                                        misses the box gets: clear values, or nothing with blend_over_target) to every pixel outside the rectangle, so the
	                                        frame is complete as with the whole-image schedule - without a workgroup per empty tile (each rendering workgroup
	                                        fills a share of the outside tiles).  Needs compact == 0 and the whole rectangle in this launch (tile_first 0,
	                                        tile_stride 1, tile_count rect.w * rect.h); the result equals the whole-image schedule's bit for bit. */
} VkvTileSchedule;

/* Everything VolumeRenderSubpass::draw binds for one volume
 * (src/volume_render_subpass.cpp:219-293) plus the output images. */
typedef struct VkvRenderParams
{
	VkvCameraUniform           camera;
	VkvRayCastUniform          ray_cast;
	VkvTransferFunctionUniform transfer_function;
	VkvRayGen                  ray_gen;
	VkvRenderOptions           options;
	uint32_t                   use_precomputed_gradient; /* PRECOMPUTED_GRADIENT variant */
	uint32_t                   image_width, image_height;
	VkvTileSchedule            tiles;
	VkvExtent3D                volume_extent;
	VkvExtent3D                map_extent;
	const uint8_t *            d_volume;                   /* R8_UNORM  W*H*D                */
	const uint8_t *            d_gradient;                 /* R8_UNORM  W*H*D, may be NULL when !use_precomputed_gradient */
	const uint8_t *            d_transfer_function;        /* RGBA8 256x256, row = gradient  */
	const uint8_t *            d_distance_maps[8];         /* [0] (or [0..7] anisotropic); unused for VKV_SKIP_NONE */
	const void *               d_packed_volume;            /* optional: vkv_pack_volume() image of d_volume (+ d_gradient); NULL = sample the linear buffers */
	const uint32_t *           d_transfer_function_bits;   /* optional: vkv_transfer_function_tables() of d_transfer_function; NULL = fetch the texel */
	float *                    d_out_color;                /* RGBA32F premultiplied, or NULL */
	uint8_t *                  d_out_rgba8;                /* RGBA8 round-to-nearest of the above, or NULL */
	uint32_t *                 d_out_counts;               /* 3 x u32 per pixel: volume samples, distance probes, empty samples; or NULL (the integrator then does not count) */
	float *                    d_out_depth;                /* gl_FragDepth (reverse-Z, 0 = far), or NULL */
	const float *              d_in_depth;                 /* options.depth_attachment: the scene depth the subpass reads as input attachment 0
	                                                          (frag:26, 122-165; reverse-Z), indexed like the outputs */
	uint32_t                   blend_over_target;          /* != 0: d_out_color / d_out_rgba8 hold the destination colour and the fragment is blended
	                                                          onto it with the subpass's blend state (volume_render_subpass.cpp:176-190):
	                                                          rgb = src + (1 - src.a) * dst, a = src.a * (1 - src.a); pixels without a fragment stay
	                                                          untouched.  0: every pixel of the schedule is overwritten (cleared to 0 first). */
} VkvRenderParams;

typedef struct vkv_ctx vkv_ctx;

/* ---- context ------------------------------------------------------------------------------- */
/* set-up call: allocates the context's device arena (VkvTuning.arena_bytes of the environment default, 8 MiB). */
int         vkv_create(int device_ordinal, vkv_ctx **out_ctx);
/* set-up call: waits for the device, then frees everything the context owns. */
void        vkv_destroy(vkv_ctx *ctx);
const char *vkv_last_error(const vkv_ctx *ctx);
const char *vkv_version(void);

/* Tuning switches of one context (A/B switches of the launchers; every setting renders the same bits).  vkv_create fills them from the
 * environment variables named below, ONCE; afterwards the environment is not consulted again: the behaviour of a linked library depends
 * on its context, not on the host's environment at first use.  vkv_set_tuning replaces the whole block (read it with vkv_get_tuning,
 * change fields, write it back); it applies to calls made after it returns. */
typedef struct VkvTuning
{
	uint32_t struct_size;              /* sizeof(VkvTuning): set by vkv_get_tuning, checked by vkv_set_tuning                       */
	int32_t  scheduler;                /* 0 lane = ray on static tiles (default); 1 persistent waves with lane re-fill   VKV_RAYMARCH_SCHEDULER=persistent */
	int32_t  batch_mode;               /* vkv_render_batch: 0 workgroup per tile (default); 1 resident workgroups pulling 8x8 units   VKV_RAYMARCH_BATCH=pull */
	int32_t  batch_sequential;         /* vkv_render_batch: 1 = frames one after the other instead of interleaved       VKV_RAYMARCH_BATCH_ORDER=sequential */
	int32_t  tile_order_linear;        /* 1 = tiles start in schedule order instead of centre-of-image first             VKV_RAYMARCH_TILE_ORDER=linear */
	int32_t  address_tables;           /* packed image: 0 none, 1 two-level LDS tables, 2 + one entry per voxel index (default)   VKV_RAYMARCH_LUT=0|2(two-level)|1 */
	uint32_t full_table_lds_limit;     /* LDS bytes per workgroup up to which the per-voxel tables are used (17920)       VKV_RAYMARCH_FULL_LIMIT */
	int32_t  screen_cull;              /* 1 = pixels outside the screen bound of the volume's box skip the ray set-up (default)   VKV_RAYMARCH_CULL=0 */
	int32_t  feedback;                 /* 1 = registered targets start their tiles in the order their last measured frame suggests (default)   VKV_RAYMARCH_FEEDBACK=0 */
	uint32_t feedback_period;          /* frames between two cost measurements of a target (8)                            VKV_RAYMARCH_FEEDBACK_PERIOD */
	int32_t  clamp_always;             /* 1 = the march loop never takes its clamp-free iterations (A/B switch, same bits)   VKV_RAYMARCH_CLAMP=always  */
	float    tile_mix_heavy;           /* experiment: central share of the tiles spread over the first tile_mix_spread of the order (0 = off)   VKV_RAYMARCH_TILE_MIX=h,s */
	float    tile_mix_spread;
	uint32_t gradient_segment;         /* vkv_gradient_map: tiles a workgroup marches in z; 0 = automatic                 VKV_GRADIENT_SEGMENT */
	int32_t  pack_tile;                /* vkv_pack_volume: 0 automatic, 2 / 4 = bricks per workgroup edge                 VKV_PACK_TILE */
	int32_t  wave_shape;               /* width in pixels of a wave's 64-pixel patch: 0 automatic (4, 8 or 16 from the view: the shape that is most compact
	                                      in voxels), or 4 / 8 / 16 (A/B switch, same bits)                                 VKV_RAYMARCH_WAVE_SHAPE */
	int32_t  occupancy_kernel;         /* vkv_occupancy_map: 0 automatic (k_occupancy_map_waves: a wave per span of 64 dwords, every block width), 1 = the
	                                      workgroup-per-cell-row kernels of rounds 1-4 (A/B switch, same map)                                     VKV_OCCUPANCY_KERNEL=rows */
	uint32_t arena_bytes;              /* read-only: size of the device arena vkv_create allocated                        VKV_ARENA_BYTES */
} VkvTuning;
int vkv_get_tuning(const vkv_ctx *ctx, VkvTuning *out);
int vkv_set_tuning(vkv_ctx *ctx, const VkvTuning *tuning);

/* ---- host-side helpers (pure CPU, no device access) ------------------------------------------ */

/* Volume::get_transfer_function_uniform, src/volume_component.cpp:226-240. */
int vkv_transfer_function_uniform(const VkvVolumeOptions *options, VkvTransferFunctionUniform *out);

/* CPU half of Volume::update_transfer_function_texture, src/volume_component.cpp:242-261:
 * fills 256*256 RGBA8 texels (row = gradient, column = intensity). */
int vkv_transfer_function_texture(const VkvVolumeOptions *options, uint8_t *out_rgba8_256x256);

/* Uniform maths of VolumeRenderSubpass::draw, src/volume_render_subpass.cpp:221-249, plus the ray
 * generator.  view / proj (already vulkan-style, y flipped) / node_transform / image_transform are
 * column-major 4x4. */
int vkv_build_uniforms(const float *view, const float *proj, const float *node_transform, const float *image_transform,
                       float clip_distance, uint32_t image_width, uint32_t image_height,
                       VkvExtent3D volume_extent, VkvExtent3D map_extent,
                       VkvCameraUniform *out_camera, VkvRayCastUniform *out_ray_cast, VkvRayGen *out_ray_gen);

/* ---- device entry points ------------------------------------------------------------------- */

/* ComputeGradientMap::compute, src/compute_gradient_map.cpp:57-81 (shaders/gradient_map.comp). */
int vkv_gradient_map(vkv_ctx *ctx, const uint8_t *d_volume, uint8_t *d_gradient, VkvExtent3D extent,
                     const VkvTransferFunctionUniform *tf, void *stream);

/* ComputeDistanceMap::computeOccupancy, src/compute_distance_map.cpp:103-140
 * (shaders/occupancy_map.comp).  d_gradient == NULL selects the on-the-fly gradient variant. */
int vkv_occupancy_map(vkv_ctx *ctx, const uint8_t *d_volume, const uint8_t *d_gradient,
                      const uint8_t *d_transfer_function, const VkvTransferFunctionUniform *tf,
                      VkvExtent3D extent, uint8_t *d_map, VkvExtent3D map_extent, void *stream);

/* ComputeDistanceMap::computeDistance, src/compute_distance_map.cpp:142-175
 * (shaders/distance_map.comp): in-place on d_map (holding the occupancy map), d_swap is scratch. */
int vkv_distance_map(vkv_ctx *ctx, uint8_t *d_map, uint8_t *d_swap, VkvExtent3D map_extent, void *stream);

/* ComputeDistanceMap::computeDistanceAnisotropic, src/compute_distance_map.cpp:177-290
 * (shaders/distance_map_anisotropic.comp): occupancy in d_maps[7]; on return d_maps[k] is the map
 * for ray-direction octant k = (dz<0) + 2(dy<0) + 4(dx<0). */
int vkv_distance_map_anisotropic(vkv_ctx *ctx, uint8_t *const d_maps[8], uint8_t *d_swap,
                                 VkvExtent3D map_extent, void *stream);

/* ComputeDistanceMap::compute, src/compute_distance_map.cpp:65-101: occupancy into
 * d_maps[n-1] (n = 8 for anisotropic, else 1), then the transform selected by skipping_type. */
int vkv_compute_distance_map(vkv_ctx *ctx, const uint8_t *d_volume, const uint8_t *d_gradient,
                             const uint8_t *d_transfer_function, const VkvTransferFunctionUniform *tf,
                             VkvExtent3D extent, uint8_t *const d_maps[8], uint8_t *d_swap,
                             VkvExtent3D map_extent, int32_t skipping_type, void *stream);

/* ComputeOccupiedVoxelCount::compute + get_result, src/compute_occupied_voxel_count.cpp:88-156
 * (shaders/occupied_voxel_count.comp + occupied_voxel_count_reduce.comp): number of voxels whose ANALYTIC
 * transfer-function alpha (the uniform's min / range_inv fields, not the texture) is > 0, written to *d_count.
 * One pass: wave ballot + one 64-bit atomic per workgroup instead of the reference's multi-dispatch tree reduce. */
int vkv_occupied_voxel_count(vkv_ctx *ctx, const uint8_t *d_volume, const uint8_t *d_gradient, const VkvTransferFunctionUniform *tf,
                             VkvExtent3D extent, uint64_t *d_count, void *stream);

/* LoadVolume::load_header / load_data, src/load_volume.cpp:33-86, :112-172 (host side). */
typedef struct VkvVolumeHeader
{
	VkvExtent3D extent;
	float       voxel_size[3];
	float       normalisation_range[2];
	char        type[16];
	char        endianness[16];
	float       image_transform[16];
} VkvVolumeHeader;
int vkv_load_header(const char *filename_header, VkvVolumeHeader *out);
int vkv_load_data(const char *filename_data, const VkvVolumeHeader *header, uint8_t *out_voxels, size_t out_bytes);

/* Device half of LoadVolume::load_data_impl (src/load_volume.cpp:151-169) for volumes whose raw file has been uploaded as
 * is: endianness conversion and normalisation to uint8, `(uint8) (255 * max(0, min(1, (v - min) / (max - min))))`, truncating.
 * d_raw holds n_voxels elements of `type`. */
enum VkvVoxelType
{
	VKV_VOXEL_UINT8  = 0,
	VKV_VOXEL_INT8   = 1,
	VKV_VOXEL_UINT16 = 2,
	VKV_VOXEL_INT16  = 3
};
int vkv_convert_volume(vkv_ctx *ctx, const void *d_raw, int32_t type, int32_t big_endian, float range_min, float range_max, uint64_t n_voxels,
                       uint8_t *d_out, void *stream);

/* Device-internal sampling layout of the volume (the counterpart of uploading into a VK_IMAGE_TILING_OPTIMAL image,
 * src/volume_component.cpp:68-83): 4x4x4-voxel bricks with a one-voxel apron, volume and gradient bytes interleaved,
 * clamp-to-edge baked in, bricks grouped 8x8x8.  Every trilinear footprint of both textures then lies inside one
 * 256-byte brick.  vkv_packed_volume_bytes() sizes the buffer; d_gradient may be NULL (gradient channel = 0).
 * Results of vkv_render are bit-identical with and without it. */
size_t vkv_packed_volume_bytes(VkvExtent3D extent);
int    vkv_pack_volume(vkv_ctx *ctx, const uint8_t *d_volume, const uint8_t *d_gradient, VkvExtent3D extent, void *d_packed, void *stream);

/* Acceleration tables of the 256x256 TF texture for the integrator (device half of Volume::update_transfer_function_texture,
 * src/volume_component.cpp:262-278, next to the texture upload).  The buffer is VKV_TF_BITS_WORDS uint32:
 *   words [0, 2048)  1 bit per texel: alpha > 0 (row = gradient), so "voxel_occupied" (frag:276) comes from LDS and only occupied
 *                    samples fetch the RGBA texel;
 *   word 2048        flags; bit 0 = "separable greyscale": every texel equals (b, b, b, b) with
 *                    b = (uint8) clamp(alpha_i[column] * alpha_g[row] * 255, 0, 255) — the product the reference always builds
 *                    (src/volume_component.cpp:246-261).  alpha_i / alpha_g are derived from `tf` (intensity_min, *_range_inv,
 *                    use_gradient) and the claim is CHECKED on the device against all 65536 texels; when it holds the integrator
 *                    takes the texel from two 256-entry LDS tables (words 2052.., float) instead of a dependent global fetch.
 * tf == NULL (or vkv_transfer_function_bits) leaves the flag clear: any RGBA texture works through the generic path.
 * Rebuild the tables whenever the texture changes. */
int vkv_transfer_function_tables(vkv_ctx *ctx, const uint8_t *d_transfer_function, const VkvTransferFunctionUniform *tf,
                                 uint32_t *d_tables, void *stream);
int vkv_transfer_function_bits(vkv_ctx *ctx, const uint8_t *d_transfer_function, uint32_t *d_tables, void *stream);

/* VolumeRenderSubpass::prepare, src/volume_render_subpass.cpp:95-157 (where the reference builds its pipelines and descriptor layouts).
 * Set-up call: creates, for `count` parameter blocks as a later vkv_render / vkv_render_batch on `stream` will pass them, everything that
 * launch takes from the context: the stream's scratch block, the address tables of the packed image's extent, the tile start order of the
 * schedule - and from the HIP runtime: the code objects of the kernels those blocks select are loaded onto the device now (the runtime
 * loads a code object at the first use of one of its kernels, which allocates device memory and takes milliseconds); uploads are
 * waited for before it returns.  A launch that finds one of them missing still creates it on the fly out of the
 * arena with an asynchronous upload on its own stream (no device-wide wait; if the arena is exhausted the launch runs without the table:
 * plain tile order / address arithmetic in registers, same bits) - vkv_prepare_render only moves that work to set-up time.  Call it
 * before capturing `stream` into a hipGraph: a launch that still has to create a table records and queries an event, which a capture
 * does not allow.  Captured launches replay with the parameter blocks they were captured with (camera included: capture one graph per
 * view, or re-capture).  A captured vkv_render_batch owns a slot of the context - a pinned host block (the graph's copy node reads its
 * source at every replay) and a device block of its own (the copy's target and the kernels' argument pointer) - so graphs may be replayed on
 * any stream, next to each other and next to live launches.  32 slots are set aside by vkv_create (nothing is allocated during such a
 * capture), later ones are allocated during the capture (2 x 96 KiB, with the thread's capture mode relaxed for the calls).  A slot stays
 * with the stream it was captured on until vkv_release_captured(stream); vkv_trim and vkv_destroy free the slots and the tables the
 * captured launches point to: graphs captured before either call must not be launched after it. */
int vkv_prepare_render(vkv_ctx *ctx, const VkvRenderParams *params, uint32_t count, void *stream);

/* The graphs captured on `stream` so far have been destroyed (or will not be launched again): the slots their vkv_render_batch launches
 * own return to the context.  A renderer that re-captures when its camera moves calls this after hipGraphExecDestroy and before the next
 * capture; without it every capture takes new slots until vkv_trim.  Does not wait and frees nothing. */
int vkv_release_captured(vkv_ctx *ctx, void *stream);

/* Start-order feedback needs device state per render target (one uint32 cost and one uint32 order entry per tile of the
 * schedule): a renderer draws into the same swap-chain images again and again with a camera that moves little, so the tiles that took longest in the
 * last measured frame are started first in the next ones (any order renders the same bits).
 * vkv_register_target - set-up call: allocates and initialises that state for frames of image_width x image_height pixels rendered with
 *   tile schedule `tiles` into `d_target` (the d_out_rgba8 or, without one, d_out_color pointer of the parameter block).  Targets that
 *   were never registered are rendered in the centre-first order: vkv_render never allocates.  Registering a target again replaces
 *   its state (waits for the device first, like vkv_forget_target).
 * vkv_forget_target - set-up call: waits for the device (launches that still use the state), then frees it.  Like freeing the target
 *   itself, it must not run concurrently with a render into that target from another thread. */
int vkv_register_target(vkv_ctx *ctx, const void *d_target, uint32_t image_width, uint32_t image_height, const VkvTileSchedule *tiles);
int vkv_forget_target(vkv_ctx *ctx, const void *d_target);

/* Set-up call: waits for the device, then drops every cached table (tile start orders, address tables) and empties the arena's table
 * region; the next launches create what they need again (asynchronously, as on first use).  For a renderer whose window or volume sizes
 * keep changing: call it at a quiet point (a resize, a scene change) - like vkv_forget_target it must not run concurrently with a launch
 * from another thread.  Stream scratch blocks and registered targets are not touched.  Also returns the pinned argument blocks of launches
 * captured into hipGraphs (vkv_prepare_render): a graph captured before the call points to tables and blocks that are gone - re-capture. */
int vkv_trim(vkv_ctx *ctx);

/* Gives the 128 KiB scratch block vkv_render_batch / vkv_compute_distance_map / ... keep per HIP stream back to the context's pool.
 * Call it before destroying a stream that was handed to this context, when all work enqueued on it has completed (it does not wait). */
int vkv_release_stream(vkv_ctx *ctx, void *stream);

/* VolumeRenderSubpass::draw, src/volume_render_subpass.cpp:159-294 (shaders/volume_render.frag). */
int vkv_render(vkv_ctx *ctx, const VkvRenderParams *params, void *stream);

/* Several frames in ONE launch: `count` (1 .. VKV_MAX_BATCH) parameter blocks that share the kernel variant (skipping type, ERT,
 * gradient mode, test mode), have a packed sampling image and the same tile SIZE (the tile counts and rectangles of the schedules may
 * differ: the frames of a multi-GPU launch each have their own screen rectangle); cameras, volumes and output buffers may differ (stereo pairs, orbit sweeps, the per-rank tile sets of a multi-GPU frame, the reference's frames in flight).  The
 * frames advance side by side inside one grid, so the long tail of each is covered by the bulk of the others without relying on
 * several hardware queues.  Output buffers of different frames must not overlap.  Results are bit-identical to `count` vkv_render calls. */
#define VKV_MAX_BATCH 32
int vkv_render_batch(vkv_ctx *ctx, const VkvRenderParams *params, uint32_t count, void *stream);

/* The tile rectangle the fragments of a frame can lie in (pure CPU, double precision from the float uniforms: the same inputs give the
 * same rectangle on every rank).  The volume's box [0,1]^3 (texture space) is cut by the clipping plane (plane_tex: the kept side is
 * dot(plane_tex.xyz, p) + plane_tex.w >= 0, the side the integrator starts its rays on, frag:117 / volume_render_clipped.vert:56), the
 * vertices of the clipped box are solved through the ray generator (p - camera_pos_tex = g (dir00 + fx ddx + fy ddy)), and the pixel
 * bound of the (fx, fy), widened by two pixels, is rounded outwards to whole tiles and then to multiples of `align_tiles` tiles
 * (0 or 1: no alignment; a renderer whose camera moves aligns to e.g. 4 tiles so that the rectangle - and with it the feedback state of a
 * registered target - changes less often; a schedule over a whole rectangle computes its start order in the kernel, no table is cached per rectangle).  A pixel outside the rectangle cannot have a fragment, whatever
 * the depth test does afterwards.  The rectangle is never empty: a box that is off screen gives the 1 x 1 rectangle of tile (0, 0)
 * (a fixed, minimal exchange); a vertex at or behind the camera plane, or a degenerate generator, gives the whole image. */
int vkv_screen_tile_rect(const VkvRayCastUniform *ray_cast, const VkvRayGen *ray_gen, uint32_t image_width, uint32_t image_height,
                         uint32_t tile_width, uint32_t tile_height, uint32_t align_tiles, VkvTileRect *out_rect);

/* Root-rank de-interleave of gathered compact tile buffers into the W×H image (multi-GPU): the tiles of `rect` (NULL or w == 0: every
 * tile of the image; numbered row-major inside it as in VkvTileSchedule) were dealt round-robin to n_ranks ranks, tile t to rank
 * t % n_ranks as its (t / n_ranks)-th.  d_gathered holds the n_ranks compact buffers, `rank_stride_tiles` tiles apart (>= the
 * ceil(tiles / n_ranks) tiles a rank holds: when the frames of a launch travel as one block per rank the stride is the block's tile
 * count and the caller offsets d_gathered to the frame), bytes_per_pixel per pixel.  Pixels outside the rectangle are CLEARED to zero:
 * the image is complete after the call. */
int vkv_scatter_tiles(vkv_ctx *ctx, const void *d_gathered, void *d_image, uint32_t image_width,
                      uint32_t image_height, uint32_t tile_width, uint32_t tile_height, const VkvTileRect *rect,
                      uint32_t n_ranks, uint32_t rank_stride_tiles, uint32_t bytes_per_pixel, void *stream);

/* ---- multi-GPU exchange step --------------------------------------------------------------------------------
 * Rays are independent: a frame is cut into screen tiles dealt round-robin to the ranks (VkvTileSchedule: tile_first = rank,
 * tile_stride = n_ranks, compact = 1, rect = vkv_screen_tile_rect of the frame's uniforms), the volume is replicated, and the ONLY
 * exchange is the gather of every rank's compact tile buffer to the frame's owner, followed by vkv_scatter_tiles there.  One process (or
 * host thread) per GPU, one ncclComm_t each.
 *
 * vkv_gather_tiles enqueues that gather on `stream` with RCCL's ncclGather (rccl.h:745; 7 concurrent point-to-point transfers into
 * the root over xGMI): every rank sends bytes_per_rank bytes from d_tiles, the root receives n_ranks * bytes_per_rank bytes into
 * d_gathered (ignored elsewhere).  `nccl_comm` is the caller's ncclComm_t.  The RCCL library is resolved at run time: the copy
 * already loaded into the process if there is one (so the communicator and the call come from the same library), else librccl.so.1;
 * VKV_RCCL_LIBRARY=<path> overrides.  Returns VKV_E_UNSUPPORTED when no RCCL can be loaded, 1000 + ncclResult_t on RCCL errors. */
int vkv_gather_tiles(vkv_ctx *ctx, const void *d_tiles, void *d_gathered, size_t bytes_per_rank, int32_t root, void *nccl_comm, void *stream);

/* The whole exchange of one frame on `stream`: vkv_gather_tiles of this rank's ceil(tiles of rect / n_ranks) compact tiles to `root`, then
 * (on the root only) vkv_scatter_tiles of the gathered buffers into d_image.  d_gathered is scratch of n_ranks * ceil(tiles / n_ranks) *
 * tile_width * tile_height * bytes_per_pixel bytes on the root; d_image / d_gathered may be NULL on the other ranks. */
int vkv_assemble_frame(vkv_ctx *ctx, const void *d_tiles, void *d_gathered, void *d_image, uint32_t image_width, uint32_t image_height,
                       uint32_t tile_width, uint32_t tile_height, const VkvTileRect *rect, uint32_t n_ranks, uint32_t rank,
                       uint32_t bytes_per_pixel, int32_t root, void *nccl_comm, void *stream);

/* The exchange of a whole vkv_render_batch launch (what a C or C++ renderer binds for frames in flight).  Frame f has its own tile
 * rectangle rects[f] (NULL: every frame the whole image) and its own owner roots[f] (NULL: every frame `root`): with the owners of a
 * launch's frames spread over the ranks every GPU receives at once, over all of its inbound xGMI links, instead of one GPU receiving
 * everything.  tpr(f) = ceil(tiles of rects[f] / n_ranks) is what a rank renders of frame f (its VkvTileSchedule.tile_count, except on
 * the last ranks of a ragged deal, whose missing tile's slot travels unused).
 *   d_tiles     this rank's compact tiles of the launch, frames back to back: [frame f][tpr(f) tiles];
 *   d_gathered  scratch of n_ranks * sum_f tpr(f) tiles on every rank that owns a frame (its layout is the call's own business: [rank][frame]
 *               [tiles] after the one gather, [frame][rank][tiles] after the group);
 *   d_images    host array of `frames` device pointers (copied at call time); entry f is only read on the owner of frame f.
 * roots == NULL: ONE ncclGather of the launch's whole block to `root`.  roots != NULL: one ncclGather per frame, all inside ONE group
 * (ncclGroupStart / ncclGroupEnd: RCCL fuses the group's point-to-point transfers into one kernel).  Then, on every rank that owns a frame,
 * ONE de-interleave kernel for the frames it owns (pixels outside a frame's rectangle are cleared).
 * Every rank must pass the same frames, rects, roots. */
int vkv_assemble_frames(vkv_ctx *ctx, const void *d_tiles, void *d_gathered, void *const *d_images, uint32_t frames, uint32_t image_width,
                        uint32_t image_height, uint32_t tile_width, uint32_t tile_height, const VkvTileRect *rects, uint32_t n_ranks, uint32_t rank,
                        uint32_t bytes_per_pixel, int32_t root, const int32_t *roots, void *nccl_comm, void *stream);

/* Deterministic synthetic uint8 volume, generated on the device. kind 0 = soft
 * sphere (config C1), kind 1 = ellipsoid shells + hash noise (configs C2..C5).  The shells take three knobs in the upper bits of `kind`
 * (kind = 1 | shells << 8 | thickness << 16 | noise << 28): only the first `shells` (1 .. 39; 0 = all 40) of the seed's shells, their
 * thickness scaled by thickness / 256 (12 bits; 0 = 1.0), the hash noise 0 .. noise (4 bits; 0 = the default 0 .. 20) -
 * tools/benchmark_sweep.py tunes the occupied-voxel share of its scenes to the reference's datasets with them. */
int vkv_synth_volume(vkv_ctx *ctx, uint8_t *d_volume, VkvExtent3D extent, uint32_t kind, uint32_t seed, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VKVOLUME_AMD_H */
