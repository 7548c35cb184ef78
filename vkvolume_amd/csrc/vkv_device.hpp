// vkv_device.hpp — shared device-side definitions for the gfx950 kernels.
//
// The arithmetic below is the product's definition of the numerics the reference leaves to the Vulkan
// implementation (DESIGN.md "Pinned numerics"): fp32 only, no implicit contraction (the library is
// built with -ffp-contract=off), fused multiply-adds only where __builtin_fmaf is written out,
// IEEE-correct division and square root (-fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/vkvolume_amd.h"

namespace vkv
{

// GLSL built-ins with specification semantics (min(x,y) = y<x ? y : x, ...).
__device__ __forceinline__ float g_min(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float g_max(float x, float y) { return (x < y) ? y : x; }
__device__ __forceinline__ float g_clamp(float x, float lo, float hi) { return g_min(g_max(x, lo), hi); }
__device__ __forceinline__ float g_step(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }
__device__ __forceinline__ float g_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
__device__ __forceinline__ int   i_clamp(int x, int lo, int hi) { return min(max(x, lo), hi); }

constexpr float kInv255 = 1.0f / 255.0f;

__device__ __forceinline__ size_t vidx(int x, int y, int z, int W, int H)
{
	return ((size_t) z * (size_t) H + (size_t) y) * (size_t) W + (size_t) x;
}

// R8_UNORM texel -> float: exactly byte / 255 (IEEE division).
__device__ __forceinline__ float unorm8(uint32_t b) { return (float) b / 255.0f; }

// NEAREST lookup coordinate of the 256-wide TF texture (sampler: src/volume_component.cpp:149-151).
__device__ __forceinline__ int tf_texel(float u) { return i_clamp((int) __builtin_floorf(u * 256.0f), 0, 255); }

// Tetrahedron gradient on integer texels, shaders/get_gradient_compute.glsl:12-20; returns the float in [0,1]
// before the UNORM store.  Term order k.xyy, k.yyx, k.yxy, k.xxx, sums left to right.
__device__ __forceinline__ float gradient_from_taps(float v1, float v2, float v3, float v4, float modifier)
{
	const float gx  = 0.25f * (((v1 - v2) - v3) + v4);
	const float gy  = 0.25f * (((-v1 - v2) + v3) + v4);
	const float gz  = 0.25f * (((-v1 + v2) - v3) + v4);
	const float len = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz);
	return g_clamp(len * modifier, 0.0f, 1.0f);
}

__device__ __forceinline__ float gradient_on_the_fly(const uint8_t *__restrict__ vol, int W, int H, int D, int x, int y, int z, float modifier)
{
	const int   xm = max(x - 1, 0), xp = min(x + 1, W - 1);
	const int   ym = max(y - 1, 0), yp = min(y + 1, H - 1);
	const int   zm = max(z - 1, 0), zp = min(z + 1, D - 1);
	const float v1 = unorm8(vol[vidx(xp, ym, zm, W, H)]);
	const float v2 = unorm8(vol[vidx(xm, ym, zp, W, H)]);
	const float v3 = unorm8(vol[vidx(xm, yp, zm, W, H)]);
	const float v4 = unorm8(vol[vidx(xp, yp, zp, W, H)]);
	return gradient_from_taps(v1, v2, v3, v4, modifier);
}

// ---- correctly rounded fp32 division out of v_rcp_f32 (the ray set-up's 22 divisions per ray; see raymarch_core.hpp, ray_setup) ----
__device__ __forceinline__ bool div_ordinary(float x)
{
	const float m = __builtin_fabsf(x);
	return m >= 0x1p-40f && m <= 0x1p40f;        // false for 0, denormals, huge values, inf and NaN
}
__device__ __forceinline__ bool div_ordinary_num(float x)
{        // (a zero numerator is NOT ordinary: the refinement loses the sign of -0 / d)
	return div_ordinary(x);
}
__device__ __forceinline__ float recip_refined(float d)
{
	const float r0 = __builtin_amdgcn_rcpf(d);
	const float e  = __builtin_fmaf(-d, r0, 1.0f);
	return __builtin_fmaf(e, r0, r0);
}
// a / d given r = recip_refined(d)
__device__ __forceinline__ float div_by(float a, float d, float r)
{
	const float q0 = a * r;
	const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, a), r, q0);
	return __builtin_fmaf(__builtin_fmaf(-d, q1, a), r, q1);
}
// 1 / d (the same sequence with a = 1: q0 = r)
__device__ __forceinline__ float recip_exact(float d)
{
	const float r  = recip_refined(d);
	const float q1 = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
	return __builtin_fmaf(__builtin_fmaf(-d, q1, 1.0f), r, q1);
}


// R8_UNORM store: round to nearest even.
__device__ __forceinline__ uint8_t store_unorm8(float g) { return (uint8_t) __builtin_rintf(g * 255.0f); }

// XCD-aware block remap: hardware deals consecutive block ids round-robin over the 8 XCDs, so give each XCD a
// contiguous range of logical ids (neighbouring screen tiles / slabs then share one L2).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nb)
{
	const uint32_t q = nb >> 3, r = nb & 7u, xcd = b & 7u, idx = b >> 3;
	return xcd * q + min(xcd, r) + idx;
}

// ---- transfer-function acceleration tables (vkv_transfer_function_tables), uint32 words ---------------------------------
// [0, 2048)     1 bit per texel of the 256x256 texture: alpha > 0 (row = gradient)
// [2048]        flags: kTfFlagSeparable = every texel is (b, b, b, b) with b = (uint8) clamp((ai[col] * ag[row]) * 255, 0, 255),
//               checked on the device for all 65536 texels with the arithmetic the integrator uses
// [2052, 2308)  ai[256] (float bits), [2308, 2564) ag[256]
constexpr uint32_t kTfFlagWord = 2048, kTfAiWord = 2052, kTfAgWord = 2308, kTfWords = 2564;
constexpr uint32_t kTfFlagSeparable = 1u;
static_assert(kTfWords == VKV_TF_BITS_WORDS, "include/vkvolume_amd.h and vkv_device.hpp disagree on the table size");

// the alpha byte of a separable greyscale transfer function (src/volume_component.cpp:246-261 builds exactly this product)
__host__ __device__ __forceinline__ uint32_t tf_separable_alpha(float ai, float ag)
{
	// ai, ag are in [0, 1] (k_tf_tables_init clamps them), so the product needs no clamp; the min keeps a table index in range whatever
	// the tables hold.  The flag that enables this path is only set after all 65536 texels were checked against exactly this function.
	const uint32_t b = (uint32_t) ((ai * ag) * 255.0f);
	return b < 255u ? b : 255u;
}

// ---- packed sampling layout (vkv_pack_volume) -------------------------------------------------------------------
// Padded index j in [0, W+2] holds voxel clamp(j-1, 0, W-1), so the clamp-to-edge footprint of texel index ix is always
// the pair (b, b+1) with b = clamp(ix, -1, W) + 1.  Brick (bx,by,bz) stores padded voxels [4b, 4b+4] per axis (5^3 with
// the apron) as interleaved (volume, gradient) byte pairs, x fastest: 250 bytes padded to 256.  Bricks are grouped
// 8x8x8 (one 128 KiB macro-brick = 32^3 voxels) so a ray's working set stays within a few pages.
struct PackedDims
{
	int bx, by, bz;        // bricks per axis
	int mx, my, mz;        // macro-bricks per axis
};

__host__ __device__ __forceinline__ PackedDims packed_dims(int W, int H, int D)
{
	PackedDims p;
	p.bx = ((W + 1) >> 2) + 1, p.by = ((H + 1) >> 2) + 1, p.bz = ((D + 1) >> 2) + 1;
	p.mx = (p.bx + 7) >> 3, p.my = (p.by + 7) >> 3, p.mz = (p.bz + 7) >> 3;
	return p;
}

__host__ __device__ __forceinline__ size_t packed_bytes(const PackedDims &p) { return (size_t) p.mx * p.my * p.mz * 512 * 256; }

__host__ __device__ __forceinline__ size_t packed_brick_offset(int bx, int by, int bz, int mx, int my)
{
	const size_t macro = ((size_t) (bz >> 3) * my + (size_t) (by >> 3)) * mx + (size_t) (bx >> 3);
	const size_t sub   = (size_t) (((bz & 7) << 6) | ((by & 7) << 3) | (bx & 7));
	return (macro * 512 + sub) * 256;
}

}        // namespace vkv

// Host-side context (capi.hip owns it).
// Device scratch is handed out PER STREAM (stream_scratch): calls on one stream are ordered, so they may share a buffer; calls on
// different streams never touch the same bytes, which makes every entry point re-entrant across streams of one context.
//
// Device memory policy (include/vkvolume_amd.h, "Conventions"): vkv_create allocates one ARENA with two regions - the scratch blocks of
// up to kScratchReserve streams, and the small immutable tables a launch needs (tile start orders, address tables) - plus a pinned host
// mirror of the table region.  A table is written into the mirror and uploaded from there asynchronously on the stream of the launch that
// first needs it; launches on other streams are ordered behind that upload with an event until it has completed.  Nothing is freed or
// re-used while a launch could read it: tables stay until vkv_trim (a set-up call that waits for the device, then empties the region) or
// vkv_destroy; when the region is full a launch runs without the table.  Only set-up calls (vkv_prepare_render, vkv_register_target) fall
// back to hipMalloc when a region is full.
struct vkv_ctx
{
	int   device;
	int   cu_count;        // compute units of the device (grid of the resident-workgroup kernels)
	char  error[512];
	void *d_trace;        // diagnostic wave timeline buffer (vkv_debug_trace), normally null
	const uint32_t *d_debug_orders;        // diagnostic per-frame tile start orders of vkv_render_batch (vkv_debug_tile_orders), normally null
	uint32_t        debug_order_frames, debug_order_count;
	std::mutex      mutex;
	VkvTuning       tuning;        // vkv_create: defaults + environment; vkv_set_tuning replaces it (read under the mutex, copied per call)
	// ---- device arena ----
	uint8_t *           arena = nullptr;
	size_t              arena_bytes = 0;
	size_t              table_base = 0;          // the arena's first table_base bytes are scratch blocks, the rest holds tables
	size_t              scratch_used = 0, table_used = 0;
	uint8_t *           table_mirror = nullptr;  // pinned host twin of the table region: the source of every asynchronous table upload
	std::vector<void *> overflow;                // hipMalloc blocks set-up calls took for tables when the region was full; freed by vkv_trim / vkv_destroy
	std::vector<void *> overflow_scratch;        // ... for scratch blocks beyond the reserve; freed by vkv_destroy
	// Argument blocks of vkv_render_batch launches captured into hipGraphs.  A captured launch owns a SLOT: a pinned host block (the graph's copy
	// node reads its source at every replay) and a device block of its own (the copy's target and the kernels' argument pointer: a graph may be
	// replayed on any stream, next to other graphs and to live launches, so it must not share the capture stream's scratch block).  kCaptureSlots
	// slots are set aside by vkv_create; later ones are allocated during the capture.  A slot belongs to the stream it was captured on until
	// vkv_release_captured(stream), vkv_trim or vkv_destroy.
	struct CaptureSlot
	{
		uint8_t *   pinned = nullptr, *device = nullptr;
		hipStream_t owner = nullptr;
		bool        in_use = false, pooled = false;
	};
	uint8_t *                capture_pool = nullptr, *capture_pool_device = nullptr;
	std::vector<CaptureSlot> capture_slots;
	std::unordered_map<hipStream_t, uint8_t *> scratch;        // stream -> kScratchBytes of device memory
	std::vector<uint8_t *>                     free_scratch;   // blocks given back by vkv_release_stream
	// an immutable device table with its host copy (the source of the asynchronous upload: it must outlive the call)
	struct Table
	{
		std::vector<uint32_t> host;
		uint32_t *            d = nullptr;
		hipEvent_t            uploaded = nullptr;        // recorded behind the upload
		hipStream_t           upload_stream = nullptr;
		bool                  ready = false;            // the upload is known to have completed: no more waits
	};
	// start orders of tile schedules (centre of the image first), built on first use and kept (heap objects: stable addresses)
	struct TileOrder
	{
		uint32_t tiles_x, tiles_y, tile_w, tile_h, img_w, img_h, first, stride, count;
		float    mix_heavy, mix_spread;
		Table    table;
	};
	std::vector<TileOrder *> tile_orders;
	// per-axis address tables of the packed sampling image, per volume extent
	struct AddrLut
	{
		int      W, H, D;
		uint32_t lut_y, lut_z, words;
		Table    table;
	};
	std::vector<AddrLut *> addr_luts;
	// start-order feedback of the targets registered with vkv_register_target: the tile costs the last measured frame into the target left
	// behind and the buffer its longest-first order is written to (raymarch.hip, apply_feedback); heap objects, so their addresses stay valid
	struct TileFeedback
	{
		const void *target;
		uint32_t    img_w, img_h, tile_w, tile_h, first, stride, count;
		uint32_t    org_x, org_y, tiles_x;        // the schedule's tile rectangle as the launcher sees it (pixels of its first tile, tile columns)
		uint32_t *  d_cost, *d_order;
		bool        has_cost;        // a frame has been rendered into this target with the cost buffer attached
		uint32_t    frames;          // frames rendered into this target so far (costs are measured and sorted every few frames)
		uint32_t    measured_at = 0; // value of `frames` at the last measured frame
		uint32_t    period = 8;      // frames until the next measurement (doubles while no frame can use the measured order)
		uint32_t    used = 0;        // frames since the last measurement that started in its order
		float       view_dir[3] = {0, 0, 0}, view_pos[3] = {0, 0, 0};        // central ray and camera position (texture space) of the measured frame
		float       prev_dir[3] = {0, 0, 0}, prev_pos[3] = {0, 0, 0};        // ... of the previous frame into the target (measured or not)
		bool        has_prev = false;
	};
	std::vector<TileFeedback *> feedback;
};

namespace vkv
{
int  set_error(vkv_ctx *ctx, int code, const char *fmt, ...);
int  check_launch(vkv_ctx *ctx, const char *what);
// this stream's scratch block (out of the arena on first use, kept until vkv_release_stream / vkv_destroy); nullptr + error set when
// there is no room
uint8_t *stream_scratch(vkv_ctx *ctx, hipStream_t stream, bool setup = false);
// start order of a tile schedule: entry indices sorted by the distance of the tile's centre from the image centre (device array of
// `count` uint32, cached per schedule shape); nullptr when the table cannot be allocated (the kernel then takes the tiles in order)
// Per-axis byte offsets of the packed sampling image: the offset of the footprint whose padded base texel is (bx, by, bz) is
// X(bx) + Y(by) + Z(bz) (the brick index and the position inside the brick are sums of per-axis terms), each in two levels:
// in-macro-brick term of b & 31 + macro-brick term of b >> 5.  Layout of the device array (uint32 words): in-macro tables of x, y, z
// at 0, 32, 64; macro terms of x at 96, of y at *lut_y, of z as 64-bit values at *lut_z (even); nullptr if it cannot be allocated.
// Both tables come out of the context's arena and are uploaded on `stream` when new (see vkv_ctx); `setup` = called from a set-up entry
// point: may fall back to hipMalloc when the arena is full.  nullptr when there is no room: the launch then runs without the table.
const uint32_t *packed_addr_lut(vkv_ctx *ctx, int W, int H, int D, uint32_t *lut_y, uint32_t *lut_z, uint32_t *words, hipStream_t stream, bool setup = false);
const uint32_t *tile_start_order(vkv_ctx *ctx, uint32_t img_w, uint32_t img_h, uint32_t tile_w, uint32_t tile_h, uint32_t first, uint32_t stride, uint32_t count,
                                 hipStream_t stream, bool setup = false);
VkvTuning tuning_of(vkv_ctx *ctx);        // a copy of the context's tuning block (taken under its mutex)
constexpr size_t kScratchBytes     = 128 * 1024;
constexpr size_t kScratchReserve   = 16;          // scratch blocks the arena keeps for streams (2 MiB of the default 8 MiB)
constexpr uint32_t kCaptureSlots   = 32;          // vkv_render_batch launches one context may have captured into hipGraphs
constexpr size_t   kCaptureSlotBytes = 96 * 1024;  // >= the pull heads + VKV_MAX_BATCH argument blocks (= the scratch block's argument area)
constexpr uint32_t kMaxDynamicLds  = 64 * 1024 - 1024;        // what a lean kernel may ask for as dynamic LDS (its tables; no hipFuncSetAttribute is called)
constexpr size_t kTfBitsOffset     = 0;           // 256*256 bits = 8 KiB: TF bit table of the map update / the voxel count (+ 8 words behind it: its column mask)
constexpr size_t kQueueHeadsOffset = 8192 + 64;       // 8 x u32 tile-queue heads of the persistent ray-march scheduler
constexpr size_t kPullHeadsBytes   = 2048;        // 8 ticket counters of k_raymarch_lean_pull, 256 bytes apart (one memory channel each), directly in
                                                  // front of the argument blocks: one upload zeroes the counters and brings the arguments
constexpr uint32_t kPullHeadStride = 64;         // in uint32 words
constexpr size_t kBatchArgsOffset  = 32 * 1024 + kPullHeadsBytes;        // vkv_render_batch: kMaxBatch argument blocks
constexpr uint32_t kMaxBatch       = VKV_MAX_BATCH;

// Every device entry point runs on the context's device whatever the calling thread's current device is, and leaves the
// caller's current device as it found it.
struct DeviceGuard
{
	int  prev = -1;
	bool switched = false;
	explicit DeviceGuard(int device)
	{
		if (hipGetDevice(&prev) == hipSuccess && prev != device)
			switched = hipSetDevice(device) == hipSuccess;
	}
	~DeviceGuard()
	{
		if (switched)
			(void) hipSetDevice(prev);
	}
	DeviceGuard(const DeviceGuard &) = delete;
	DeviceGuard &operator=(const DeviceGuard &) = delete;
};
}        // namespace vkv
