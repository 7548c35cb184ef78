// raymarch.hip — launchers of the ray-march integrator (device code: raymarch_core.hpp).
#include <cmath>
#include <cstdlib>

#include "raymarch_core.hpp"

namespace vkv
{

enum Scheduler
{
	kSchedLean       = 0,        // k_raymarch_lean: one lane per ray, predicated loop body (default)
	kSchedPersistent = 1         // k_raymarch_persistent: resident waves, ballot + mbcnt lane re-fill (bit-identical, slower)
};

template <int SKIP, bool ERT, int GRAD, bool PACKED>
static int launch_one(vkv_ctx *ctx, int sched, RayMarchArgs &a, hipStream_t s)
{
	if (sched == kSchedPersistent)
	{
		int per_cu = 0;
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess ||
		    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_raymarch_persistent<SKIP, ERT, GRAD, PACKED>, 256, 0) != hipSuccess || per_cu < 1)
			return set_error(ctx, VKV_E_NO_DEVICE, "render: occupancy query failed");
		const uint32_t resident = (uint32_t) per_cu * (uint32_t) prop.multiProcessorCount;
		// the tile-queue heads live in this stream's scratch: launches on other streams have their own
		uint8_t *scratch = stream_scratch(ctx, s);
		if (!scratch)
			return VKV_E_UNSUPPORTED;
		a.queue_heads      = reinterpret_cast<uint32_t *>(scratch + kQueueHeadsOffset);
		const hipError_t e = hipMemsetAsync(a.queue_heads, 0, 8 * sizeof(uint32_t), s);
		if (e != hipSuccess)
			return set_error(ctx, (int) e, "render: queue reset: %s", hipGetErrorString(e));
		// never more workgroups than there are 8x8 units to hand out (4 waves per workgroup)
		const uint32_t grid = resident < a.nblocks ? resident : a.nblocks;
		hipLaunchKernelGGL((k_raymarch_persistent<SKIP, ERT, GRAD, PACKED>), dim3(grid), dim3(256), 0, s, a);
	}
	else
	{
		// ids are dealt round-robin to the XCDs, each XCD walking its own tiles: pad the tile count to a multiple of 8
		const uint32_t grid = ((a.tile_count + 7u) / 8u) * 8u * a.blocks_per_tile;
		hipLaunchKernelGGL((k_raymarch_lean<SKIP, ERT, GRAD, PACKED, kLeanUniform>), dim3(grid), dim3(256), 0, s, a);
	}
	return check_launch(ctx, "render");
}

template <int SKIP, bool ERT, bool PACKED>
static int launch_grad(vkv_ctx *ctx, int sched, int grad, RayMarchArgs &a, hipStream_t s)
{
	if (grad == 0)
		return launch_one<SKIP, ERT, 0, PACKED>(ctx, sched, a, s);
	if (grad == 1)
		return launch_one<SKIP, ERT, 1, PACKED>(ctx, sched, a, s);
	return launch_one<SKIP, ERT, 2, PACKED>(ctx, sched, a, s);
}

template <int SKIP>
static int launch_ert(vkv_ctx *ctx, int sched, bool ert, int grad, RayMarchArgs &a, hipStream_t s)
{
	if (a.packed)
		return ert ? launch_grad<SKIP, true, true>(ctx, sched, grad, a, s) : launch_grad<SKIP, false, true>(ctx, sched, grad, a, s);
	return ert ? launch_grad<SKIP, true, false>(ctx, sched, grad, a, s) : launch_grad<SKIP, false, false>(ctx, sched, grad, a, s);
}

// VkvRenderParams -> kernel arguments.  Returns VKV_OK with a.nblocks == 0 when the schedule is empty.
int fill_render_args(vkv_ctx *ctx, const VkvRenderParams *P, const float *alpha_lut, RayMarchArgs &a)
{
	for (int i = 0; i < 3; ++i)
	{
		a.dir00[i] = P->ray_gen.dir00[i], a.ddx[i] = P->ray_gen.ddx[i], a.ddy[i] = P->ray_gen.ddy[i];
		a.cam[i]        = P->ray_cast.camera_pos_tex[i];
		a.block_size[i] = P->ray_cast.block_size[i];
	}
	for (int i = 0; i < 4; ++i)
		a.plane_tex[i] = P->ray_cast.plane_tex[i];
	for (int i = 0; i < 16; ++i)
		a.model[i] = P->camera.model[i], a.view[i] = P->camera.camera_view[i], a.proj[i] = P->camera.camera_proj[i],
		a.view_proj_inv[i] = P->camera.camera_view_proj_inv[i], a.model_inv[i] = P->camera.model_inv[i];
	a.sampling_factor = P->transfer_function.sampling_factor;
	a.grad_modifier   = P->transfer_function.grad_magnitude_modifier;
	a.W = (int) P->volume_extent.width, a.H = (int) P->volume_extent.height, a.D = (int) P->volume_extent.depth;
	a.mw = (int) P->map_extent.width, a.mh = (int) P->map_extent.height, a.md = (int) P->map_extent.depth;
	a.vol = P->d_volume, a.grad = P->d_gradient, a.tf = P->d_transfer_function;
	a.packed  = static_cast<const uint8_t *>(P->d_packed_volume);
	a.tf_bits = P->d_transfer_function_bits;
	{
		const PackedDims pd = packed_dims(a.W, a.H, a.D);
		a.pmx = pd.mx, a.pmy = pd.my;
	}
	for (int i = 0; i < 8; ++i)
		a.maps[i] = P->d_distance_maps[i];
	a.out_color = P->d_out_color, a.out_rgba8 = P->d_out_rgba8, a.out_counts = P->d_out_counts, a.out_depth = P->d_out_depth;
	a.in_depth         = P->d_in_depth;
	a.depth_attachment = P->options.depth_attachment != 0, a.blend = P->blend_over_target != 0;
	a.img_w = P->image_width, a.img_h = P->image_height;
	a.tile_w = P->tiles.tile_width, a.tile_h = P->tiles.tile_height;
	a.tiles_x    = (a.img_w + a.tile_w - 1) / a.tile_w;
	a.tile_first = P->tiles.tile_first, a.tile_stride = P->tiles.tile_stride, a.tile_count = P->tiles.tile_count, a.compact = P->tiles.compact;
	a.blocks_per_tile_x = a.tile_w / 16;
	a.blocks_per_tile   = a.blocks_per_tile_x * (a.tile_h / 16);
	const uint64_t nb   = (uint64_t) a.blocks_per_tile * a.tile_count;
	a.nblocks           = 0;
	if (nb == 0)
		return VKV_OK;
	if (nb > 0x3fffffffull)
		return set_error(ctx, VKV_E_UNSUPPORTED, "render: too many tiles for one launch");
	// output indices are 32-bit inside the kernel
	if ((uint64_t) a.img_w * a.img_h > 0xffffffffull / 4 || nb * 256 > 0xffffffffull / 4)
		return set_error(ctx, VKV_E_UNSUPPORTED, "render: frame too large for one launch");
	a.nblocks     = (uint32_t) nb;
	a.test        = P->options.test;
	a.queue_heads = nullptr;        // persistent scheduler: set at launch (per-stream scratch)
	a.trace       = reinterpret_cast<unsigned long long *>(ctx->d_trace);
	a.back        = (int) std::ceil(P->transfer_function.sampling_factor);
	for (int i = 0; i < 256; ++i)
		a.alpha_lut[i] = alpha_lut[i];
	return VKV_OK;
}

int launch_render(vkv_ctx *ctx, const VkvRenderParams *P, const float *alpha_lut, hipStream_t s)
{
	RayMarchArgs a;
	const int    rc = fill_render_args(ctx, P, alpha_lut, a);
	if (rc != VKV_OK || a.nblocks == 0)
		return rc;

	// scheduler: one lane per ray on static 8x8 tiles; VKV_RAYMARCH_SCHEDULER=persistent selects the lane-refilling persistent
	// waves (bit-identical output; measured 2.7x slower: the re-fill breaks the spatial coherence of a wave)
	const char *env   = std::getenv("VKV_RAYMARCH_SCHEDULER");        // read per call so a test can flip it
	const int   sched = (env && env[0] == 'p') ? (int) kSchedPersistent : (int) kSchedLean;

	const bool ert  = P->options.early_ray_termination != 0;
	const int  grad = !P->transfer_function.use_gradient ? 0 : (P->use_precomputed_gradient ? 1 : 2);
	switch (P->options.skipping_type)
	{
		case VKV_SKIP_NONE: return launch_ert<VKV_SKIP_NONE>(ctx, sched, ert, grad, a, s);
		case VKV_SKIP_BLOCK: return launch_ert<VKV_SKIP_BLOCK>(ctx, sched, ert, grad, a, s);
		case VKV_SKIP_DISTANCE: return launch_ert<VKV_SKIP_DISTANCE>(ctx, sched, ert, grad, a, s);
		case VKV_SKIP_ANISOTROPIC_DISTANCE: return launch_ert<VKV_SKIP_ANISOTROPIC_DISTANCE>(ctx, sched, ert, grad, a, s);
		default: return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: bad skipping_type %d", P->options.skipping_type);
	}
}

}        // namespace vkv
