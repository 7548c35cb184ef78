// raymarch.hip — launchers of the ray-march integrator (device code: raymarch_core.hpp).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "raymarch_core.hpp"

namespace vkv
{

enum Scheduler
{
	kSchedLean       = 0,        // k_raymarch_lean: one lane per ray, predicated loop body (default)
	kSchedPersistent = 1         // k_raymarch_persistent: resident waves, ballot + mbcnt lane re-fill (bit-identical, slower)
};

// The three instantiations of the lean kernel a launch chooses from (raymarch_core.hpp explains the flags):
constexpr uint32_t kLfPlain = kLeanDefault | kLeanNest | kLeanKeep | kLeanTf | kLeanWb | kLeanFloatI;        // footprint address worked out in registers: any volume, any map
constexpr uint32_t kLfLut   = kLfPlain | kLeanScalar | kLeanLut;           // two-level address tables in LDS (volumes up to ~2500 voxels per axis)
constexpr uint32_t kLfFull  = kLfLut | kLeanFull;                          // + one entry per voxel index with the separable transfer function
constexpr uint32_t kLfFullNc = kLfFull | kLeanNoCounts;                    // the same without the per-pixel counters (no d_out_counts: what a renderer launches)
constexpr uint32_t kLfLutNc  = kLfLut | kLeanNoCounts;

struct LeanChoice
{
	int    kind;        // 0 plain, 1 two-level tables, 2 full tables
	size_t lds;         // dynamic LDS bytes (lean_lds_bytes: the kernels' whole LDS layout lives in the dynamic segment)
};

static LeanChoice choose_lean(const RayMarchArgs &a, const VkvTuning &T)
{
	const size_t lut_bytes = (size_t) a.lut_words * sizeof(uint32_t);
	if (!a.packed || !a.addr_lut || lut_bytes > kMaxLutBytes || !map_fits_u24((uint32_t) a.mw, (uint32_t) a.mh, (uint32_t) a.md))
		return {0, lean_lds_bytes(0, 0, a.W, a.H, a.D)};
	const bool        no_full  = T.address_tables < 2;        // A/B switch: two-level only
	const size_t      full_end = kFullLutWord * 4 + full_lut_bytes(a.W, a.H, a.D);        // from the start of RmLds
	// the full tables hold offsets in units of two bytes in 32 bits: a packed image of up to 8 GiB
	const bool        fits_u32 = packed_bytes(packed_dims(a.W, a.H, a.D)) <= (1ull << 33);
	if (!no_full && fits_u32 && full_end <= (size_t) T.full_table_lds_limit)
		return {2, lean_lds_bytes(2, a.lut_words, a.W, a.H, a.D)};
	return {1, lean_lds_bytes(1, a.lut_words, a.W, a.H, a.D)};
}

template <int SKIP, bool ERT, int GRAD, bool PACKED>
static int launch_one(vkv_ctx *ctx, int sched, const VkvTuning &T, RayMarchArgs &a, hipStream_t s)
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
		bool launched = false;
		if constexpr (PACKED && GRAD != 2)
		{
			const LeanChoice c = choose_lean(a, T);
			bool no_counts = false;
			if constexpr (SKIP != VKV_SKIP_NONE && ERT && GRAD == 1)
				no_counts = c.kind != 0 && !a.out_counts && !a.pixel_cost;        // the common configuration only: every further instantiation costs build time
			if (no_counts)
			{
				if constexpr (SKIP != VKV_SKIP_NONE && ERT && GRAD == 1)
				{
					if (c.kind == 2)
						hipLaunchKernelGGL((k_raymarch_lean<SKIP, ERT, GRAD, PACKED, kLfFullNc>), dim3(grid), dim3(256), c.lds, s, a);
					else
						hipLaunchKernelGGL((k_raymarch_lean<SKIP, ERT, GRAD, PACKED, kLfLutNc>), dim3(grid), dim3(256), c.lds, s, a);
				}
			}
			else if (c.kind == 2)
				hipLaunchKernelGGL((k_raymarch_lean<SKIP, ERT, GRAD, PACKED, kLfFull>), dim3(grid), dim3(256), c.lds, s, a);
			else if (c.kind == 1)
				hipLaunchKernelGGL((k_raymarch_lean<SKIP, ERT, GRAD, PACKED, kLfLut>), dim3(grid), dim3(256), c.lds, s, a);
			launched = c.kind != 0;
		}
		if (!launched)
			hipLaunchKernelGGL((k_raymarch_lean<SKIP, ERT, GRAD, PACKED, kLfPlain>), dim3(grid), dim3(256), lean_lds_bytes(0, 0, a.W, a.H, a.D), s, a);
	}
	return check_launch(ctx, "render");
}

template <int SKIP, bool ERT, bool PACKED>
static int launch_grad(vkv_ctx *ctx, int sched, const VkvTuning &T, int grad, RayMarchArgs &a, hipStream_t s)
{
	if (grad == 0)
		return launch_one<SKIP, ERT, 0, PACKED>(ctx, sched, T, a, s);
	if (grad == 1)
		return launch_one<SKIP, ERT, 1, PACKED>(ctx, sched, T, a, s);
	return launch_one<SKIP, ERT, 2, PACKED>(ctx, sched, T, a, s);
}

template <int SKIP>
static int launch_ert(vkv_ctx *ctx, int sched, const VkvTuning &T, bool ert, int grad, RayMarchArgs &a, hipStream_t s)
{
	if (a.packed)
		return ert ? launch_grad<SKIP, true, true>(ctx, sched, T, grad, a, s) : launch_grad<SKIP, false, true>(ctx, sched, T, grad, a, s);
	return ert ? launch_grad<SKIP, true, false>(ctx, sched, T, grad, a, s) : launch_grad<SKIP, false, false>(ctx, sched, T, grad, a, s);
}

// Conservative pixel bound of the unit box [0,1]^3 (texture space) as seen through the ray generator of the kernel: pixel (px, py) looks
// along dir00 + (px + 0.5) ddx + (py + 0.5) ddy from cam, so a corner c is seen at the (fx, fy) with c - cam = g (dir00 + fx ddx + fy ddy),
// g > 0.  The bound is the min / max over the eight corners, widened by two pixels (the device evaluates the direction in fp32: it can
// disagree with this double-precision solve by a tiny fraction of a pixel); a corner at or behind the camera plane, or a degenerate
// generator, disables it.  Pixels outside cannot hit the box, whatever the clip plane or the depth test do afterwards.
static void screen_bound_of_box(RayMarchArgs &a, const VkvTuning &T)
{
	a.cull_x0 = 0u, a.cull_x1 = ~0u, a.cull_y0 = 0u, a.cull_y1 = ~0u;
	if (!T.screen_cull)        // A/B switch
		return;
	// inverse of M = [ddx ddy dir00] (columns) by the adjugate
	const double M[3][3] = {{a.ddx[0], a.ddy[0], a.dir00[0]}, {a.ddx[1], a.ddy[1], a.dir00[1]}, {a.ddx[2], a.ddy[2], a.dir00[2]}};
	double       inv[3][3], scale = 0.0;
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j)
		{
			const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
			inv[j][i]    = M[i1][j1] * M[i2][j2] - M[i1][j2] * M[i2][j1];        // cofactor (i, j) -> adjugate (j, i)
			scale        = std::max(scale, std::fabs(M[i][j]));
		}
	const double det = M[0][0] * inv[0][0] + M[0][1] * inv[1][0] + M[0][2] * inv[2][0];
	if (!std::isfinite(det) || !(std::fabs(det) > 1e-12 * scale * scale * scale))
		return;
	double lo_x = 1e300, hi_x = -1e300, lo_y = 1e300, hi_y = -1e300;
	for (int c = 0; c < 8; ++c)
	{
		const double v[3] = {(double) (c & 1) - a.cam[0], (double) ((c >> 1) & 1) - a.cam[1], (double) ((c >> 2) & 1) - a.cam[2]};
		const double fa = (inv[0][0] * v[0] + inv[0][1] * v[1] + inv[0][2] * v[2]) / det, fb = (inv[1][0] * v[0] + inv[1][1] * v[1] + inv[1][2] * v[2]) / det;
		const double g  = (inv[2][0] * v[0] + inv[2][1] * v[1] + inv[2][2] * v[2]) / det;
		if (!(g > 1e-6) || !std::isfinite(fa) || !std::isfinite(fb))
			return;        // a corner beside or behind the camera: its projection says nothing
		lo_x = std::min(lo_x, fa / g), hi_x = std::max(hi_x, fa / g), lo_y = std::min(lo_y, fb / g), hi_y = std::max(hi_y, fb / g);
	}
	// pixel p is sampled at p + 0.5
	const double x0 = std::floor(lo_x - 0.5) - 2.0, x1 = std::ceil(hi_x - 0.5) + 2.0, y0 = std::floor(lo_y - 0.5) - 2.0, y1 = std::ceil(hi_y - 0.5) + 2.0;
	if (x1 < 0.0 || y1 < 0.0 || x0 > 4.0e9 || y0 > 4.0e9)
	{        // the box is off screen: an empty bound
		a.cull_x0 = 1u, a.cull_x1 = 0u;
		return;
	}
	a.cull_x0 = x0 <= 0.0 ? 0u : (uint32_t) x0, a.cull_y0 = y0 <= 0.0 ? 0u : (uint32_t) y0;
	a.cull_x1 = x1 >= 4.0e9 ? ~0u : (uint32_t) x1, a.cull_y1 = y1 >= 4.0e9 ? ~0u : (uint32_t) y1;
}

// VkvRenderParams -> kernel arguments.  Returns VKV_OK with a.nblocks == 0 when the schedule is empty.
int fill_render_args(vkv_ctx *ctx, const VkvRenderParams *P, const float *alpha_lut, RayMarchArgs &a, hipStream_t s, const VkvTuning &T, bool setup)
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
	a.tile_cost = nullptr, a.order_out = nullptr, a.pixel_cost = nullptr;        // start-order / ray-order feedback: attached by the launchers
	a.trace       = reinterpret_cast<unsigned long long *>(ctx->d_trace);
	a.back        = (int) std::ceil(P->transfer_function.sampling_factor);
	a.addr_lut = nullptr, a.lut_y = a.lut_z = a.lut_words = 0;
	if (a.packed && T.address_tables != 0)
		a.addr_lut = packed_addr_lut(ctx, a.W, a.H, a.D, &a.lut_y, &a.lut_z, &a.lut_words, s, setup);
	screen_bound_of_box(a, T);
	a.tile_order  = T.tile_order_linear ? nullptr : tile_start_order(ctx, a.img_w, a.img_h, a.tile_w, a.tile_h, a.tile_first, a.tile_stride, a.tile_count, s, setup);
	for (int i = 0; i < 256; ++i)
		a.alpha_lut[i] = alpha_lut[i];
	return VKV_OK;
}

// Start-order feedback.  A renderer draws into the same target again and again with a camera that moves little from frame to frame,
// so the tiles that were expensive last time are expensive now: every marching wave leaves its iteration count in a per-target cost
// buffer (atomicMax per tile), and a small sort kernel behind the render (k_tile_orders_from_cost, one workgroup per frame, same stream)
// turns them into the longest-first order the next frame into that target starts its tiles in.  The per-target device state is created by
// vkv_register_target (a set-up call): a target that was never registered, the first frame into a registered one, and every frame when
// VkvTuning.feedback is 0, use the centre-of-image-first order.  Nothing is allocated, freed or waited for here.  Any order renders the
// same frame.  Returns true when `a` now asks for a sort (order_out set).
static bool apply_feedback(vkv_ctx *ctx, RayMarchArgs &a, hipStream_t s)
{
	(void) s;
	// costs are measured (and sorted behind the render) on the first frame into a target and then every `period`-th one: a camera that
	// moves little keeps the order good for a few frames, and the sort kernel + the cost atomics are then paid once per period
	const void *      target = a.out_rgba8 ? (const void *) a.out_rgba8 : (const void *) a.out_color;
	vkv_ctx::TileFeedback *     f = nullptr;
	std::lock_guard<std::mutex> lock(ctx->mutex);
	const uint32_t              period = ctx->tuning.feedback_period < 1u ? 1u : ctx->tuning.feedback_period;
	if (!ctx->tuning.feedback || !target || a.tile_count < 64 || ctx->d_debug_orders)
		return false;
	for (auto *e : ctx->feedback)
		if (e->target == target && e->img_w == a.img_w && e->img_h == a.img_h && e->tile_w == a.tile_w && e->tile_h == a.tile_h && e->first == a.tile_first &&
		    e->stride == a.tile_stride && e->count == a.tile_count)
		{
			f = e;
			break;
		}
	if (!f)
		return false;        // not registered (or registered for another schedule): centre-first
	if (ctx->tuning.ray_order && f->compact == a.compact)
		a.pixel_cost = f->d_pixel;        // read (sort key of this frame) and written (key of the next) by every frame into the target
	// the view of this frame: central ray (normalised) and camera position in texture space.  Costs measured on a view that was more
	// than ~12 degrees away (or from a camera that has moved by more than a fifth of its distance to the volume's centre) say little
	// about this frame - an order sorted by them scatters the heavy tiles (measured: -3 % for targets that alternate between views
	// 45 degrees apart) - so such a frame starts centre-first.
	float dir[3], len2 = 0.0f;
	for (int i = 0; i < 3; ++i)
	{
		dir[i] = a.dir00[i] + 0.5f * ((float) a.img_w * a.ddx[i] + (float) a.img_h * a.ddy[i]);
		len2 += dir[i] * dir[i];
	}
	const float inv = len2 > 0.0f ? 1.0f / std::sqrt(len2) : 0.0f;
	float       cosine = 0.0f, moved2 = 0.0f, dist2 = 0.0f;
	for (int i = 0; i < 3; ++i)
	{
		dir[i] *= inv;
		cosine += dir[i] * f->view_dir[i];
		moved2 += (a.cam[i] - f->view_pos[i]) * (a.cam[i] - f->view_pos[i]);
		dist2 += (f->view_pos[i] - 0.5f) * (f->view_pos[i] - 0.5f);
	}
	const bool     stale   = f->has_cost && !(cosine >= 0.978f && moved2 <= 0.04f * dist2);
	const uint32_t since   = f->frames - f->measured_at;
	const bool     measure = !f->has_cost || since >= f->period;
	if (measure)
	{
		// a measurement whose order no frame could use (the target kept jumping between far views) doubles the distance to the next
		// one, up to 8 periods: such a target then pays next to nothing for the feedback it cannot use
		f->period      = (f->has_cost && f->used == 0) ? std::min(2u * f->period, 8u * period) : period;
		f->measured_at = f->frames;
		f->used        = 0;
	}
	++f->frames;
	if (f->has_cost && !stale)
	{
		a.tile_order = f->d_order;        // the order the last sort behind a frame into this target left
		++f->used;
	}
	if (!measure)
		return false;
	for (int i = 0; i < 3; ++i)
		f->view_dir[i] = dir[i], f->view_pos[i] = a.cam[i];
	a.tile_cost = f->d_cost;
	a.order_out = f->d_order;        // written by the sort that FOLLOWS this frame's render on the stream
	f->has_cost = true;
	return true;
}

int launch_render(vkv_ctx *ctx, const VkvRenderParams *P, const float *alpha_lut, hipStream_t s)
{
	const VkvTuning T = tuning_of(ctx);
	RayMarchArgs a;
	const int    rc = fill_render_args(ctx, P, alpha_lut, a, s, T, false);
	if (rc != VKV_OK || a.nblocks == 0)
		return rc;

	// scheduler: one lane per ray on static 8x8 tiles; VkvTuning.scheduler = 1 selects the lane-refilling persistent
	// waves (bit-identical output; measured 2.7x slower: the re-fill breaks the spatial coherence of a wave)
	const int   sched = T.scheduler == 1 ? (int) kSchedPersistent : (int) kSchedLean;

	const bool ert  = P->options.early_ray_termination != 0;
	const int  grad = !P->transfer_function.use_gradient ? 0 : (P->use_precomputed_gradient ? 1 : 2);
	// start-order feedback as in vkv_render_batch (with early ray termination only; the sort runs BEHIND the render on the stream): a
	// frame that runs alone also ends earlier when its long tiles start first (C3: 0.256 -> 0.246 ms)
	const bool sort = sched == (int) kSchedLean && ert && apply_feedback(ctx, a, s);
	int        rc2;
	switch (P->options.skipping_type)
	{
		case VKV_SKIP_NONE: rc2 = launch_ert<VKV_SKIP_NONE>(ctx, sched, T, ert, grad, a, s); break;
		case VKV_SKIP_BLOCK: rc2 = launch_ert<VKV_SKIP_BLOCK>(ctx, sched, T, ert, grad, a, s); break;
		case VKV_SKIP_DISTANCE: rc2 = launch_ert<VKV_SKIP_DISTANCE>(ctx, sched, T, ert, grad, a, s); break;
		case VKV_SKIP_ANISOTROPIC_DISTANCE: rc2 = launch_ert<VKV_SKIP_ANISOTROPIC_DISTANCE>(ctx, sched, T, ert, grad, a, s); break;
		default: return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: bad skipping_type %d", P->options.skipping_type);
	}
	if (rc2 == VKV_OK && sort)
	{
		hipLaunchKernelGGL(k_tile_order_from_cost, dim3(1), dim3(256), 0, s, a.tile_cost, a.order_out, a.tile_count);
		rc2 = check_launch(ctx, "render (start order)");
	}
	return rc2;
}

// vkv_prepare_render: everything a later launch of these parameter blocks on `s` takes from the context, created now (set-up call)
int prepare_render(vkv_ctx *ctx, const VkvRenderParams *P, uint32_t n, hipStream_t s)
{
	const VkvTuning T = tuning_of(ctx);
	float           lut[256] = {};
	for (uint32_t i = 0; i < n; ++i)
	{
		RayMarchArgs a;
		const int    rc = fill_render_args(ctx, &P[i], lut, a, s, T, true);
		if (rc != VKV_OK)
			return rc;
	}
	if (!stream_scratch(ctx, s, true))
		return VKV_E_UNSUPPORTED;
	return VKV_OK;
}

// ---- several frames in one launch --------------------------------------------------------------------------------
template <int SKIP, bool ERT, int GRAD>
static void launch_batch_kind(LeanChoice c, bool no_counts, const RayMarchArgs *d_frames, uint32_t n, uint32_t grid, uint32_t gpf, hipStream_t s)
{
	if constexpr (GRAD != 2)
	{
		if constexpr (SKIP != VKV_SKIP_NONE && ERT && GRAD == 1)
		{
			if (c.kind != 0 && no_counts)
			{
				if (c.kind == 2)
					hipLaunchKernelGGL((k_raymarch_lean_batch<SKIP, ERT, GRAD, kLfFullNc>), dim3(grid), dim3(256), c.lds, s, d_frames, n, gpf);
				else
					hipLaunchKernelGGL((k_raymarch_lean_batch<SKIP, ERT, GRAD, kLfLutNc>), dim3(grid), dim3(256), c.lds, s, d_frames, n, gpf);
				return;
			}
		}
		if (c.kind == 2)
			hipLaunchKernelGGL((k_raymarch_lean_batch<SKIP, ERT, GRAD, kLfFull>), dim3(grid), dim3(256), c.lds, s, d_frames, n, gpf);
		else if (c.kind == 1)
			hipLaunchKernelGGL((k_raymarch_lean_batch<SKIP, ERT, GRAD, kLfLut>), dim3(grid), dim3(256), c.lds, s, d_frames, n, gpf);
		if (c.kind != 0)
			return;
	}
	hipLaunchKernelGGL((k_raymarch_lean_batch<SKIP, ERT, GRAD, kLfPlain>), dim3(grid), dim3(256), lean_lds_bytes(0, 0, 0, 0, 0), s, d_frames, n, gpf);
}

template <int SKIP, bool ERT>
static void launch_batch_grad(int grad, const RayMarchArgs *d_frames, uint32_t n, uint32_t grid, uint32_t gpf, LeanChoice c, bool no_counts, hipStream_t s)
{
	// c.kind > 0: every frame of the batch has address tables of that size (same volume extent) and they fit the LDS budget
	if (grad == 0)
		launch_batch_kind<SKIP, ERT, 0>(c, no_counts, d_frames, n, grid, gpf, s);
	else if (grad == 1)
		launch_batch_kind<SKIP, ERT, 1>(c, no_counts, d_frames, n, grid, gpf, s);
	else
		launch_batch_kind<SKIP, ERT, 2>(c, no_counts, d_frames, n, grid, gpf, s);
}

template <int SKIP>
static void launch_batch_ert(bool ert, int grad, const RayMarchArgs *d_frames, uint32_t n, uint32_t grid, uint32_t gpf, LeanChoice c, bool no_counts, hipStream_t s)
{
	if (ert)
		launch_batch_grad<SKIP, true>(grad, d_frames, n, grid, gpf, c, no_counts, s);
	else
		launch_batch_grad<SKIP, false>(grad, d_frames, n, grid, gpf, c, no_counts, s);
}

// ---- the same, with resident workgroups whose waves pull their units ------------------------------------------------
// grid = the workgroups the device holds at once (occupancy of this instantiation x CUs), never more than there are units / 4
template <int SKIP, bool ERT, int GRAD, uint32_t LF>
static uint32_t launch_pull_one(vkv_ctx *ctx, const RayMarchArgs *d_frames, uint32_t n, uint32_t *d_heads, size_t lds, uint64_t units, hipStream_t s)
{
	// eight workgroups of four waves fill a CU's 32 wave slots (this instantiation is held to 64 VGPRs); should fewer fit, the surplus
	// workgroups start when others have finished and take what tickets are left
	const uint64_t resident = (uint64_t) 8 * (uint64_t) std::max(1, ctx->cu_count);
	const uint32_t grid     = (uint32_t) std::max<uint64_t>(8, std::min<uint64_t>(resident, (units + 3) / 4));
	hipLaunchKernelGGL((k_raymarch_lean_pull<SKIP, ERT, GRAD, LF>), dim3(grid), dim3(256), lds, s, d_frames, n, d_heads);
	return grid;
}

template <int SKIP, bool ERT, int GRAD>
static uint32_t launch_pull_kind(vkv_ctx *ctx, LeanChoice c, const RayMarchArgs *d_frames, uint32_t n, uint32_t *d_heads, uint64_t units, hipStream_t s)
{
	if constexpr (GRAD != 2)
	{
		if (c.kind == 2)
			return launch_pull_one<SKIP, ERT, GRAD, kLfFull>(ctx, d_frames, n, d_heads, c.lds, units, s);
		if (c.kind == 1)
			return launch_pull_one<SKIP, ERT, GRAD, kLfLut>(ctx, d_frames, n, d_heads, c.lds, units, s);
	}
	return launch_pull_one<SKIP, ERT, GRAD, kLfPlain>(ctx, d_frames, n, d_heads, lean_lds_bytes(0, 0, 0, 0, 0), units, s);
}

template <int SKIP>
static uint32_t launch_pull_ert(vkv_ctx *ctx, bool ert, int grad, const RayMarchArgs *d_frames, uint32_t n, uint32_t *d_heads, LeanChoice c, uint64_t units,
                                hipStream_t s)
{
	if (ert)
		return grad == 0 ? launch_pull_kind<SKIP, true, 0>(ctx, c, d_frames, n, d_heads, units, s)
		                 : (grad == 1 ? launch_pull_kind<SKIP, true, 1>(ctx, c, d_frames, n, d_heads, units, s) : launch_pull_kind<SKIP, true, 2>(ctx, c, d_frames, n, d_heads, units, s));
	return grad == 0 ? launch_pull_kind<SKIP, false, 0>(ctx, c, d_frames, n, d_heads, units, s)
	                 : (grad == 1 ? launch_pull_kind<SKIP, false, 1>(ctx, c, d_frames, n, d_heads, units, s) : launch_pull_kind<SKIP, false, 2>(ctx, c, d_frames, n, d_heads, units, s));
}

int launch_render_batch(vkv_ctx *ctx, const VkvRenderParams *P, uint32_t n, const float *alpha_luts, hipStream_t s)
{
	// the argument blocks go through this stream's scratch buffer: an earlier batch on the same stream has finished with it by the
	// time the copy (same stream) runs
	static_assert(kMaxBatch * sizeof(RayMarchArgs) <= kScratchBytes - kBatchArgsOffset, "batch argument blocks must fit the stream scratch");
	const VkvTuning           T = tuning_of(ctx);
	std::vector<RayMarchArgs> host(n);
	for (uint32_t i = 0; i < n; ++i)
	{
		const int rc = fill_render_args(ctx, &P[i], alpha_luts + (size_t) i * 256, host[i], s, T, false);
		if (rc != VKV_OK)
			return rc;
		if (!host[i].packed)
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render_batch: frame %u has no packed sampling image (d_packed_volume)", i);
		if (host[i].nblocks != host[0].nblocks || host[i].tile_count != host[0].tile_count || host[i].blocks_per_tile != host[0].blocks_per_tile)
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render_batch: frame %u has a different tile schedule size than frame 0", i);
		if (ctx->d_debug_orders && i < ctx->debug_order_frames && ctx->debug_order_count == host[i].tile_count)
			host[i].tile_order = ctx->d_debug_orders + (size_t) i * ctx->debug_order_count;        // diagnostic start orders
	}
	if (host[0].nblocks == 0)
		return VKV_OK;
	// measured start order only with early ray termination: without it every covered tile is about equally long, nothing is gained by
	// starting the longest first, and an order sorted by cost scatters neighbouring tiles (C3 without ERT: 0.394 against 0.373 ms per frame)
	bool any_sort = false;
	for (uint32_t i = 0; i < n && P[0].options.early_ray_termination != 0; ++i)
		any_sort = apply_feedback(ctx, host[i], s) || any_sort;
	uint8_t *scratch = stream_scratch(ctx, s);
	if (!scratch)
		return VKV_E_UNSUPPORTED;
	RayMarchArgs *   d_frames = reinterpret_cast<RayMarchArgs *>(scratch + kBatchArgsOffset);
	uint32_t *       d_heads  = reinterpret_cast<uint32_t *>(scratch + kBatchArgsOffset - kPullHeadsBytes);
	// one upload: the (zeroed) ticket counters of the pull kernel, then the argument blocks
	std::vector<uint8_t> upload(kPullHeadsBytes + n * sizeof(RayMarchArgs), 0);
	std::memcpy(upload.data() + kPullHeadsBytes, host.data(), n * sizeof(RayMarchArgs));
	const hipError_t e = hipMemcpyAsync(d_heads, upload.data(), upload.size(), hipMemcpyHostToDevice, s);
	if (e != hipSuccess)
		return set_error(ctx, (int) e, "render_batch: argument upload: %s", hipGetErrorString(e));
	const uint64_t grid = (uint64_t) ((host[0].tile_count + 7u) / 8u) * 8u * host[0].blocks_per_tile * n;
	if (grid > 0x7fffffffull)
		return set_error(ctx, VKV_E_UNSUPPORTED, "render_batch: too many workgroups for one launch");
	const bool ert  = P[0].options.early_ray_termination != 0;
	const int  grad = !P[0].transfer_function.use_gradient ? 0 : (P[0].use_precomputed_gradient ? 1 : 2);
	// frames interleaved in groups of eight workgroups (default) or one frame after the other (VkvTuning.batch_sequential, A/B
	// switch: measured 0.183 vs 0.169 ms per frame on C3 with 8 frames per launch)
	const bool        sequential = T.batch_sequential != 0;
	const uint32_t    gpf        = sequential ? (uint32_t) (grid / n / 8) : 0u;
	LeanChoice        choice     = choose_lean(host[0], T);
	bool              no_counts  = true;        // no frame wants the per-pixel counters (or sorts its rays by them): the loop without them
	for (uint32_t i = 0; i < n; ++i)
		no_counts = no_counts && !host[i].out_counts && !host[i].pixel_cost;
	for (uint32_t i = 1; i < n; ++i)
	{        // one kernel for all frames: the most general choice any of them needs
		const LeanChoice c = choose_lean(host[i], T);
		if (c.kind != choice.kind || c.lds != choice.lds || host[i].lut_words != host[0].lut_words)
			choice = {0, 0};
	}
	// VkvTuning.batch_mode = 1 (pull): resident workgroups whose waves pull 8x8 units from per-XCD ticket counters (k_raymarch_lean_pull), possible
	// when every frame shares the LDS tables (same packed image and extents, TF tables, opacity table).  Bit-identical; measured on C3 with 8
	// frames per launch: the CU stays full (7 900 of 8 192 wave slots against 5 300) and everything but the last marching tiles is done after
	// 0.83 ms instead of 1.0, but those last tiles - the volume's silhouette, 100-250 iterations of cold probes - then run 340 us with their
	// four 8x8 units on four different CUs (150 us as one workgroup on one CU): 0.151 ms per frame against 0.137.  Not the default.
	bool              pull      = T.batch_mode == 1 && !sequential;
	const uint64_t    units     = (uint64_t) host[0].tile_count * host[0].blocks_per_tile * 4u * n;
	for (uint32_t i = 1; i < n && pull; ++i)
	{
		const RayMarchArgs &a = host[i], &b = host[0];
		pull = a.packed == b.packed && a.tf_bits == b.tf_bits && a.tf == b.tf && a.addr_lut == b.addr_lut && a.W == b.W && a.H == b.H && a.D == b.D &&
		       std::memcmp(a.alpha_lut, b.alpha_lut, sizeof(a.alpha_lut)) == 0;
	}
	if (pull)
	{
		uint32_t resident = 0;
		switch (P[0].options.skipping_type)
		{
			case VKV_SKIP_NONE: resident = launch_pull_ert<VKV_SKIP_NONE>(ctx, ert, grad, d_frames, n, d_heads, choice, units, s); break;
			case VKV_SKIP_BLOCK: resident = launch_pull_ert<VKV_SKIP_BLOCK>(ctx, ert, grad, d_frames, n, d_heads, choice, units, s); break;
			case VKV_SKIP_DISTANCE: resident = launch_pull_ert<VKV_SKIP_DISTANCE>(ctx, ert, grad, d_frames, n, d_heads, choice, units, s); break;
			case VKV_SKIP_ANISOTROPIC_DISTANCE: resident = launch_pull_ert<VKV_SKIP_ANISOTROPIC_DISTANCE>(ctx, ert, grad, d_frames, n, d_heads, choice, units, s); break;
			default: return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render_batch: bad skipping_type %d", P[0].options.skipping_type);
		}
		if (resident == 0)
			return set_error(ctx, VKV_E_NO_DEVICE, "render_batch: occupancy query failed");
		if (any_sort)
			hipLaunchKernelGGL(k_tile_orders_from_cost, dim3(n), dim3(256), 0, s, d_frames);
		return check_launch(ctx, "render_batch");
	}
	switch (P[0].options.skipping_type)
	{
		case VKV_SKIP_NONE: launch_batch_ert<VKV_SKIP_NONE>(ert, grad, d_frames, n, (uint32_t) grid, gpf, choice, no_counts, s); break;
		case VKV_SKIP_BLOCK: launch_batch_ert<VKV_SKIP_BLOCK>(ert, grad, d_frames, n, (uint32_t) grid, gpf, choice, no_counts, s); break;
		case VKV_SKIP_DISTANCE: launch_batch_ert<VKV_SKIP_DISTANCE>(ert, grad, d_frames, n, (uint32_t) grid, gpf, choice, no_counts, s); break;
		case VKV_SKIP_ANISOTROPIC_DISTANCE: launch_batch_ert<VKV_SKIP_ANISOTROPIC_DISTANCE>(ert, grad, d_frames, n, (uint32_t) grid, gpf, choice, no_counts, s); break;
		default: return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render_batch: bad skipping_type %d", P[0].options.skipping_type);
	}
	// behind the render, on the same stream: the costs it measured become the start order of the next frames into these targets (the
	// sort is not in front of anybody's render this way; in front it cost 70 us per 20-frame block of three launches)
	if (any_sort)
		hipLaunchKernelGGL(k_tile_orders_from_cost, dim3(n), dim3(256), 0, s, d_frames);
	return check_launch(ctx, "render_batch");
}

}        // namespace vkv
