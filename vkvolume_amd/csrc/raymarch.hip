// raymarch.hip — launchers of the ray-march integrator (device code: raymarch_core.hpp).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "raymarch_inst.hpp"

// ---------------------------------------------------------------------------------------------------------------
// Start order from measured costs: one workgroup per frame sorts the schedule entries 0 .. count - 1 by the cost the previous frame
// into the same target left in tile_cost, longest first (a counting sort over min(cost, 1023); entries of equal cost keep no
// particular order: any permutation renders the same frame, and tie orders measured the same), writes the order to order_out and
// clears the costs for the frame that is about to be rendered.  Centre-of-image-first is only a guess at "longest first": on C3 the tiles that finish
// last are the volume's silhouette.  With the measured order a launch of 8 frames takes 1.01 instead of 1.10 ms.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kCostBins = 1024;

__device__ __forceinline__ void tile_order_from_cost(uint32_t *__restrict__ cost, uint32_t *__restrict__ order_out, uint32_t count)
{
	__shared__ uint32_t s_bin[kCostBins];        // histogram, then the next free position of every bin
	constexpr int       kPerThread = 40;         // 256 threads x 40 = 10 240 tiles per pass, costs kept in registers between the two passes
	for (int b = threadIdx.x; b < kCostBins; b += blockDim.x)
		s_bin[b] = 0;
	__syncthreads();
	for (uint32_t base = 0; base < count; base += 256u * kPerThread)
	{
		uint32_t c[kPerThread];
#pragma unroll
		for (int j = 0; j < kPerThread; ++j)
		{        // all loads of a thread in flight together (coalesced: consecutive threads, consecutive tiles)
			const uint32_t t = base + (uint32_t) j * 256u + threadIdx.x;
			c[j]             = t < count ? min(cost[t], (uint32_t) kCostBins - 1u) : 0xffffffffu;
		}
		// Half of a frame's tiles cost nothing (no ray of theirs marches): one LDS atomic per tile puts 4 000 of them on ONE address, which the
		// LDS serialises; the zero-cost tiles of a wave are counted with one ballot and one atomic by the wave's first such lane.  (Round 6,
		// the sort alone on 8 160 tiles: 23 us, of which 4.9 launch, ~5 the loads, 3.8 the histogram, 9.3 prefix + scatter - one workgroup is
		// a chain of latencies; it runs once per 8 launches and stream, 0.5 % of the frame time, and was left at that.)
#pragma unroll
		for (int j = 0; j < kPerThread; ++j)
		{
			const bool               zero = c[j] == 0u;
			const unsigned long long zm   = __ballot(zero);
			if (zero)
			{
				if ((zm & ((1ull << (threadIdx.x & 63u)) - 1ull)) == 0ull)
					atomicAdd(&s_bin[kCostBins - 1], (uint32_t) __popcll(zm));
			}
			else if (c[j] != 0xffffffffu)
				atomicAdd(&s_bin[kCostBins - 1 - c[j]], 1u);        // bin 0 = the dearest tiles
		}
		if (base + 256u * kPerThread >= count)
		{        // the common case (up to 10 240 tiles, a 2560 x 1024 frame): one pass, the scatter reuses the registers
			__syncthreads();
			if (threadIdx.x < 64)        // (workgroups of 256 threads: four waves find a place on a CU that render workgroups keep full)
			{        // exclusive prefix sum of the 1024 bins by one wave: 16 bins per lane
				uint32_t local[kCostBins / 64], sum = 0;
#pragma unroll
				for (int j = 0; j < kCostBins / 64; ++j)
					local[j] = s_bin[threadIdx.x * (kCostBins / 64) + j], sum += local[j];
				uint32_t incl = sum;
				for (int o = 1; o < 64; o <<= 1)
				{
					const uint32_t up = (uint32_t) __shfl_up((int) incl, o);
					if ((int) threadIdx.x >= o)
						incl += up;
				}
				uint32_t run = incl - sum;
#pragma unroll
				for (int j = 0; j < kCostBins / 64; ++j)
					s_bin[threadIdx.x * (kCostBins / 64) + j] = run, run += local[j];
			}
			__syncthreads();
			if (base == 0)
			{
#pragma unroll
				for (int j = 0; j < kPerThread; ++j)
				{
					const uint32_t           t    = (uint32_t) j * 256u + threadIdx.x;
					const bool               zero = c[j] == 0u;
					const unsigned long long zm   = __ballot(zero);
					if (zero)
					{        // the wave's zero-cost tiles take consecutive places behind one atomic (their costs are zero already)
						const unsigned long long below = zm & ((1ull << (threadIdx.x & 63u)) - 1ull);
						uint32_t                 first = 0;
						if (below == 0ull)
							first = atomicAdd(&s_bin[kCostBins - 1], (uint32_t) __popcll(zm));
						first = (uint32_t) __shfl((int) first, __ffsll((long long) zm) - 1);
						order_out[first + (uint32_t) __popcll(below)] = t;
					}
					else if (c[j] != 0xffffffffu)
					{
						order_out[atomicAdd(&s_bin[kCostBins - 1 - c[j]], 1u)] = t;
						cost[t] = 0;
					}
				}
				return;
			}
		}
	}
	// larger schedules: second pass over memory
	for (uint32_t t = threadIdx.x; t < count; t += blockDim.x)
	{
		const uint32_t pos = atomicAdd(&s_bin[kCostBins - 1 - min(cost[t], (uint32_t) kCostBins - 1u)], 1u);
		if (pos < count)        // always, unless somebody changed the costs between the two passes
			order_out[pos] = t;
		cost[t] = 0;
	}
}

__global__ void __launch_bounds__(256) k_tile_order_from_cost(uint32_t *__restrict__ cost, uint32_t *__restrict__ order_out, uint32_t count)
{
	tile_order_from_cost(cost, order_out, count);
}

// the frames of a batch whose argument block asks for it (order_out set)
__global__ void __launch_bounds__(256) k_tile_orders_from_cost(const RayMarchArgs *__restrict__ frames)
{
	const RayMarchArgs &A = frames[blockIdx.x];
	if (A.order_out)
		tile_order_from_cost(A.tile_cost, A.order_out, A.tile_count);
}

namespace vkv
{

LeanChoice choose_lean(const RayMarchArgs &a, const VkvTuning &T)
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

// (skipping type, early ray termination) -> the translation unit that holds the kernels of that pair
static int launch_single(vkv_ctx *ctx, int skip, bool ert, int sched, const VkvTuning &T, int grad, RayMarchArgs &a, hipStream_t s)
{
	switch (skip)
	{
		case VKV_SKIP_NONE: return ert ? RayMarchLaunchers<VKV_SKIP_NONE, true>::single(ctx, sched, T, grad, a, s) : RayMarchLaunchers<VKV_SKIP_NONE, false>::single(ctx, sched, T, grad, a, s);
		case VKV_SKIP_BLOCK: return ert ? RayMarchLaunchers<VKV_SKIP_BLOCK, true>::single(ctx, sched, T, grad, a, s) : RayMarchLaunchers<VKV_SKIP_BLOCK, false>::single(ctx, sched, T, grad, a, s);
		case VKV_SKIP_DISTANCE: return ert ? RayMarchLaunchers<VKV_SKIP_DISTANCE, true>::single(ctx, sched, T, grad, a, s) : RayMarchLaunchers<VKV_SKIP_DISTANCE, false>::single(ctx, sched, T, grad, a, s);
		case VKV_SKIP_ANISOTROPIC_DISTANCE:
			return ert ? RayMarchLaunchers<VKV_SKIP_ANISOTROPIC_DISTANCE, true>::single(ctx, sched, T, grad, a, s) : RayMarchLaunchers<VKV_SKIP_ANISOTROPIC_DISTANCE, false>::single(ctx, sched, T, grad, a, s);
		default: return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: bad skipping_type %d", skip);
	}
}

static void load_render_code(int skip, bool ert)
{
	switch (skip)
	{
		case VKV_SKIP_NONE: ert ? RayMarchLaunchers<VKV_SKIP_NONE, true>::load() : RayMarchLaunchers<VKV_SKIP_NONE, false>::load(); break;
		case VKV_SKIP_BLOCK: ert ? RayMarchLaunchers<VKV_SKIP_BLOCK, true>::load() : RayMarchLaunchers<VKV_SKIP_BLOCK, false>::load(); break;
		case VKV_SKIP_DISTANCE: ert ? RayMarchLaunchers<VKV_SKIP_DISTANCE, true>::load() : RayMarchLaunchers<VKV_SKIP_DISTANCE, false>::load(); break;
		case VKV_SKIP_ANISOTROPIC_DISTANCE: ert ? RayMarchLaunchers<VKV_SKIP_ANISOTROPIC_DISTANCE, true>::load() : RayMarchLaunchers<VKV_SKIP_ANISOTROPIC_DISTANCE, false>::load(); break;
		default: break;
	}
}

// the start-order kernels of this file (vkv_register_target: the first launch into a registered target would load them otherwise)
void load_feedback_code()
{

	hipFuncAttributes at;
	(void) hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_tile_order_from_cost));
}

static bool launch_batch(int skip, bool ert, int grad, const RayMarchArgs *d_frames, uint32_t n, uint32_t grid, uint32_t gpf, LeanChoice c, bool no_counts, hipStream_t s)
{
	switch (skip)
	{
		case VKV_SKIP_NONE: ert ? RayMarchLaunchers<VKV_SKIP_NONE, true>::batch(grad, d_frames, n, grid, gpf, c, no_counts, s) : RayMarchLaunchers<VKV_SKIP_NONE, false>::batch(grad, d_frames, n, grid, gpf, c, no_counts, s); return true;
		case VKV_SKIP_BLOCK: ert ? RayMarchLaunchers<VKV_SKIP_BLOCK, true>::batch(grad, d_frames, n, grid, gpf, c, no_counts, s) : RayMarchLaunchers<VKV_SKIP_BLOCK, false>::batch(grad, d_frames, n, grid, gpf, c, no_counts, s); return true;
		case VKV_SKIP_DISTANCE: ert ? RayMarchLaunchers<VKV_SKIP_DISTANCE, true>::batch(grad, d_frames, n, grid, gpf, c, no_counts, s) : RayMarchLaunchers<VKV_SKIP_DISTANCE, false>::batch(grad, d_frames, n, grid, gpf, c, no_counts, s); return true;
		case VKV_SKIP_ANISOTROPIC_DISTANCE:
			ert ? RayMarchLaunchers<VKV_SKIP_ANISOTROPIC_DISTANCE, true>::batch(grad, d_frames, n, grid, gpf, c, no_counts, s) : RayMarchLaunchers<VKV_SKIP_ANISOTROPIC_DISTANCE, false>::batch(grad, d_frames, n, grid, gpf, c, no_counts, s);
			return true;
		default: return false;
	}
}

static bool launch_pull(vkv_ctx *ctx, int skip, bool ert, int grad, const RayMarchArgs *d_frames, uint32_t n, uint32_t *d_heads, LeanChoice c, uint64_t units, hipStream_t s)
{
	switch (skip)
	{
		case VKV_SKIP_NONE: ert ? RayMarchLaunchers<VKV_SKIP_NONE, true>::pull(ctx, grad, d_frames, n, d_heads, c, units, s) : RayMarchLaunchers<VKV_SKIP_NONE, false>::pull(ctx, grad, d_frames, n, d_heads, c, units, s); return true;
		case VKV_SKIP_BLOCK: ert ? RayMarchLaunchers<VKV_SKIP_BLOCK, true>::pull(ctx, grad, d_frames, n, d_heads, c, units, s) : RayMarchLaunchers<VKV_SKIP_BLOCK, false>::pull(ctx, grad, d_frames, n, d_heads, c, units, s); return true;
		case VKV_SKIP_DISTANCE: ert ? RayMarchLaunchers<VKV_SKIP_DISTANCE, true>::pull(ctx, grad, d_frames, n, d_heads, c, units, s) : RayMarchLaunchers<VKV_SKIP_DISTANCE, false>::pull(ctx, grad, d_frames, n, d_heads, c, units, s); return true;
		case VKV_SKIP_ANISOTROPIC_DISTANCE:
			ert ? RayMarchLaunchers<VKV_SKIP_ANISOTROPIC_DISTANCE, true>::pull(ctx, grad, d_frames, n, d_heads, c, units, s) : RayMarchLaunchers<VKV_SKIP_ANISOTROPIC_DISTANCE, false>::pull(ctx, grad, d_frames, n, d_heads, c, units, s);
			return true;
		default: return false;
	}
}

// Conservative pixel bound of what a frame's fragments can see: the unit box [0,1]^3 (texture space) cut by the clipping plane (kept side:
// dot(plane_tex.xyz, p) + plane_tex.w >= 0 - ray_setup_impl starts a ray at t0 = max(t_box_near, t_plane) and needs t0 < t_box_far, so every
// fragment's ray holds a point of that clipped box), as seen through the ray generator of the kernel: pixel (px, py) looks along
// dir00 + (px + 0.5) ddx + (py + 0.5) ddy from cam, so a point c is seen at the (fx, fy) with c - cam = g (dir00 + fx ddx + fy ddy), g > 0.
// The clipped box is convex and the map to (fx, fy) keeps convexity in front of the camera: the bound is the min / max over its VERTICES - the
// box corners on the kept side and the points where the plane cuts an edge (the plane is moved outwards by 1e-4 of its normal's length
// first: the device evaluates in fp32) -, widened by two pixels (the device evaluates the direction in fp32: it can disagree with this
// double-precision solve by a tiny fraction of a pixel).  A vertex at or behind the camera plane, or a degenerate generator, disables it
// (kScreenBoundNone); no vertex at all = nothing can be seen (kScreenBoundEmpty).  With the application's plane (through a point in front of the
// camera, facing away from it) every vertex lies in front of the camera, also for a camera inside the box: round 6 - the bound of the
// un-clipped box (rounds 2-5) gave up there.  Pixels outside cannot have a fragment, whatever the depth test does afterwards.
enum
{
	kScreenBoundNone  = 0,
	kScreenBoundEmpty = 1,
	kScreenBoundRect  = 2
};
int screen_bound(const float cam[3], const float dir00[3], const float ddx[3], const float ddy[3], const float plane_tex[4], double out[4])
{
	// inverse of M = [ddx ddy dir00] (columns) by the adjugate
	const double M[3][3] = {{ddx[0], ddy[0], dir00[0]}, {ddx[1], ddy[1], dir00[1]}, {ddx[2], ddy[2], dir00[2]}};
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
		return kScreenBoundNone;
	// vertices of the clipped box
	double    vert[8 + 12][3];
	int       n_vert = 0;
	double    sd[8];        // signed plane values of the corners (+ the outward shift)
	bool      clip = plane_tex != nullptr;
	if (clip)
	{
		const double pn = std::sqrt((double) plane_tex[0] * plane_tex[0] + (double) plane_tex[1] * plane_tex[1] + (double) plane_tex[2] * plane_tex[2]);
		if (!std::isfinite(pn) || !(pn > 0.0) || !std::isfinite((double) plane_tex[3]))
			clip = false;        // no usable plane: the whole box
		for (int c = 0; c < 8 && clip; ++c)
			sd[c] = (double) plane_tex[0] * (c & 1) + (double) plane_tex[1] * ((c >> 1) & 1) + (double) plane_tex[2] * ((c >> 2) & 1) + (double) plane_tex[3] + 1e-4 * pn;
	}
	for (int c = 0; c < 8; ++c)
		if (!clip || sd[c] >= 0.0)
			vert[n_vert][0] = (double) (c & 1), vert[n_vert][1] = (double) ((c >> 1) & 1), vert[n_vert][2] = (double) ((c >> 2) & 1), ++n_vert;
	if (clip)
		for (int c = 0; c < 8; ++c)
			for (int axis = 0; axis < 3; ++axis)
			{
				const int d = c | (1 << axis);
				if (d == c || (sd[c] >= 0.0) == (sd[d] >= 0.0))
					continue;        // (each edge once: from its corner with the axis bit clear) the plane does not cut this edge
				const double u = sd[c] / (sd[c] - sd[d]);
				for (int k = 0; k < 3; ++k)
					vert[n_vert][k] = (double) ((c >> k) & 1) + u * ((double) ((d >> k) & 1) - (double) ((c >> k) & 1));
				++n_vert;
			}
	if (n_vert == 0)
		return kScreenBoundEmpty;
	double lo_x = 1e300, hi_x = -1e300, lo_y = 1e300, hi_y = -1e300;
	for (int i = 0; i < n_vert; ++i)
	{
		const double v[3] = {vert[i][0] - cam[0], vert[i][1] - cam[1], vert[i][2] - cam[2]};
		const double fa = (inv[0][0] * v[0] + inv[0][1] * v[1] + inv[0][2] * v[2]) / det, fb = (inv[1][0] * v[0] + inv[1][1] * v[1] + inv[1][2] * v[2]) / det;
		const double g  = (inv[2][0] * v[0] + inv[2][1] * v[1] + inv[2][2] * v[2]) / det;
		if (!(g > 1e-6) || !std::isfinite(fa) || !std::isfinite(fb))
			return kScreenBoundNone;        // a vertex beside or behind the camera: its projection says nothing
		lo_x = std::min(lo_x, fa / g), hi_x = std::max(hi_x, fa / g), lo_y = std::min(lo_y, fb / g), hi_y = std::max(hi_y, fb / g);
	}
	// pixel p is sampled at p + 0.5
	out[0] = std::floor(lo_x - 0.5) - 2.0, out[1] = std::ceil(hi_x - 0.5) + 2.0, out[2] = std::floor(lo_y - 0.5) - 2.0, out[3] = std::ceil(hi_y - 0.5) + 2.0;
	if (out[1] < 0.0 || out[3] < 0.0 || out[0] > 4.0e9 || out[2] > 4.0e9)
		return kScreenBoundEmpty;        // off screen
	return kScreenBoundRect;
}

static void screen_bound_of_box(RayMarchArgs &a, const VkvTuning &T)
{
	a.cull_x0 = 0u, a.cull_x1 = ~0u, a.cull_y0 = 0u, a.cull_y1 = ~0u;
	if (!T.screen_cull)        // A/B switch
		return;
	double    b[4];
	const int kind = screen_bound(a.cam, a.dir00, a.ddx, a.ddy, a.plane_tex, b);
	if (kind == kScreenBoundNone)
		return;
	if (kind == kScreenBoundEmpty)
	{        // nothing of the clipped box is on screen: an empty bound
		a.cull_x0 = 1u, a.cull_x1 = 0u;
		return;
	}
	a.cull_x0 = b[0] <= 0.0 ? 0u : (uint32_t) b[0], a.cull_y0 = b[2] <= 0.0 ? 0u : (uint32_t) b[2];
	a.cull_x1 = b[1] >= 4.0e9 ? ~0u : (uint32_t) b[1], a.cull_y1 = b[3] >= 4.0e9 ? ~0u : (uint32_t) b[3];
}

// vkv_screen_tile_rect: the same bound in whole tiles
void screen_tile_rect(const VkvRayCastUniform *rc, const VkvRayGen *rg, uint32_t iw, uint32_t ih, uint32_t tw, uint32_t th, uint32_t align, VkvTileRect *out)
{
	const uint32_t tiles_x = (iw + tw - 1) / tw, tiles_y = (ih + th - 1) / th;
	*out = VkvTileRect{0u, 0u, tiles_x, tiles_y};
	double    b[4];
	const int kind = screen_bound(rc->camera_pos_tex, rg->dir00, rg->ddx, rg->ddy, rc->plane_tex, b);
	if (kind == kScreenBoundNone)
		return;
	if (kind == kScreenBoundEmpty || b[0] >= (double) iw || b[2] >= (double) ih)
	{
		*out = VkvTileRect{0u, 0u, 1u, 1u};
		return;
	}
	const uint32_t x0 = b[0] <= 0.0 ? 0u : (uint32_t) b[0], y0 = b[2] <= 0.0 ? 0u : (uint32_t) b[2];
	const uint32_t x1 = b[1] >= (double) (iw - 1) ? iw - 1 : (uint32_t) b[1], y1 = b[3] >= (double) (ih - 1) ? ih - 1 : (uint32_t) b[3];        // inclusive
	const uint32_t q = align > 1 ? align : 1u;
	uint32_t       tx0 = (x0 / tw) / q * q, ty0 = (y0 / th) / q * q;
	uint32_t       tx1 = std::min(tiles_x, (x1 / tw + q) / q * q), ty1 = std::min(tiles_y, (y1 / th + q) / q * q);        // exclusive
	*out = VkvTileRect{tx0, ty0, tx1 - tx0, ty1 - ty0};
}

// VkvRenderParams -> kernel arguments.  Returns VKV_OK with a.nblocks == 0 when the schedule is empty.
int fill_render_args(vkv_ctx *ctx, const VkvRenderParams *P, const float *alpha_lut, RayMarchArgs &a, hipStream_t s, const VkvTuning &T, bool setup, bool batch)
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
	a.tile_first = P->tiles.tile_first, a.tile_stride = P->tiles.tile_stride, a.tile_count = P->tiles.tile_count, a.compact = P->tiles.compact;
	a.fill_rgba8_rows = (a.out_rgba8 && !a.out_color && !a.out_counts && !a.out_depth && !a.blend && (a.img_w & 3u) == 0 && (((uintptr_t) a.out_rgba8) & 15u) == 0) ? 1u : 0u;
	bool whole_schedule;        // every tile of the image is scheduled (no rectangle, or a fill_outside rectangle handed to a resident-wave scheduler)
	{        // the schedule's tile rectangle (all zero: the whole image); tiles are numbered row-major inside it
		const VkvTileRect &r = P->tiles.rect;
		const uint32_t     full_x = (a.img_w + a.tile_w - 1) / a.tile_w, full_y = (a.img_h + a.tile_h - 1) / a.tile_h;
		bool               whole  = r.w == 0 || r.h == 0;
		a.fill_tiles = 0, a.img_tiles_x = full_x, a.rect_tx0 = a.rect_ty0 = a.rect_th = 0;
		if (!whole && P->tiles.fill_outside)
		{
			// fill_outside (checked by check_render_params: the whole rectangle, image-indexed outputs): the workgroups of a vkv_render_batch launch
			// fill the tiles outside the rectangle themselves; vkv_render (argument block by value: lean_block) and the resident-wave schedulers
			// (A/B switches) simply render the whole-image schedule - the same frame either way
			if (!batch || T.scheduler == 1 || T.batch_mode == 1 || (uint64_t) full_x * full_y - (uint64_t) r.w * r.h > (uint64_t) kFillPerTile * r.w * r.h)
				whole = true, a.tile_count = full_x * full_y;        // (also a rectangle so small that its tiles could not fill the rest: a cheap frame anyway)
			else
				a.fill_tiles = full_x * full_y - r.w * r.h, a.rect_tx0 = r.x0, a.rect_ty0 = r.y0, a.rect_th = r.h;
		}
		a.tiles_x = whole ? full_x : r.w;
		a.org_x = whole ? 0u : r.x0 * a.tile_w, a.org_y = whole ? 0u : r.y0 * a.tile_h;
		whole_schedule = whole;
	}
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
	a.tile_cost = nullptr, a.order_out = nullptr;        // start-order feedback: attached by the launchers
	a.trace       = reinterpret_cast<unsigned long long *>(ctx->d_trace);
	a.back        = (int) std::ceil(P->transfer_function.sampling_factor);
	a.clamp_always = T.clamp_always != 0 ? 1u : 0u;
	{        // shape of a wave's pixel patch: voxels per pixel step in x against y (texture-space ray increments x the extent)
		double vx = 0.0, vy = 0.0;
		const double dim[3] = {(double) a.W, (double) a.H, (double) a.D};
		for (int k = 0; k < 3; ++k)
			vx += (double) a.ddx[k] * dim[k] * (double) a.ddx[k] * dim[k], vy += (double) a.ddy[k] * dim[k] * (double) a.ddy[k] * dim[k];
		const double r = (vx > 0.0 && vy > 0.0) ? std::sqrt(vx / vy) : 1.0;
		a.wave_pw_log2 = T.wave_shape == 4 ? 2u : (T.wave_shape == 8 ? 3u : (T.wave_shape == 16 ? 4u : (r >= 1.6 ? 2u : (r <= 1.0 / 1.6 ? 4u : 3u))));
	}
	{
		const float m[3] = {(float) a.mw, (float) a.mh, (float) a.md};
		for (int k = 0; k < 3; ++k)
			a.mapf[k] = m[k], a.mapb[k] = std::nextafterf(m[k], 0.0f);
	}
	a.addr_lut = nullptr, a.lut_y = a.lut_z = a.lut_words = 0;
	if (a.packed && T.address_tables != 0)
		a.addr_lut = packed_addr_lut(ctx, a.W, a.H, a.D, &a.lut_y, &a.lut_z, &a.lut_words, s, setup);
	screen_bound_of_box(a, T);
	// Start order.  Whole-image schedules: centre of the image first (a cached table per schedule shape).  A schedule that holds every tile of a
	// tile rectangle computes the same kind of order in the kernel (start_entry: the rectangle's rings from the innermost outwards) - a table per
	// rectangle SIZE does not pay when the camera moves (every frame in flight its own 16 KB table through the scalar cache: +2 ... 4 % per frame
	// with 24 in flight, profiles/r6_rect_schedules.txt).  A rank's share of a rectangle (tile_stride > 1) starts in plain order.  A registered
	// target's measured order applies to all of them (apply_feedback).
	a.tile_order = (T.tile_order_linear || !whole_schedule) ? nullptr
	                                                         : tile_start_order(ctx, a.img_w, a.img_h, a.tile_w, a.tile_h, a.tile_first, a.tile_stride, a.tile_count, s, setup);
	a.order_h = 0;
	if (!whole_schedule && !T.tile_order_linear && a.tile_first == 0 && a.tile_stride == 1 && (uint64_t) a.tile_count == (uint64_t) a.tiles_x * P->tiles.rect.h)
		a.order_h = P->tiles.rect.h;
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
		    e->stride == a.tile_stride && e->count == a.tile_count && e->org_x == a.org_x && e->org_y == a.org_y && e->tiles_x == a.tiles_x)
		{
			f = e;
			break;
		}
	if (!f)
		return false;        // not registered (or registered for another schedule): centre-first
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
	// Round 6: a target is only measured again while its camera holds still or moves slowly - this frame's view within the same ~12 degrees
	// of the PREVIOUS frame into the target.  A camera that moves fast between a target's frames (24 targets in flight and one degree per
	// frame: 6 - 72 degrees from one frame into a target to the next) cannot use a measured order, and measuring cost it 0.3 - 2.9 % (the
	// cost atomics of every measured frame, the sort behind it: profiles/r6_feedback_moving_camera.txt); now it pays for the first frame only.
	float prev_cos = 0.0f, prev_moved2 = 0.0f, prev_dist2 = 0.0f;
	for (int i = 0; i < 3; ++i)
	{
		prev_cos += dir[i] * f->prev_dir[i];
		prev_moved2 += (a.cam[i] - f->prev_pos[i]) * (a.cam[i] - f->prev_pos[i]);
		prev_dist2 += (f->prev_pos[i] - 0.5f) * (f->prev_pos[i] - 0.5f);
	}
	const bool steady = f->has_prev && prev_cos >= 0.978f && prev_moved2 <= 0.04f * prev_dist2;
	for (int i = 0; i < 3; ++i)
		f->prev_dir[i] = dir[i], f->prev_pos[i] = a.cam[i];
	f->has_prev        = true;
	const bool measure = !f->has_cost || (since >= f->period && steady);
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
	const int    rc = fill_render_args(ctx, P, alpha_lut, a, s, T, false, false);
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
	int        rc2 = launch_single(ctx, P->options.skipping_type, ert, sched, T, grad, a, s);
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
		int rc = fill_render_args(ctx, &P[i], lut, a, s, T, true, false);
		if (rc == VKV_OK && P[i].tiles.fill_outside)        // (a fill_outside schedule starts its tiles in another order in a batch launch: both tables)
			rc = fill_render_args(ctx, &P[i], lut, a, s, T, true, true);
		if (rc != VKV_OK)
			return rc;
		load_render_code(P[i].options.skipping_type, P[i].options.early_ray_termination != 0);
	}
	load_feedback_code();
	if (!stream_scratch(ctx, s, true))
		return VKV_E_UNSUPPORTED;
	return VKV_OK;
}

// ---- several frames in one launch (vkv_render_batch) --------------------------------------------------------------
int launch_render_batch(vkv_ctx *ctx, const VkvRenderParams *P, uint32_t n, const float *alpha_luts, hipStream_t s)
{
	// the argument blocks go through this stream's scratch buffer: an earlier batch on the same stream has finished with it by the
	// time the copy (same stream) runs
	static_assert(kMaxBatch * sizeof(RayMarchArgs) <= kScratchBytes - kBatchArgsOffset, "batch argument blocks must fit the stream scratch");
	static_assert(kPullHeadsBytes + kMaxBatch * sizeof(RayMarchArgs) <= kCaptureSlotBytes, "a captured launch's upload must fit its pinned slot");
	const VkvTuning           T = tuning_of(ctx);
	std::vector<RayMarchArgs> host(n);
	for (uint32_t i = 0; i < n; ++i)
	{
		const int rc = fill_render_args(ctx, &P[i], alpha_luts + (size_t) i * 256, host[i], s, T, false, true);
		if (rc != VKV_OK)
			return rc;
		if (!host[i].packed)
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render_batch: frame %u has no packed sampling image (d_packed_volume)", i);
		// the frames of a launch share the tile SIZE; their tile counts may differ (every frame of a multi-GPU launch has its own screen
		// rectangle): the grid is sized for the largest, a frame's surplus workgroups leave at once
		if (host[i].blocks_per_tile != host[0].blocks_per_tile)
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render_batch: frame %u has a different tile size than frame 0", i);
		if (ctx->d_debug_orders && i < ctx->debug_order_frames && ctx->debug_order_count == host[i].tile_count)
			host[i].tile_order = ctx->d_debug_orders + (size_t) i * ctx->debug_order_count;        // diagnostic start orders
	}
	uint32_t max_tiles = 0, ref = 0;        // ref: the first frame with tiles (an empty frame's argument block is only filled up to its tile count)
	bool     same_count = true;
	for (uint32_t i = n; i-- > 0;)
	{
		max_tiles = std::max(max_tiles, host[i].nblocks ? host[i].tile_count : 0u);
		if (host[i].nblocks)
			ref = i;
	}
	if (max_tiles == 0)
		return VKV_OK;
	for (uint32_t i = 0; i < n; ++i)
		same_count = same_count && host[i].nblocks != 0 && host[i].tile_count == max_tiles;
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
	const void *upload_src = upload.data();
	// a capture slot taken below returns to the context on every error path: only a launch that was recorded keeps it
	struct SlotGuard
	{
		vkv_ctx *ctx;
		size_t   index;
		bool     armed;
		~SlotGuard()
		{
			if (!armed)
				return;
			std::lock_guard<std::mutex> lock(ctx->mutex);
			if (index < ctx->capture_slots.size())
				ctx->capture_slots[index].in_use = false, ctx->capture_slots[index].owner = nullptr;
		}
	} slot_guard{ctx, 0, false};
	{        // A stream that is being captured into a hipGraph records the copy's SOURCE POINTER and reads it at every replay: the block then has to
		 // outlive the call - a pinned slot the context keeps until vkv_trim / vkv_destroy
		hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
		if (hipStreamIsCapturing(s, &capture) == hipSuccess && capture == hipStreamCaptureStatusActive)
		{
			// A captured launch gets a slot of its own - pinned source AND device target of the upload: the graph can be replayed on any stream,
			// concurrently with other graphs and with live launches on the capture stream, so it must not go through that stream's scratch block
			// (whose users are serialised by the stream, which a replay elsewhere is not).  vkv_create set kCaptureSlots slots aside, so that the
			// usual capture allocates nothing; beyond them a slot is allocated here, with this thread's capture mode relaxed for the calls (an
			// allocation under the global / thread-local mode would invalidate the capture).  Slots return with vkv_release_captured(stream),
			// vkv_trim and vkv_destroy.
			std::lock_guard<std::mutex> lock(ctx->mutex);
			vkv_ctx::CaptureSlot *      slot = nullptr;
			for (auto &c : ctx->capture_slots)
				if (!c.in_use)
				{
					slot = &c;
					break;
				}
			if (!slot)
			{
				vkv_ctx::CaptureSlot c;
				void *               hp = nullptr, *dp = nullptr;
				hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
				(void) hipThreadExchangeStreamCaptureMode(&mode);
				const hipError_t ea = hipHostMalloc(&hp, kCaptureSlotBytes, hipHostMallocDefault);
				const hipError_t eb = ea == hipSuccess ? hipMalloc(&dp, kCaptureSlotBytes) : ea;
				const bool failed = ea != hipSuccess || eb != hipSuccess || !hp || !dp;
				if (failed && hp)
					(void) hipHostFree(hp);        // still inside the relaxed window: a free under the capture's own mode could invalidate it
				(void) hipThreadExchangeStreamCaptureMode(&mode);
				if (failed)
					return set_error(ctx, VKV_E_UNSUPPORTED, "render_batch: no argument block for a captured launch: %s", hipGetErrorString(ea != hipSuccess ? ea : eb));
				c.pinned = static_cast<uint8_t *>(hp), c.device = static_cast<uint8_t *>(dp);
				ctx->capture_slots.push_back(c);
				slot = &ctx->capture_slots.back();
			}
			slot->in_use = true, slot->owner = s;
			slot_guard.index = (size_t) (slot - ctx->capture_slots.data()), slot_guard.armed = true;
			std::memcpy(slot->pinned, upload.data(), upload.size());
			upload_src = slot->pinned;
			d_heads    = reinterpret_cast<uint32_t *>(slot->device);
			d_frames   = reinterpret_cast<RayMarchArgs *>(slot->device + kPullHeadsBytes);
		}
	}
	const hipError_t e = hipMemcpyAsync(d_heads, upload_src, upload.size(), hipMemcpyHostToDevice, s);
	if (e != hipSuccess)
		return set_error(ctx, (int) e, "render_batch: argument upload: %s", hipGetErrorString(e));
	const uint64_t grid = (uint64_t) ((max_tiles + 7u) / 8u) * 8u * host[0].blocks_per_tile * n;
	if (grid > 0x7fffffffull)
		return set_error(ctx, VKV_E_UNSUPPORTED, "render_batch: too many workgroups for one launch");
	const bool ert  = P[0].options.early_ray_termination != 0;
	const int  grad = !P[0].transfer_function.use_gradient ? 0 : (P[0].use_precomputed_gradient ? 1 : 2);
	// frames interleaved in groups of eight workgroups (default) or one frame after the other (VkvTuning.batch_sequential, A/B
	// switch: measured 0.183 vs 0.169 ms per frame on C3 with 8 frames per launch)
	const bool        sequential = T.batch_sequential != 0;
	const uint32_t    gpf        = sequential ? (uint32_t) (grid / n / 8) : 0u;
	LeanChoice        choice     = choose_lean(host[ref], T);
	bool              no_counts  = true;        // no frame reads the per-pixel counters (no counter buffer, no sample-count test output): the loop without them
	for (uint32_t i = 0; i < n; ++i)
		no_counts = no_counts && (host[i].nblocks == 0 || !wants_counts(host[i]));
	for (uint32_t i = 0; i < n; ++i)
	{        // one kernel for all frames: the most general choice any of them needs
		if (i == ref || host[i].nblocks == 0)
			continue;
		const LeanChoice c = choose_lean(host[i], T);
		if (c.kind != choice.kind || c.lds != choice.lds || host[i].lut_words != host[ref].lut_words)
			choice = {0, 0};
	}
	// VkvTuning.batch_mode = 1 (pull): resident workgroups whose waves pull 8x8 units from per-XCD ticket counters (k_raymarch_lean_pull), possible
	// when every frame shares the LDS tables (same packed image and extents, TF tables, opacity table).  Bit-identical; measured on C3 with 8
	// frames per launch: the CU stays full (7 900 of 8 192 wave slots against 5 300) and everything but the last marching tiles is done after
	// 0.83 ms instead of 1.0, but those last tiles - the volume's silhouette, 100-250 iterations of cold probes - then run 340 us with their
	// four 8x8 units on four different CUs (150 us as one workgroup on one CU): 0.151 ms per frame against 0.137.  Not the default.
	bool              pull      = T.batch_mode == 1 && !sequential && same_count;        // (its tickets count one tile count for every frame)
	const uint64_t    units     = (uint64_t) max_tiles * host[0].blocks_per_tile * 4u * n;
	for (uint32_t i = 1; i < n && pull; ++i)
	{
		const RayMarchArgs &a = host[i], &b = host[0];
		pull = a.packed == b.packed && a.tf_bits == b.tf_bits && a.tf == b.tf && a.addr_lut == b.addr_lut && a.W == b.W && a.H == b.H && a.D == b.D &&
		       std::memcmp(a.alpha_lut, b.alpha_lut, sizeof(a.alpha_lut)) == 0;
	}
	if (pull)
	{
		if (!launch_pull(ctx, P[0].options.skipping_type, ert, grad, d_frames, n, d_heads, choice, units, s))
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render_batch: bad skipping_type %d", P[0].options.skipping_type);
		if (any_sort)
			hipLaunchKernelGGL(k_tile_orders_from_cost, dim3(n), dim3(256), 0, s, d_frames);
		const int rc_pull = check_launch(ctx, "render_batch");
		slot_guard.armed = slot_guard.armed && rc_pull != VKV_OK;
		return rc_pull;
	}
	if (!launch_batch(P[0].options.skipping_type, ert, grad, d_frames, n, (uint32_t) grid, gpf, choice, no_counts, s))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render_batch: bad skipping_type %d", P[0].options.skipping_type);
	// behind the render, on the same stream: the costs it measured become the start order of the next frames into these targets (the
	// sort is not in front of anybody's render this way; in front it cost 70 us per 20-frame block of three launches)
	if (any_sort)
		hipLaunchKernelGGL(k_tile_orders_from_cost, dim3(n), dim3(256), 0, s, d_frames);
	const int rc_launch = check_launch(ctx, "render_batch");
	slot_guard.armed = slot_guard.armed && rc_launch != VKV_OK;
	return rc_launch;
}

}        // namespace vkv
