// raymarch_core.hpp — device code of the ray-march integrator with block / Chebyshev / anisotropic-Chebyshev empty-space
// skipping and early ray termination, for gfx950.  Included by the product's translation units (raymarch.hip: argument set-up and
// dispatch; raymarch_inst.hpp + raymarch_s<skip>e<ert>.hip: the kernel instantiations, one file per (skipping type, early ray termination) pair) and by
// tools/lab/raymarch_lab.hpp (experimental and retired variants: they live there, not here).
//
// Replaces VolumeRenderSubpass::draw (src/volume_render_subpass.cpp:159-294) and the shaders it binds:
// shaders/volume_render.frag (integrator), shaders/transfer_function.glsl (get_color) and the two vertex
// shaders (ray entry; here an analytic per-pixel box / clip-plane intersection, there is no rasteriser on CDNA).
//
// The frag's compile-time variants (volume_render_subpass.cpp:57-92) are template parameters.  The kernels share the per-ray
// set-up and finish (ray_setup / ray_finish):
//   k_raymarch_lean        the product's kernel: one lane per ray, wave = 8x8 pixels, workgroup = 16x16; the frag's loop body
//                          (frag:215-312) as two EXEC-masked blocks (probe outcome / sample outcome) behind loads issued ahead of
//                          both (lean_march)
//   k_raymarch_lean_batch  the same workgroup body for several frames in one launch (vkv_render_batch)
//   k_raymarch_lean_pull   resident workgroups whose waves pull 8x8 units from per-XCD ticket counters (VkvTuning.batch_mode = 1)
//   k_raymarch_persistent  resident waves pull 8x8 tiles from per-XCD queues and re-fill lanes whose rays have ended
//                          (ballot + mbcnt compaction, ray_event); bit-identical, slower (VkvTuning.scheduler = 1)
#pragma once

#include <type_traits>

#include "vkv_device.hpp"

using namespace vkv;

constexpr int kTraceWords = 10;        // u64 words per wave of the diagnostic trace buffer
constexpr uint32_t kFillPerTile = 3;   // VkvTileSchedule.fill_outside: outside tiles a rendering tile fills at most (lean_block)

struct RayMarchArgs
{
	// ray generator + RayCastUniform
	float dir00[3], ddx[3], ddy[3];
	float cam[3];
	float plane_tex[4];
	float block_size[3];
	// CameraUniform matrices needed for gl_FragDepth (frag:319)
	float model[16], view[16], proj[16];
	float view_proj_inv[16], model_inv[16];        // DEPTH_ATTACHMENT only (frag:154-156)
	// TransferFunctionUniform
	float sampling_factor, grad_modifier;
	// extents
	int W, H, D, mw, mh, md;
	const uint8_t * vol, *grad, *tf;
	const uint8_t * packed;         // vkv_pack_volume image (PACKED variants) or null
	int             pmx, pmy;       // macro-bricks per axis of the packed image
	const uint32_t *tf_bits;        // vkv_transfer_function_tables buffer (alpha>0 bit table, flags, separable alpha tables) or null
	const uint8_t * maps[8];
	float *         out_color;
	uint8_t *       out_rgba8;
	uint32_t *      out_counts;
	float *         out_depth;
	const float *   in_depth;           // scene depth (options.depth_attachment) or null
	uint32_t        depth_attachment, blend;
	uint32_t        img_w, img_h, tile_w, tile_h, tiles_x, tile_first, tile_stride, tile_count, compact;        // tiles_x: tile columns of the schedule's rectangle
	uint32_t        org_x, org_y;   // first pixel column / row of the schedule's tile rectangle (VkvTileSchedule.rect; 0, 0 = the whole image)
	// VkvTileSchedule.fill_outside (k_raymarch_lean*): fill_tiles = tiles of the image OUTSIDE the rectangle (0 = nothing to fill); the rendering
	// workgroups write the no-fragment result there, schedule entry k the outside tiles k, k + tile_count, ...
	uint32_t        fill_tiles, img_tiles_x, rect_tx0, rect_ty0, rect_th;
	uint32_t        fill_rgba8_rows;        // != 0: the only output is RGBA8, nothing is blended, image rows are 16-byte aligned - outside tiles are cleared with 16-byte stores
	uint32_t        blocks_per_tile_x, blocks_per_tile, nblocks;
	int             test;
	unsigned long long *trace;      // diagnostic (tools/wave_trace.py): kTraceWords x u64 per wave {t_start, t_end, iterations, unit, phase sums}, or null
	int             back;           // ceil(sampling_factor): the step back after a probe that found an occupied cell (frag:253)
	uint32_t        clamp_always;   // k_raymarch_lean: 1 = no clamp-free march loop (VkvTuning.clamp_always: A/B switch, same bits)
	float           mapf[3], mapb[3];        // k_raymarch_lean, clamp-free loop: the map extent as floats, and the largest floats below them
	uint32_t        wave_pw_log2;            // k_raymarch_lean: log2 of the width in pixels of a wave's 64-pixel patch (2, 3, 4: 4x16, 8x8, 16x4)
	const uint32_t *addr_lut;       // k_raymarch_lean: per-axis byte offsets of the packed image (packed_addr_lut, vkv_device.hpp), or null
	uint32_t        lut_y, lut_z, lut_words;        // word offsets of the y and z tables inside addr_lut and its total length
	uint32_t        cull_x0, cull_x1, cull_y0, cull_y1;        // k_raymarch_lean: pixels outside [x0, x1] x [y0, y1] cannot see the volume's box
	                                                           // (conservative screen bound from the launcher); 0, ~0, 0, ~0 = no bound
	const uint32_t *tile_order;     // k_raymarch_lean: the r-th tile to be started is schedule entry tile_order[r] (centre of the image first), or null
	// no tile_order and order_h != 0: the start order is computed (start_entry) - the rings of the schedule's tiles_x x order_h rectangle of tiles from the
	// innermost outwards (schedules that hold every tile of their rectangle)
	uint32_t        order_h;
	uint32_t *      queue_heads;        // persistent scheduler: 8 tile-queue heads (one per XCD label), zeroed per launch
	// start-order feedback (raymarch.hip, TileFeedback): every marching wave leaves max(its iteration count) in tile_cost[schedule entry];
	// before the next frame into the same target k_tile_order_from_cost turns the costs into a longest-first order (order_out = the
	// buffer tile_order then points to).  Both null when unused.
	uint32_t *      tile_cost;
	uint32_t *      order_out;
	float           alpha_lut[256];     // opacity correction keyed by the TF alpha byte (frag:283)
};

// Per-lane ray state (everything main() of the frag keeps across loop iterations).
struct Ray
{
	float          ex, ey, ez;         // ray_entry
	float          sx, sy, sz;         // step_volume
	float          six, siy, siz;      // step_dist_texel_inv (frag:195)
	const uint8_t *dmap;               // distance map of this ray (anisotropic: chosen by direction octant, frag:209)
	int            n_steps, i, i_min;
	int            ulx, uly, ulz;      // u_last_alpha
	uint32_t       ul;                 // u_last_alpha as a linear cell index (k_raymarch_er)
	int            first_hit;
	bool           occupied;
	float          r, g, b, a;         // out_color
	float          depth;
	uint32_t       n_vol, n_dist, n_empty;
	uint32_t       o;                  // output index of the pixel
	bool           fragment;           // false: no fragment for this pixel (not covered, or discarded by the depth test)
};

// Linear filter, clamp-to-edge (sampler: src/volume_component.cpp:139-148); see DESIGN.md "Pinned numerics".
__device__ __forceinline__ float sample_linear(const uint8_t *__restrict__ tex, int W, int H, int D, float px, float py, float pz)
{
	const float cx = __builtin_fmaf(px, (float) W, -0.5f), cy = __builtin_fmaf(py, (float) H, -0.5f), cz = __builtin_fmaf(pz, (float) D, -0.5f);
	const float fx = __builtin_floorf(cx), fy = __builtin_floorf(cy), fz = __builtin_floorf(cz);
	const float wx = cx - fx, wy = cy - fy, wz = cz - fz;
	const int   ix = (int) fx, iy = (int) fy, iz = (int) fz;
	const int   x0 = i_clamp(ix, 0, W - 1), x1 = i_clamp(ix + 1, 0, W - 1);
	const int   y0 = i_clamp(iy, 0, H - 1), y1 = i_clamp(iy + 1, 0, H - 1);
	const int   z0 = i_clamp(iz, 0, D - 1), z1 = i_clamp(iz + 1, 0, D - 1);
	const size_t r00 = ((size_t) z0 * (size_t) H + (size_t) y0) * (size_t) W, r10 = ((size_t) z0 * (size_t) H + (size_t) y1) * (size_t) W;
	const size_t r01 = ((size_t) z1 * (size_t) H + (size_t) y0) * (size_t) W, r11 = ((size_t) z1 * (size_t) H + (size_t) y1) * (size_t) W;
	const float b000 = tex[r00 + x0], b100 = tex[r00 + x1];
	const float b010 = tex[r10 + x0], b110 = tex[r10 + x1];
	const float b001 = tex[r01 + x0], b101 = tex[r01 + x1];
	const float b011 = tex[r11 + x0], b111 = tex[r11 + x1];
	const float c00 = __builtin_fmaf(wx, b100 - b000, b000), c10 = __builtin_fmaf(wx, b110 - b010, b010);
	const float c01 = __builtin_fmaf(wx, b101 - b001, b001), c11 = __builtin_fmaf(wx, b111 - b011, b011);
	const float c0 = __builtin_fmaf(wy, c10 - c00, c00), c1 = __builtin_fmaf(wy, c11 - c01, c01);
	return __builtin_fmaf(wz, c1 - c0, c0) * kInv255;
}

// Same filter on the packed image: the whole 2x2x2 footprint of BOTH textures sits in one 256-byte brick, the x pair of
// a row is one (2-byte aligned) dword = (v0, g0, v1, g1).  Arithmetic identical to sample_linear, so results are too.
typedef uint32_t u32_align2 __attribute__((aligned(2)));

// address of the footprint's first dword + the three filter weights
__device__ __forceinline__ const uint8_t *packed_footprint(const uint8_t *__restrict__ P, int W, int H, int D, int pmx, int pmy, float px, float py, float pz,
                                                           float &wx, float &wy, float &wz)
{
	const float cx = __builtin_fmaf(px, (float) W, -0.5f), cy = __builtin_fmaf(py, (float) H, -0.5f), cz = __builtin_fmaf(pz, (float) D, -0.5f);
	const float fx = __builtin_floorf(cx), fy = __builtin_floorf(cy), fz = __builtin_floorf(cz);
	wx = cx - fx, wy = cy - fy, wz = cz - fz;
	const int bx = i_clamp((int) fx, -1, W) + 1, by = i_clamp((int) fy, -1, H) + 1, bz = i_clamp((int) fz, -1, D) + 1;
	// 32-bit brick index (macro-brick * 512 + brick-in-macro), one 64-bit shift for the byte offset
	const uint32_t macro = ((uint32_t) (bz >> 5) * (uint32_t) pmy + (uint32_t) (by >> 5)) * (uint32_t) pmx + (uint32_t) (bx >> 5);
	const uint32_t sub   = (uint32_t) ((((bz >> 2) & 7) << 6) | (((by >> 2) & 7) << 3) | ((bx >> 2) & 7));
	const uint32_t in    = (uint32_t) ((((bz & 3) * 5 + (by & 3)) * 5 + (bx & 3)) * 2);
	return P + (((uint64_t) (macro * 512u + sub)) << 8) + in;
}

// the four x-pair dwords (v0, g0, v1, g1) of rows (y0,z0), (y1,z0), (y0,z1), (y1,z1) -> filtered volume (and gradient) value
template <bool WANT_G>
__device__ __forceinline__ void packed_filter(uint32_t q00, uint32_t q10, uint32_t q01, uint32_t q11, float wx, float wy, float wz, float &out_v, float &out_g)
{
	{
		const float b000 = (float) (q00 & 255u), b100 = (float) ((q00 >> 16) & 255u);
		const float b010 = (float) (q10 & 255u), b110 = (float) ((q10 >> 16) & 255u);
		const float b001 = (float) (q01 & 255u), b101 = (float) ((q01 >> 16) & 255u);
		const float b011 = (float) (q11 & 255u), b111 = (float) ((q11 >> 16) & 255u);
		const float c00 = __builtin_fmaf(wx, b100 - b000, b000), c10 = __builtin_fmaf(wx, b110 - b010, b010);
		const float c01 = __builtin_fmaf(wx, b101 - b001, b001), c11 = __builtin_fmaf(wx, b111 - b011, b011);
		const float c0 = __builtin_fmaf(wy, c10 - c00, c00), c1 = __builtin_fmaf(wy, c11 - c01, c01);
		out_v = __builtin_fmaf(wz, c1 - c0, c0) * kInv255;
	}
	if (WANT_G)
	{
		const float b000 = (float) ((q00 >> 8) & 255u), b100 = (float) (q00 >> 24);
		const float b010 = (float) ((q10 >> 8) & 255u), b110 = (float) (q10 >> 24);
		const float b001 = (float) ((q01 >> 8) & 255u), b101 = (float) (q01 >> 24);
		const float b011 = (float) ((q11 >> 8) & 255u), b111 = (float) (q11 >> 24);
		const float c00 = __builtin_fmaf(wx, b100 - b000, b000), c10 = __builtin_fmaf(wx, b110 - b010, b010);
		const float c01 = __builtin_fmaf(wx, b101 - b001, b001), c11 = __builtin_fmaf(wx, b111 - b011, b011);
		const float c0 = __builtin_fmaf(wy, c10 - c00, c00), c1 = __builtin_fmaf(wy, c11 - c01, c01);
		out_g = __builtin_fmaf(wz, c1 - c0, c0) * kInv255;
	}
}

template <bool WANT_G>
__device__ __forceinline__ void sample_packed(const uint8_t *__restrict__ P, int W, int H, int D, int pmx, int pmy, float px, float py, float pz,
                                              float &out_v, float &out_g)
{
	float          wx, wy, wz;
	const uint8_t *b   = packed_footprint(P, W, H, D, pmx, pmy, px, py, pz, wx, wy, wz);
	const uint32_t q00 = *reinterpret_cast<const u32_align2 *>(b);
	const uint32_t q10 = *reinterpret_cast<const u32_align2 *>(b + 10);
	const uint32_t q01 = *reinterpret_cast<const u32_align2 *>(b + 50);
	const uint32_t q11 = *reinterpret_cast<const u32_align2 *>(b + 60);
	packed_filter<WANT_G>(q00, q10, q01, q11, wx, wy, wz, out_v, out_g);
}

__device__ __forceinline__ void mat4_mul_vec4(const float *m, const float *v, float *r)
{
#pragma unroll
	for (int i = 0; i < 4; ++i)
		r[i] = __builtin_fmaf(m[12 + i], v[3], __builtin_fmaf(m[8 + i], v[2], __builtin_fmaf(m[4 + i], v[1], m[i] * v[0])));
}

__device__ __forceinline__ uint8_t quantise_rgba8(float c) { return (uint8_t) __builtin_rintf(g_clamp(c, 0.0f, 1.0f) * 255.0f); }

// ---------------------------------------------------------------------------------------------------------------
// Correctly rounded fp32 division out of v_rcp_f32 - the sequence the compiler itself emits for `a / b` under
// -fhip-fp32-correctly-rounded-divide-sqrt, without its range scaling (v_div_scale / v_div_fmas) and special-case fix-up
// (v_div_fixup), which do nothing for "ordinary" operands (both magnitudes in [2^-40, 2^40]; a ZERO numerator is not ordinary - the
// refinement would lose the sign of -0 / d - and takes the IEEE path like everything else outside the range: div_ordinary_num):
//     r0 = rcp(d); e = fma(-d, r0, 1); r = fma(e, r0, r0)                          <- depends on the denominator only
//     q0 = a * r; q1 = fma(fma(-d, q0, a), r, q0); q = fma(fma(-d, q1, a), r, q1)
// The ray set-up divides three numerators by the same length (twice), by the same step count, takes three reciprocals of a
// direction ...: 22 divisions per covered ray, 13 % of a C3 frame.  Sharing r between the quotients of one denominator and dropping
// the scale / fix-up instructions leaves 149 of their 242 instructions, with the same bits: checked on the device against the IEEE
// division for every float as denominator (vkv_debug_check what = 2: reciprocals; what = 3: quotients with hashed numerators; what = 4:
// the dispatch itself with numerators +0 and -0).
// Operands outside the ordinary range (axis-parallel rays: 1 / 0; NaNs of a degenerate camera) send the whole wave through the
// plain IEEE set-up (ray_setup below), so the fast path never has to be right about them.
// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// Ray generation + frag:147-210.  Returns true when the ray has to be marched; false when the pixel is finished
// already (not covered, grazing-ray early-out, or a RayEntry / RayExit test output) with its result in R.
// FAST: divisions through div_by / recip_exact; `ok` comes back false when an operand was outside their range (the caller then
// repeats the set-up with FAST = false for the whole wave).
// ---------------------------------------------------------------------------------------------------------------
template <int SKIP, bool FAST>
__device__ __forceinline__ bool ray_setup_impl(const RayMarchArgs &A, uint32_t px, uint32_t py, Ray &R, bool &ok)
{
	R.r = R.g = R.b = R.a = 0.0f;        // out_color = vec4(0) (frag:120)
	R.depth = 0.0f;                      // gl_FragDepth = 0 (frag:140)
	R.n_vol = R.n_dist = R.n_empty = 0;
	R.n_steps = 0, R.i = 0, R.i_min = 0, R.first_hit = 0, R.ul = 0;
	R.fragment = false;
	const int W = A.W, H = A.H, D = A.D;

	// ---- ray generation (replaces volume_render_clipped.vert + volume_render_plane_intersection.vert) ----------
	const float fx = (float) px + 0.5f, fy = (float) py + 0.5f;
	float       dx = __builtin_fmaf(fy, A.ddy[0], __builtin_fmaf(fx, A.ddx[0], A.dir00[0]));
	float       dy = __builtin_fmaf(fy, A.ddy[1], __builtin_fmaf(fx, A.ddx[1], A.dir00[1]));
	float       dz = __builtin_fmaf(fy, A.ddy[2], __builtin_fmaf(fx, A.ddx[2], A.dir00[2]));
	{
		const float len = __builtin_sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
		if (FAST)
		{
			ok = ok && div_ordinary(len) && div_ordinary_num(dx) && div_ordinary_num(dy) && div_ordinary_num(dz);
			const float r = recip_refined(len);
			dx = div_by(dx, len, r), dy = div_by(dy, len, r), dz = div_by(dz, len, r);
		}
		else
			dx /= len, dy /= len, dz /= len;
	}
	const float ox = A.cam[0], oy = A.cam[1], oz = A.cam[2];
	float       t_near = -INFINITY, t_far = INFINITY;
	bool        miss   = false;
	{
		const float dv[3] = {dx, dy, dz}, ov[3] = {ox, oy, oz};
#pragma unroll
		for (int a = 0; a < 3; ++a)
		{
			if (dv[a] == 0.0f)
			{
				if (ov[a] < 0.0f || ov[a] > 1.0f)
					miss = true;
			}
			else
			{
				float inv;
				if (FAST)
				{
					ok  = ok && div_ordinary(dv[a]);
					inv = recip_exact(dv[a]);
				}
				else
					inv = 1.0f / dv[a];
				const float ta = (0.0f - ov[a]) * inv, tb = (1.0f - ov[a]) * inv;
				t_near = g_max(t_near, g_min(ta, tb));
				t_far  = g_min(t_far, g_max(ta, tb));
			}
		}
	}
	if (miss)
		return false;
	const float Ap = __builtin_fmaf(A.plane_tex[2], oz, __builtin_fmaf(A.plane_tex[1], oy, A.plane_tex[0] * ox)) + A.plane_tex[3];
	const float Bp = __builtin_fmaf(A.plane_tex[2], dz, __builtin_fmaf(A.plane_tex[1], dy, A.plane_tex[0] * dx));
	if (!(Bp > 0.0f))
		return false;
	float t_plane;
	if (FAST)
	{
		ok      = ok && div_ordinary(Bp) && div_ordinary_num(0.0f - Ap);
		t_plane = div_by(0.0f - Ap, Bp, recip_refined(Bp));
	}
	else
		t_plane = (0.0f - Ap) / Bp;
	const float t0      = g_max(t_near, t_plane);
	if (!(t0 < t_far))
		return false;
	const float ex = __builtin_fmaf(t0, dx, ox), ey = __builtin_fmaf(t0, dy, oy), ez = __builtin_fmaf(t0, dz, oz);        // ray_entry
	R.fragment = true;

	// ---- DEPTH_ATTACHMENT, frag:122-136: manual z-test of the front face against the scene depth (reverse-Z) ----
	float frag_depth = 0.0f, frag_depth_front = 0.0f, position[4] = {0, 0, 0, 0};
	if (A.depth_attachment)
	{
		const float pm[4] = {ex - 0.5f, ey - 0.5f, ez - 0.5f, 1.0f};        // position = proj * view * model * (ray_entry - 0.5) (clipped.vert:62)
		float       a4[4], b4[4];
		mat4_mul_vec4(A.model, pm, a4);
		mat4_mul_vec4(A.view, a4, b4);
		mat4_mul_vec4(A.proj, b4, position);
		frag_depth       = A.in_depth[R.o];
		frag_depth_front = position[2] / position[3];
		if (frag_depth > frag_depth_front)
		{        // discard
			R.fragment = false;
			return false;
		}
		R.depth = frag_depth;        // gl_FragDepth = frag_depth (frag:135)
	}

	// ---- frag:147-149 --------------------------------------------------------------------------------------
	float rdx, rdy, rdz;
	{
		const float vx = ex - ox, vy = ey - oy, vz = ez - oz;
		const float len = __builtin_sqrtf(__builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx)));
		if (FAST)
		{
			ok = ok && div_ordinary(len) && div_ordinary_num(vx) && div_ordinary_num(vy) && div_ordinary_num(vz);
			const float r = recip_refined(len);
			rdx = div_by(vx, len, r), rdy = div_by(vy, len, r), rdz = div_by(vz, len, r);
		}
		else
			rdx = vx / len, rdy = vy / len, rdz = vz / len;
	}
	float xx, xy, xz, ray_distance;        // ray_exit
	{
		float ix, iy, iz;
		if (FAST)
		{
			ok = ok && div_ordinary(rdx) && div_ordinary(rdy) && div_ordinary(rdz);
			ix = recip_exact(rdx), iy = recip_exact(rdy), iz = recip_exact(rdz);
		}
		else
			ix = 1.0f / rdx, iy = 1.0f / rdy, iz = 1.0f / rdz;
		const float tminx = -ex * ix, tminy = -ey * iy, tminz = -ez * iz;
		const float tmaxx = (1.0f - ex) * ix, tmaxy = (1.0f - ey) * iy, tmaxz = (1.0f - ez) * iz;
		const float t2x = g_max(tminx, tmaxx), t2y = g_max(tminy, tmaxy), t2z = g_max(tminz, tmaxz);
		const float tFar = g_min(g_min(t2x, t2y), t2z);
		xx = __builtin_fmaf(tFar, rdx, ex), xy = __builtin_fmaf(tFar, rdy, ey), xz = __builtin_fmaf(tFar, rdz, ez);
		const float vx = ex - xx, vy = ey - xy, vz = ez - xz;
		ray_distance = __builtin_sqrtf(__builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx)));
	}
	if (A.depth_attachment)
	{        // frag:152-164: stop the ray where it meets the depth buffer
		const float clip[4] = {(position[0] * frag_depth) / frag_depth_front, (position[1] * frag_depth) / frag_depth_front,
		                       (position[2] * frag_depth) / frag_depth_front, position[3]};
		float       w4[4], m4[4];
		mat4_mul_vec4(A.view_proj_inv, clip, w4);
		w4[0] /= w4[3], w4[1] /= w4[3], w4[2] /= w4[3], w4[3] /= w4[3];
		mat4_mul_vec4(A.model_inv, w4, m4);
		const float ix = m4[0] + 0.5f, iy = m4[1] + 0.5f, iz = m4[2] + 0.5f;
		const float vx = ex - ix, vy = ey - iy, vz = ez - iz;
		const float dd = __builtin_sqrtf(__builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx)));
		if (dd < ray_distance)
		{
			xx = ix, xy = iy, xz = iz;
			ray_distance = dd;
		}
	}
	if (A.test == VKV_TEST_RAY_ENTRY)
	{
		R.r = ex, R.g = ey, R.b = ez, R.a = 1.0f;
		return false;
	}
	if (A.test == VKV_TEST_RAY_EXIT)
	{
		R.r = xx, R.g = xy, R.b = xz, R.a = 1.0f;
		return false;
	}

	// ---- frag:176-187 --------------------------------------------------------------------------------------
	const int   dim_max = max(max(W, H), D);
	const float nf      = __builtin_ceilf((float) dim_max * ray_distance * A.sampling_factor);
	if (!(nf >= 2.0f && nf <= 16777216.0f))
		return false;
	float sx, sy, sz;
	{
		const float den = nf - 1.0f, nx = rdx * ray_distance, ny = rdy * ray_distance, nz = rdz * ray_distance;
		if (FAST)
		{
			ok = ok && div_ordinary_num(nx) && div_ordinary_num(ny) && div_ordinary_num(nz);        // den is 1 .. 2^24
			const float r = recip_refined(den);
			sx = div_by(nx, den, r), sy = div_by(ny, den, r), sz = div_by(nz, den, r);
		}
		else
			sx = nx / den, sy = ny / den, sz = nz / den;
	}
	{
		const float qx = ex + sx, qy = ey + sy, qz = ez + sz;
		if (qx <= 0.0f || qy <= 0.0f || qz <= 0.0f || qx >= 1.0f || qy >= 1.0f || qz >= 1.0f)
			return false;
	}
	R.ex = ex, R.ey = ey, R.ez = ez, R.sx = sx, R.sy = sy, R.sz = sz;
	R.n_steps = (int) nf;

	// ---- frag:191-214 --------------------------------------------------------------------------------------
	R.six = R.siy = R.siz = 0.0f;
	R.dmap                = nullptr;
	if (SKIP != VKV_SKIP_NONE)
	{
		if (FAST)
		{
			const float nx = sx * (float) W, ny = sy * (float) H, nz = sz * (float) D;
			ok = ok && div_ordinary(A.block_size[0]) && div_ordinary(A.block_size[1]) && div_ordinary(A.block_size[2]) && div_ordinary_num(nx) &&
			     div_ordinary_num(ny) && div_ordinary_num(nz);
			const float tx = div_by(nx, A.block_size[0], recip_refined(A.block_size[0]));
			const float ty = div_by(ny, A.block_size[1], recip_refined(A.block_size[1]));
			const float tz = div_by(nz, A.block_size[2], recip_refined(A.block_size[2]));
			ok    = ok && div_ordinary(tx) && div_ordinary(ty) && div_ordinary(tz);
			R.six = recip_exact(tx), R.siy = recip_exact(ty), R.siz = recip_exact(tz);
		}
		else
		{
			R.six = 1.0f / ((sx * (float) W) / A.block_size[0]);
			R.siy = 1.0f / ((sy * (float) H) / A.block_size[1]);
			R.siz = 1.0f / ((sz * (float) D) / A.block_size[2]);
		}
		if (SKIP == VKV_SKIP_ANISOTROPIC_DISTANCE)
			R.dmap = A.maps[(rdz < 0 ? 1 : 0) + (rdy < 0 ? 2 : 0) + (rdx < 0 ? 4 : 0)];
		else
			R.dmap = A.maps[0];
	}
	R.i = 0, R.i_min = 0, R.ulx = R.uly = R.ulz = 0, R.ul = 0;
	R.occupied  = true;
	R.first_hit = R.n_steps;
	return true;
}

// The set-up every kernel calls: the fast divisions when every operand of every lane of the wave is ordinary, else (axis-parallel
// rays, degenerate cameras) the plain IEEE ones for the whole wave - the same bits either way.
template <int SKIP>
__device__ __forceinline__ bool ray_setup(const RayMarchArgs &A, uint32_t px, uint32_t py, Ray &R)
{
	bool       ok      = true;
	const bool marched = ray_setup_impl<SKIP, true>(A, px, py, R, ok);
	if (__builtin_amdgcn_ballot_w64(!ok) == 0ull)
		return marched;
	bool unused = true;
	return ray_setup_impl<SKIP, false>(A, px, py, R, unused);
}

// the state ray_setup leaves behind for a pixel the volume's box does not cover
__device__ __forceinline__ void ray_clear(Ray &R)
{
	R.r = R.g = R.b = R.a = 0.0f;
	R.depth = 0.0f;
	R.n_vol = R.n_dist = R.n_empty = 0;
	R.n_steps = 0, R.i = 0, R.i_min = 0, R.first_hit = 0, R.ul = 0;
	R.fragment = false;
}

// ---------------------------------------------------------------------------------------------------------------
// One iteration of the frag's loop (frag:215-312): either one distance-map probe or one volume sample.
// Returns true when the ray has ended (ran past n_steps, or early ray termination).
// ---------------------------------------------------------------------------------------------------------------
template <int SKIP, bool ERT, int GRAD, bool PACKED>
__device__ __forceinline__ bool ray_event(const RayMarchArgs &A, Ray &R, const float *s_alpha, const float *s_unorm, const uint32_t *s_bits, bool tf_bits)
{
	const int   W = A.W, H = A.H, D = A.D;
	const int   i  = R.i;
	const float fi = (float) i;
	const float posx = __builtin_fmaf(fi, R.sx, R.ex), posy = __builtin_fmaf(fi, R.sy, R.ey), posz = __builtin_fmaf(fi, R.sz, R.ez);
	int         uix = 0, uiy = 0, uiz = 0;
	float       ux = 0, uy = 0, uz = 0;
	if (SKIP != VKV_SKIP_NONE)
	{        // frag:192, 220-221 (volume_to_distance_map_u is the same for every ray)
		const float kx = (float) W / A.block_size[0], ky = (float) H / A.block_size[1], kz = (float) D / A.block_size[2];
		ux = kx * posx, uy = ky * posy, uz = kz * posz;
		uix = i_clamp((int) ux, 0, A.mw - 1), uiy = i_clamp((int) uy, 0, A.mh - 1), uiz = i_clamp((int) uz, 0, A.md - 1);
	}
	const bool probe = SKIP != VKV_SKIP_NONE && !R.occupied && (uix != R.ulx || uiy != R.uly || uiz != R.ulz);        // frag:224

	// ---- issue phase --------------------------------------------------------------------------------------------
	// A wave usually holds probing and sampling lanes at once.  Issue this iteration's loads for BOTH kinds before either
	// is consumed (lanes of the other kind read a dummy address that every lane shares), so the probe's and the sample's
	// memory latencies overlap instead of adding up on the critical path of the wave.
	constexpr bool kHoist = PACKED && GRAD != 2 && SKIP != VKV_SKIP_NONE;
	uint32_t       dist_h = 0, q00 = 0, q10 = 0, q01 = 0, q11 = 0;
	float          hwx = 0, hwy = 0, hwz = 0;
	if (kHoist)
	{
		// a wave whose live lanes all probe (the empty space in front of the volume) or all sample skips the other kind's address
		// arithmetic and loads: wave-uniform scalar branches
		if (__ballot(probe) != 0ull)
		{
			const uint32_t cell = ((uint32_t) uiz * (uint32_t) A.mh + (uint32_t) uiy) * (uint32_t) A.mw + (uint32_t) uix;
			dist_h              = R.dmap[probe ? cell : 0u];
		}
		if (__ballot(!probe) != 0ull)
		{
			const uint8_t *fp = packed_footprint(A.packed, W, H, D, A.pmx, A.pmy, posx, posy, posz, hwx, hwy, hwz);
			const uint8_t *ba = probe ? A.packed : fp;
			q00 = *reinterpret_cast<const u32_align2 *>(ba);
			q10 = *reinterpret_cast<const u32_align2 *>(ba + 10);
			q01 = *reinterpret_cast<const u32_align2 *>(ba + 50);
			q11 = *reinterpret_cast<const u32_align2 *>(ba + 60);
		}
		// keep the five loads above the divergent consume code (the compiler would otherwise sink each into its branch)
		asm volatile("" : "+v"(dist_h), "+v"(q00), "+v"(q10), "+v"(q01), "+v"(q11));
	}

	if (probe)
	{        // frag:224-263
		++R.n_dist;
		uint32_t dist;
		if (kHoist)
			dist = dist_h;
		else
			dist = R.dmap[((uint32_t) uiz * (uint32_t) A.mh + (uint32_t) uiy) * (uint32_t) A.mw + (uint32_t) uix];
		// Both outcomes of the probe are computed and selected (no nested branch: the two groups of lanes would serialise):
		// dist > 0 skips forward (frag:236-247), dist == 0 marks the cell occupied and steps back (frag:248-262).
		// r = clamp(u_i - u, -1, 0) (frag:234); the operand is never NaN, so the median-of-three instruction gives the same
		// value as min(max(x, -1), 0)
		const float rx = __builtin_amdgcn_fmed3f((float) uix - ux, -1.0f, 0.0f);
		const float ry = __builtin_amdgcn_fmed3f((float) uiy - uy, -1.0f, 0.0f);
		const float rz = __builtin_amdgcn_fmed3f((float) uiz - uz, -1.0f, 0.0f);
		float       ax, ay, az;
		if (SKIP == VKV_SKIP_BLOCK)
		{        // frag:239: step(0, s) is 1 for s >= 0 (and for the impossible NaN), 0 for s < 0
			ax = (((R.six < 0.0f) ? 0.0f : 1.0f) + rx) * R.six;
			ay = (((R.siy < 0.0f) ? 0.0f : 1.0f) + ry) * R.siy;
			az = (((R.siz < 0.0f) ? 0.0f : 1.0f) + rz) * R.siz;
		}
		else
		{        // frag:242: step(0, -s) + sign(s) * dist is exactly dist for s > 0 and 1 - dist for s < 0 (s is never 0 or NaN:
			 // it is the reciprocal of a finite number)
			const float fd = (float) dist;
			ax = (((R.six > 0.0f) ? fd : 1.0f - fd) + rx) * R.six;
			ay = (((R.siy > 0.0f) ? fd : 1.0f - fd) + ry) * R.siy;
			az = (((R.siz > 0.0f) ? fd : 1.0f - fd) + rz) * R.siz;
		}
		if (ax != ax) ax = INFINITY;
		if (ay != ay) ay = INFINITY;
		if (az != az) az = INFINITY;
		float m = g_min(g_min(ax, ay), az);
		m       = (m < 1073741824.0f) ? m : 1073741824.0f;
		const int  i_skip = i + max(1, (int) __builtin_ceilf(m));
		const int  i_back = max(i - (int) __builtin_ceilf(A.sampling_factor), R.i_min);
		const bool empty  = dist > 0u;
		R.i               = empty ? i_skip : i_back;
		R.occupied        = !empty;
		R.ulx = empty ? R.ulx : uix, R.uly = empty ? R.uly : uiy, R.ulz = empty ? R.ulz : uiz;
		return R.i >= R.n_steps;
	}

	// frag:266-310
	++R.n_vol;
	uint32_t texel = 0;
	{
	float intensity, gradient = 1.0f;
	if (kHoist)
	{
		float unused;
		if (GRAD == 1)
			packed_filter<true>(q00, q10, q01, q11, hwx, hwy, hwz, intensity, gradient);
		else
			packed_filter<false>(q00, q10, q01, q11, hwx, hwy, hwz, intensity, unused);
	}
	else if (PACKED)
	{
		float unused;
		if (GRAD == 1)
			sample_packed<true>(A.packed, W, H, D, A.pmx, A.pmy, posx, posy, posz, intensity, gradient);
		else
			sample_packed<false>(A.packed, W, H, D, A.pmx, A.pmy, posx, posy, posz, intensity, unused);
	}
	else
	{
		intensity = sample_linear(A.vol, W, H, D, posx, posy, posz);
		if (GRAD == 1)
			gradient = sample_linear(A.grad, W, H, D, posx, posy, posz);
	}
	if (GRAD == 2)
	{        // frag:92-97
		const float dix = 1.0f / (float) W, diy = 1.0f / (float) H, diz = 1.0f / (float) D;
		float       t1, t2, t3, t4, unused;
		if (PACKED)
		{
			sample_packed<false>(A.packed, W, H, D, A.pmx, A.pmy, posx + dix, posy - diy, posz - diz, t1, unused);
			sample_packed<false>(A.packed, W, H, D, A.pmx, A.pmy, posx - dix, posy - diy, posz + diz, t2, unused);
			sample_packed<false>(A.packed, W, H, D, A.pmx, A.pmy, posx - dix, posy + diy, posz - diz, t3, unused);
			sample_packed<false>(A.packed, W, H, D, A.pmx, A.pmy, posx + dix, posy + diy, posz + diz, t4, unused);
		}
		else
		{
			t1 = sample_linear(A.vol, W, H, D, posx + dix, posy - diy, posz - diz);
			t2 = sample_linear(A.vol, W, H, D, posx - dix, posy - diy, posz + diz);
			t3 = sample_linear(A.vol, W, H, D, posx - dix, posy + diy, posz - diz);
			t4 = sample_linear(A.vol, W, H, D, posx + dix, posy + diy, posz + diz);
		}
		const float gx = (((t1 - t2) - t3) + t4) * 0.25f;
		const float gy = (((-t1 - t2) + t3) + t4) * 0.25f;
		const float gz = (((-t1 + t2) - t3) + t4) * 0.25f;
		const float len = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz);
		gradient = g_clamp(len * A.grad_modifier, 0.0f, 1.0f);
	}
	// get_color (transfer_function.glsl:35-38): NEAREST texel.  With the bit table the occupied test (frag:276) comes
	// from LDS and only occupied samples pay the dependent RGBA fetch.
	const uint32_t tidx  = (uint32_t) tf_texel(gradient) * 256u + (uint32_t) tf_texel(intensity);
	if (tf_bits)
	{
		if ((s_bits[tidx >> 5] >> (tidx & 31u)) & 1u)
			texel = reinterpret_cast<const uint32_t *>(A.tf)[tidx];
	}
	else
		texel = reinterpret_cast<const uint32_t *>(A.tf)[tidx];
	}
	const uint32_t ab = texel >> 24;
	R.occupied        = ab > 0;
	bool ended        = false;
	if (R.occupied)
	{
		if (SKIP != VKV_SKIP_NONE)
			R.ulx = uix, R.uly = uiy, R.ulz = uiz;
		const float a  = s_alpha[ab];        // frag:283
		// R8G8B8A8_UNORM -> float is exactly c / 255: read from the LDS table the workgroup filled with that division
		const float r_ = s_unorm[texel & 255u] * a, g_ = s_unorm[(texel >> 8) & 255u] * a, b_ = s_unorm[(texel >> 16) & 255u] * a;
		const float om = 1.0f - R.a;         // frag:287
		R.r = __builtin_fmaf(om, r_, R.r), R.g = __builtin_fmaf(om, g_, R.g), R.b = __builtin_fmaf(om, b_, R.b);
		R.a = __builtin_fmaf(om, a, R.a);
		if (a > 0.0f)
			R.first_hit = i;
		if (ERT && R.a > 0.99f)
		{        // frag:293-299
			R.a   = 1.0f;
			ended = true;
		}
	}
	else
		++R.n_empty;
	if (!ended)
	{
		R.i     = i + 1;
		R.i_min = R.i;
		ended   = R.i >= R.n_steps;
	}
	return ended;
}

// frag:315-334 + the stores.  `marched` is false for pixels that never entered the loop.
// 16-byte non-temporal store of one RGBA32F pixel (an ext_vector so that it stays one global_store_dwordx4 nt)
__device__ __forceinline__ void store_float4_nt(float *base, size_t pixel, float r, float g, float b, float a)
{
	typedef float float4v __attribute__((ext_vector_type(4)));
	const float4v v = {r, g, b, a};
	__builtin_nontemporal_store(v, reinterpret_cast<float4v *>(base) + pixel);
}

__device__ __forceinline__ void ray_finish(const RayMarchArgs &A, Ray &R, bool marched)
{
	if (marched)
	{
		if (A.out_depth && R.a > 0.0f && R.first_hit < R.n_steps)
		{        // frag:315-321
			const float fi   = (float) R.first_hit;
			const float p[4] = {__builtin_fmaf(fi, R.sx, R.ex) - 0.5f, __builtin_fmaf(fi, R.sy, R.ey) - 0.5f, __builtin_fmaf(fi, R.sz, R.ez) - 0.5f, 1.0f};
			float       a4[4], b4[4], c4[4];
			mat4_mul_vec4(A.model, p, a4);
			mat4_mul_vec4(A.view, a4, b4);
			mat4_mul_vec4(A.proj, b4, c4);
			R.depth = c4[2] / c4[3];
		}
		if (A.test == VKV_TEST_NUM_TEXTURE_SAMPLES)
		{        // frag:324-334
			const int      dim_max     = max(max(A.W, A.H), A.D);
			const uint32_t n_steps_max = (uint32_t) (__builtin_ceilf((float) dim_max * __builtin_sqrtf(3.0f)) * A.sampling_factor);
			const float    v           = (float) (R.n_vol + R.n_dist) / (float) n_steps_max;
			R.r = R.g = R.b = v;
			R.a             = 1.0f;
		}
	}
	const size_t o = R.o;
	if (!R.fragment)
	{        // no fragment: an existing target stays as it is, a fresh one holds the clear values
		if (A.blend)
		{
			if (A.out_counts)
				A.out_counts[o * 3 + 0] = A.out_counts[o * 3 + 1] = A.out_counts[o * 3 + 2] = 0;
			return;
		}
		if (A.depth_attachment)
			R.depth = A.in_depth[o];
		// a fresh target gets the clear values (out_color = 0, counters 0: nothing was added to them since ray_setup / ray_clear)
		if (A.out_color)
			store_float4_nt(A.out_color, o, 0.0f, 0.0f, 0.0f, 0.0f);
		if (A.out_rgba8)
			__builtin_nontemporal_store(0u, reinterpret_cast<uint32_t *>(A.out_rgba8) + o);
		if (A.out_counts)
			A.out_counts[o * 3 + 0] = A.out_counts[o * 3 + 1] = A.out_counts[o * 3 + 2] = 0;
		if (A.out_depth)
			__builtin_nontemporal_store(R.depth, A.out_depth + o);
		return;
	}
	else if (A.blend)
	{        // blend state of the subpass (src/volume_render_subpass.cpp:176-190): rgb = src + (1 - src.a) * dst, a = src.a * (1 - src.a)
		const float om = 1.0f - R.a;
		if (A.out_color)
		{
			float4 d = reinterpret_cast<float4 *>(A.out_color)[o];
			d.x = __builtin_fmaf(om, d.x, R.r), d.y = __builtin_fmaf(om, d.y, R.g), d.z = __builtin_fmaf(om, d.z, R.b), d.w = R.a * om;
			reinterpret_cast<float4 *>(A.out_color)[o] = d;
		}
		if (A.out_rgba8)
		{
			const uint32_t d = reinterpret_cast<uint32_t *>(A.out_rgba8)[o];
			const float    r = __builtin_fmaf(om, unorm8(d & 255u), R.r), g = __builtin_fmaf(om, unorm8((d >> 8) & 255u), R.g),
			            b = __builtin_fmaf(om, unorm8((d >> 16) & 255u), R.b);
			reinterpret_cast<uint32_t *>(A.out_rgba8)[o] = (uint32_t) quantise_rgba8(r) | ((uint32_t) quantise_rgba8(g) << 8) |
			                                               ((uint32_t) quantise_rgba8(b) << 16) | ((uint32_t) quantise_rgba8(R.a * om) << 24);
		}
		if (A.out_counts)
		{
			A.out_counts[o * 3 + 0] = R.n_vol;
			A.out_counts[o * 3 + 1] = R.n_dist;
			A.out_counts[o * 3 + 2] = R.n_empty;
		}
		if (A.out_depth)
			A.out_depth[o] = R.depth;
		return;
	}
	if (A.out_color)
		store_float4_nt(A.out_color, o, R.r, R.g, R.b, R.a);
	// (non-temporal, like the float colour above and the depth below: the frame is not read again by this kernel, and its 8 MB per frame would otherwise push volume bricks out of the L2s:
	// 0.1157 -> 0.1139 ms per frame on C3)
	if (A.out_rgba8)
		__builtin_nontemporal_store((uint32_t) quantise_rgba8(R.r) | ((uint32_t) quantise_rgba8(R.g) << 8) |
		                                               ((uint32_t) quantise_rgba8(R.b) << 16) | ((uint32_t) quantise_rgba8(R.a) << 24), reinterpret_cast<uint32_t *>(A.out_rgba8) + o);
	if (A.out_counts)
	{
		A.out_counts[o * 3 + 0] = R.n_vol;
		A.out_counts[o * 3 + 1] = R.n_dist;
		A.out_counts[o * 3 + 2] = R.n_empty;
	}
	if (A.out_depth)
		__builtin_nontemporal_store(R.depth, A.out_depth + o);
}

// 8x8 work unit `u` (4 per 16x16 block of the tile schedule) + ray slot in the unit -> pixel and output index.
__device__ __forceinline__ bool unit_pixel(const RayMarchArgs &A, uint32_t u, uint32_t slot, uint32_t &px, uint32_t &py, uint32_t &o)
{
	const uint32_t b = u >> 2, w = u & 3u;
	const uint32_t k = b / A.blocks_per_tile, sb = b % A.blocks_per_tile;
	const uint32_t t = A.tile_first + k * A.tile_stride;
	const uint32_t lx = (sb % A.blocks_per_tile_x) * 16 + (w & 1) * 8 + (slot & 7);
	const uint32_t ly = (sb / A.blocks_per_tile_x) * 16 + (w >> 1) * 8 + (slot >> 3);
	px = A.org_x + (t % A.tiles_x) * A.tile_w + lx, py = A.org_y + (t / A.tiles_x) * A.tile_h + ly;
	o  = A.compact ? (k * A.tile_h + ly) * A.tile_w + lx : py * A.img_w + px;
	return px < A.img_w && py < A.img_h;
}

__device__ __forceinline__ void stage_tables(const RayMarchArgs &A, float *s_alpha, float *s_unorm, uint32_t *s_bits)
{
	for (int i = threadIdx.x; i < 256; i += blockDim.x)
	{
		s_alpha[i] = A.alpha_lut[i];
		s_unorm[i] = unorm8(i);        // exact IEEE division, once per workgroup
	}
	if (A.tf_bits)
		for (int i = threadIdx.x; i < 2048; i += blockDim.x)
			s_bits[i] = A.tf_bits[i];
	__syncthreads();
}
// ---------------------------------------------------------------------------------------------------------------
// Persistent scheduler.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kInvalidUnit  = 0xffffffffu;
constexpr uint32_t kRefillLanes  = 16;        // re-fill a wave once this many lanes are idle

// Queue q owns the schedule's tiles k = q, q + 8, ... (same tile -> XCD mapping as the static scheduler); its v-th
// work unit is 8x8 sub-tile v % upt of its (v / upt)-th tile, upt = 4 * blocks_per_tile units per tile.
__device__ __forceinline__ uint32_t queue_units(const RayMarchArgs &A, uint32_t q)
{
	const uint32_t tiles = A.tile_count > q ? (A.tile_count - q + 7u) >> 3 : 0u;
	return tiles * A.blocks_per_tile * 4u;
}

// Pop one unit for this wave (wave-uniform result).  Starts at the wave's own queue and steals from the others once
// it is empty.  `q` is updated to the queue that delivered.
__device__ __forceinline__ uint32_t pop_unit(const RayMarchArgs &A, uint32_t &q)
{
	const uint32_t upt = A.blocks_per_tile * 4u;
	for (uint32_t tries = 0; tries < 8; ++tries)
	{
		uint32_t v = 0;
		if ((threadIdx.x & 63) == 0)
			v = atomicAdd(&A.queue_heads[q], 1u);
		v = __builtin_amdgcn_readfirstlane(v);
		if (v < queue_units(A, q))
			return ((v / upt) * 8u + q) * upt + v % upt;
		q = (q + 1) & 7u;
	}
	return kInvalidUnit;
}

template <int SKIP, bool ERT, int GRAD, bool PACKED>
__global__ void __launch_bounds__(256) k_raymarch_persistent(const RayMarchArgs A)
{
	__shared__ float    s_alpha[256], s_unorm[256];
	__shared__ uint32_t s_bits[2048];
	stage_tables(A, s_alpha, s_unorm, s_bits);
	const bool     tf_bits = A.tf_bits != nullptr;
	// blocks b and b + 8 share an XCD under the observed round-robin placement (speed only, never correctness)
	uint32_t q      = blockIdx.x & 7u;
	uint32_t unit   = pop_unit(A, q);
	uint32_t cursor = 0;        // next unassigned ray slot of `unit`
	bool     active = false;
	Ray      R;
	R.o = 0;

	for (;;)
	{
		uint64_t idle   = __ballot(!active);
		uint32_t n_idle = (uint32_t) __popcll(idle);
		// ---- re-fill: idle lanes take the next ray slots of the current unit, in lane order ----
		while (n_idle >= kRefillLanes && unit != kInvalidUnit)
		{
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) idle, 0u));
			const uint32_t take = min(n_idle, 64u - cursor);
			if (!active && rank < take)
			{
				uint32_t px, py, o;
				if (unit_pixel(A, unit, cursor + rank, px, py, o))
				{
					R.o    = o;
					active = ray_setup<SKIP>(A, px, py, R);
					if (!active)
						ray_finish(A, R, false);        // not covered / early-out / entry-exit test: result is final
				}
			}
			cursor += take;
			if (cursor == 64)
			{
				unit   = pop_unit(A, q);
				cursor = 0;
			}
			idle   = __ballot(!active);
			n_idle = (uint32_t) __popcll(idle);
		}
		if (n_idle == 64)
			break;        // nothing in flight and the queues are empty
		// ---- one event per active lane ----
		if (active)
		{
			if (ray_event<SKIP, ERT, GRAD, PACKED>(A, R, s_alpha, s_unorm, s_bits, tf_bits))
			{
				ray_finish(A, R, true);
				active = false;
			}
		}
	}
}


// ===============================================================================================================
// "Evaluate + replay" scheduler (k_raymarch_er)
// ===============================================================================================================
// LDS tables of a workgroup.  sep = the transfer function is the separable greyscale product (kTfFlagSeparable): the texel's
// alpha byte comes from ai[] x ag[], its colour contribution from pair[]; otherwise the alpha > 0 bit table gates the RGBA fetch.
struct RmLds
{
	union
	{
		struct
		{
			float    alpha[256];        // opacity correction keyed by the alpha byte (frag:283)
			uint32_t bits[2048];
			float    unorm[256];        // byte / 255 (exact IEEE division)
		} g;
		struct
		{
			float  ai[258], ag[258];        // entry 256 repeats entry 255: int(u * 256) of u = 1.0 needs no clamp (kLeanTf)
			float2 pair[256];               // {alpha_lut[b], (b / 255) * alpha_lut[b]}: corrected opacity and premultiplied grey of alpha byte b
		} s;        // 4 112 bytes: the rest of the block (and the dynamic segment behind it) holds the full address tables (kLeanFull)
	};
};

__device__ __forceinline__ bool tf_is_separable(const RayMarchArgs &A)
{
	return A.tf_bits != nullptr && (A.tf_bits[kTfFlagWord] & kTfFlagSeparable) != 0u;        // wave-uniform (scalar load)
}

__device__ __forceinline__ bool stage_tables_er(const RayMarchArgs &A, RmLds &L)
{
	const bool sep = tf_is_separable(A);
	for (int i = threadIdx.x; i < 256; i += blockDim.x)
	{
		const float a = A.alpha_lut[i];
		if (!sep)
			L.g.alpha[i] = a;
		if (sep)
		{
			L.s.ai[i]   = __uint_as_float(A.tf_bits[kTfAiWord + i]);
			L.s.ag[i]   = __uint_as_float(A.tf_bits[kTfAgWord + i]);
			if (i == 255)
				L.s.ai[256] = L.s.ai[255], L.s.ag[256] = L.s.ag[255];
			L.s.pair[i] = i == 0 ? make_float2(0.0f, 0.0f) : make_float2(a, unorm8(i) * a);        // (alpha byte 0 blends nothing: lean_march relies on it)
		}
		else
			L.g.unorm[i] = unorm8(i);
	}
	if (!sep && A.tf_bits)
		for (int i = threadIdx.x; i < 2048; i += blockDim.x)
			L.g.bits[i] = A.tf_bits[i];
	__syncthreads();
	return sep;
}
// ray `rb` (0..255) of 16x16 block `b` of the tile schedule -> pixel and output index.  A wave's 64/W rays form a compact patch.
template <int W>
__device__ __forceinline__ bool block_pixel(const RayMarchArgs &A, uint32_t b, uint32_t rb, uint32_t &px, uint32_t &py, uint32_t &o)
{
	// A wave's 64 pixels (W = 1) are 8x8, 4 wide x 16 tall or 16 wide x 4 tall: the launcher picks the shape whose footprint in VOXELS is the most
	// compact for this view (RayMarchArgs::wave_pw: with anisotropic voxels a pixel step in x can cover twice the voxels of a step in y; the rays of
	// a wave then share more bricks and stay in step longer in the narrow shape: C3, whose z voxels are 2.3 x its x voxels, 0.1054 -> 0.1040 ms per
	// frame and 0.212 -> 0.206 for one frame alone with 4x16; the isotropic cube is best at 8x8, and 16x4 loses 9 % on C3)
	constexpr uint32_t G = 64 / W;
	uint32_t           bx, by;
	const uint32_t     wb = rb / G, g = rb % G;        // wave of the block, ray of the wave
	if (W == 1)
	{        // (shifts and masks: the shape is a run-time power of two)
		const uint32_t pwl = A.wave_pw_log2, pprl = 4u - pwl;        // patch 2^pwl wide, 64 >> pwl tall, 16 >> pwl patches per row of the block
		bx = ((wb & ((1u << pprl) - 1u)) << pwl) + (g & ((1u << pwl) - 1u));
		by = ((wb >> pprl) << (6u - pwl)) + (g >> pwl);
	}
	else
	{
		constexpr uint32_t pw = G >= 32 ? 8 : (G >= 8 ? 4 : 2), ph = G / pw, ppr = 16 / pw;
		bx = (wb % ppr) * pw + g % pw, by = (wb / ppr) * ph + g / pw;
	}
	const uint32_t k = b / A.blocks_per_tile, sb = b % A.blocks_per_tile;
	const uint32_t t  = A.tile_first + k * A.tile_stride;
	const uint32_t lx = (sb % A.blocks_per_tile_x) * 16 + bx, ly = (sb / A.blocks_per_tile_x) * 16 + by;
	px = A.org_x + (t % A.tiles_x) * A.tile_w + lx, py = A.org_y + (t / A.tiles_x) * A.tile_h + ly;
	o  = A.compact ? (k * A.tile_h + ly) * A.tile_w + lx : py * A.img_w + px;
	return px < A.img_w && py < A.img_h;
}
// ===============================================================================================================
// k_raymarch_lean: one lane per ray, the frag's loop body as straight-line predicated code (lean_march).  What it does, each step
// measured against its predecessor on the GPU (profiles/HISTORY.md; the alternatives that lost live in tools/lab/raymarch_lab.hpp):
//   * the transfer function of the reference (a separable greyscale product, checked on the device) comes from two 257-entry LDS
//     tables indexed straight from the filtered values, and one colour channel is blended: no dependent global texel fetch;
//   * the probe byte (probing lanes) and the four footprint dwords (sampling lanes) are issued under EXEC AHEAD of both outcome blocks
//     (a wave barrier keeps the compiler from sinking them); lanes that cannot need a load sit it out;
//   * loads (probe byte | footprint) and outcomes (probe | sample) are ONE if / else each, probe side first; the loop ends on its position
//     (early termination moves it to the end: no done flag), the blend of a sample sits behind one wave-uniform branch (round 5: what costs
//     in this loop is the number of instructions, EXEC writes and mask bookkeeping above all; 61 VGPRs = 8 waves per SIMD);
//   * the footprint address is a sum of per-axis terms read from tables in LDS (kLeanLut / kLeanFull) instead of ~20 half-rate
//     shift / mask / multiply instructions; clamp bounds come from scalar registers, the cell index from 24-bit multiply-adds;
//   * the loop position, its bounds and the first hit are floats (exact: n_steps <= 2^24): no conversion at the head of an iteration;
//   * byte -> float through v_cvt_f32_ubyteN with the differences taken in float (exact), `fma` instead of a select in the skip length.
// ===============================================================================================================
__device__ __forceinline__ int med3_i32(int x, int lo, int hi)
{        // clamp(x, lo, hi) for lo <= hi in one instruction (the compiler only forms it when it can prove lo <= hi)
	int r;
	asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
	return r;
}

template <typename T>
__device__ __forceinline__ T undefined_value()
{        // whatever the register holds (for values that are never read in the lanes that get them): an empty asm "defines" it, so
	 // the compiler neither zeroes it nor reasons about an undefined value
	T x;
	asm volatile("" : "=v"(x));
	return x;
}

// a wave-uniform 64-bit value moved into scalar registers
__device__ __forceinline__ uint64_t uniform_u64(uint64_t x)
{
	const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) x), hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (x >> 32));
	return ((uint64_t) hi << 32) | lo;
}

// A wave-uniform float the march loop reads in every iteration, kept in a SCALAR register: v_readfirstlane moves it there and the empty asm
// stops the compiler from re-deriving it in the loop (in the batch kernel the arguments sit in memory: it would re-load and re-convert
// them).  A VALU instruction reads one scalar operand for free, and the ten or so uniform operands of the loop then stop counting
// against the 64 vector registers that 8 waves per SIMD allow.
__device__ __forceinline__ float uniform_f32(float x)
{        // the empty asm hides from the compiler that x is uniform already (it would fold the readfirstlane away and leave the value in
	 // its VGPR); the move itself is the builtin, so that the compiler's own hazard and wait-count tracking covers it (a v_readfirstlane
	 // written as inline asm rendered BLOCK-mode frames wrong, differently from run to run)
	asm volatile("" : "+v"(x));
	return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

__device__ __forceinline__ float max_f32_raw(float a, float b)
{        // v_max_f32 without the canonicalisation fmaxf puts in front of it (both operands are ordinary numbers here)
	float r;
	asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

__device__ __forceinline__ int clamp0_i32(int x, int hi)
{        // clamp(x, 0, hi) with a wave-uniform hi kept in a scalar register (no per-iteration v_mov of the bound)
	int r;
	asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
	return r;
}

// the clamp of lean_march's rare iterations (a ray's first and last loop positions): volatile, so that the compiler keeps it behind its
// wave-uniform branch instead of executing it always and selecting
__device__ __forceinline__ float clamp0_f32_cold(float x, float hi)
{        // clamp(x, 0, hi), hi wave-uniform (a scalar operand)
	float r;
	hi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hi)));
	asm volatile("v_med3_f32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
	return r;
}

__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c)
{        // a * b + c for a, b < 2^24 and a wave-uniform b: one half-rate instruction (the compiler picks v_mad_u64_u32 + v_mul_u32_u24 + add)
	uint32_t r;
	asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
	return r;
}

// Template flags of lean_march (LF).  raymarch_inst.hpp lists the combinations the product instantiates.
constexpr uint32_t kLeanLut      = 1u;        // footprint address = X[bx] + Y[by] + Z[bz] from two-level per-axis tables in LDS; clamp bounds of the cell
                                              // coordinates from scalar registers and 24-bit multiply-adds for the cell index (the launcher checks
                                              // map_fits_u24: z * mh + y and mw below 2^24 - every map with block >= 2 that fits the device)
constexpr uint32_t kLeanFull     = 2u;        // with kLeanLut and the separable transfer function: one table entry per voxel index and axis
constexpr uint32_t kLeanNoCounts = 4u;        // the three per-pixel counters (volume samples, map probes, empty samples) are not kept: for launches without
                                              // d_out_counts and outside Test::NumTextureSamples - the reference keeps them only in its test modes - two
                                              // additions and the EXEC-masked else-branch of the empty sample leave the loop
constexpr uint32_t kLeanSafe     = 16u;       // with kLeanFull: waves whose rays provably never meet a clamp between their first and last loop position march
                                              // without the nine clamps of an iteration (lean_march: "clamp-free march loop")
constexpr uint32_t kLeanAsync    = 32u;       // separable transfer function + empty-space skipping: the loop's global loads are issued through inline asm and waited
                                              // for with hand-set counts, so that the probe outcome runs while the footprint gathers are still in flight
constexpr uint32_t kLeanStamp    = 8u;        // (diagnostic, instantiated by tools/lab only, with the trace buffer) s_memtime at the top of every iteration,
                                              // summed per wave by the iteration's kind: only probing lanes / only sampling lanes / both
constexpr size_t   kMaxLutBytes  = 8 * 1024;  // LDS budget of the two-level address tables (1.2 KB at 1024 voxels per axis, 1.9 KB at 2048)

__device__ __forceinline__ float cvt_ubyte0(uint32_t q) { float f; asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(q)); return f; }
__device__ __forceinline__ float cvt_ubyte1(uint32_t q) { float f; asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(q)); return f; }
__device__ __forceinline__ float cvt_ubyte2(uint32_t q) { float f; asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(q)); return f; }
__device__ __forceinline__ float cvt_ubyte3(uint32_t q) { float f; asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(q)); return f; }

// packed_filter with every byte converted by v_cvt_f32_ubyteN and the differences taken in float (exact either way: both operands are
// integers below 256): the compiler's integer-subtract + v_cvt_f32_i32 form costs two half-rate instructions per difference, this one
// a half-rate conversion and a full-rate subtraction
// TABLE: the results come out as value * 1024 (byte offset scale of the 256-entry float tables) - ONE multiplication by kInv255 * 1024,
// which is the same float as (x * kInv255) * 1024: scaling by a power of two is exact and nothing here is near the subnormal range
template <bool WANT_G, bool TABLE = false>
__device__ __forceinline__ void packed_filter_cvt(uint32_t q00, uint32_t q10, uint32_t q01, uint32_t q11, float wx, float wy, float wz, float &out_v, float &out_g)
{
	constexpr float kScale = TABLE ? kInv255 * 1024.0f : kInv255;
	{
		const float b000 = cvt_ubyte0(q00), b100 = cvt_ubyte2(q00), b010 = cvt_ubyte0(q10), b110 = cvt_ubyte2(q10);
		const float b001 = cvt_ubyte0(q01), b101 = cvt_ubyte2(q01), b011 = cvt_ubyte0(q11), b111 = cvt_ubyte2(q11);
		const float c00 = __builtin_fmaf(wx, b100 - b000, b000), c10 = __builtin_fmaf(wx, b110 - b010, b010);
		const float c01 = __builtin_fmaf(wx, b101 - b001, b001), c11 = __builtin_fmaf(wx, b111 - b011, b011);
		const float c0 = __builtin_fmaf(wy, c10 - c00, c00), c1 = __builtin_fmaf(wy, c11 - c01, c01);
		out_v = __builtin_fmaf(wz, c1 - c0, c0) * kScale;
	}
	if (WANT_G)
	{
		const float b000 = cvt_ubyte1(q00), b100 = cvt_ubyte3(q00), b010 = cvt_ubyte1(q10), b110 = cvt_ubyte3(q10);
		const float b001 = cvt_ubyte1(q01), b101 = cvt_ubyte3(q01), b011 = cvt_ubyte1(q11), b111 = cvt_ubyte3(q11);
		const float c00 = __builtin_fmaf(wx, b100 - b000, b000), c10 = __builtin_fmaf(wx, b110 - b010, b010);
		const float c01 = __builtin_fmaf(wx, b101 - b001, b001), c11 = __builtin_fmaf(wx, b111 - b011, b011);
		const float c0 = __builtin_fmaf(wy, c10 - c00, c00), c1 = __builtin_fmaf(wy, c11 - c01, c01);
		out_g = __builtin_fmaf(wz, c1 - c0, c0) * kScale;
	}
}
// The LDS of the lean kernels is ONE dynamic segment whose layout the launcher sizes (lean_lds_bytes): RmLds at its start, behind it
// either the two-level address tables (kLeanLut) or, starting inside RmLds behind the separable transfer-function tables, the
// per-voxel-index tables (kLeanFull).  The kernels have NO static LDS object, so the segment starts at LDS address 0 - RmLds sits at
// kLdsBase = 16 (address 0 would be a null pointer to the compiler) - and every table address is a compile-time constant that folds into the offset field of its ds_read (with a relocatable base the march loop carried five
// more VALU additions per iteration: 443 -> 455 priced cycles, 0.1143 -> 0.1155 ms per C3 frame).  lean_lds_check() stops a kernel whose
// segment does not start at 0 (a compiler or an inlined helper that brings a static __shared__ object - __syncthreads_or does: its
// workgroup reduction owns 256 static bytes, hence wg_any below), instead of letting it read the wrong tables.
extern __shared__ __align__(16) uint32_t s_lean_lds[];
static_assert(sizeof(RmLds) % 16 == 0, "the address tables behind RmLds are read as 64-bit words");

typedef __attribute__((address_space(3))) RmLds *   lds_rmlds_ptr;
typedef __attribute__((address_space(3))) uint32_t *lds_u32_ptr;

constexpr uintptr_t kLdsBase = 16;
__device__ __forceinline__ RmLds &lean_lds() { return *(RmLds *) (lds_rmlds_ptr) kLdsBase; }

__device__ __forceinline__ void lean_lds_check()
{        // (through an empty asm: the compiler must not reason about the comparison)
	uint32_t base = (uint32_t) reinterpret_cast<uintptr_t>(s_lean_lds);
	asm volatile("" : "+s"(base));
	if (base != 0u)
		__builtin_trap();
}

// workgroup-wide "any": one flag word per wave at the start of the segment (before the tables are staged there), two barriers
__device__ __forceinline__ bool wg_any(bool pred)
{
	uint32_t *flags = (uint32_t *) (lds_u32_ptr) kLdsBase;
	const bool mine = __builtin_amdgcn_ballot_w64(pred) != 0ull;
	if ((threadIdx.x & 63u) == 0u)
		flags[threadIdx.x >> 6] = mine ? 1u : 0u;
	__syncthreads();
	const uint32_t nw  = blockDim.x >> 6;        // 4 in the product; the lab builds workgroups of 1 and 2 waves
	const uint4    f   = *reinterpret_cast<const uint4 *>(flags);
	const bool     any = (f.x | (nw > 1u ? f.y : 0u) | (nw > 2u ? f.z : 0u) | (nw > 3u ? f.w : 0u)) != 0u;
	__syncthreads();
	return any;
}

// LDS copy of the two-level per-axis address tables (behind RmLds; ~1.5 KB):
// words [0, 32) x position inside a macro-brick, [32, 64) y, [64, 96) z; then the macro-brick terms: x at word 96, y at A.lut_y,
// z (64-bit, already an address inside the packed image) at A.lut_z.
#define s_addr_lut ((uint32_t *) (lds_u32_ptr) (kLdsBase + sizeof(RmLds)))
constexpr uint32_t kLutXm = 96;

__device__ __forceinline__ void stage_addr_lut(const RayMarchArgs &A)
{
	for (uint32_t i = threadIdx.x; i < A.lut_z; i += blockDim.x)
		s_addr_lut[i] = A.addr_lut[i];
	const uint64_t base = reinterpret_cast<uint64_t>(A.packed);
	for (uint32_t i = threadIdx.x; 2 * i + A.lut_z < A.lut_words; i += blockDim.x)
	{
		const uint64_t z = (((uint64_t) A.addr_lut[A.lut_z + 2 * i + 1]) << 32 | A.addr_lut[A.lut_z + 2 * i]) + base;
		s_addr_lut[A.lut_z + 2 * i] = (uint32_t) z, s_addr_lut[A.lut_z + 2 * i + 1] = (uint32_t) (z >> 32);
	}
}

// Address of the footprint's first dword from the tables.  bx = clamp(ix, -1, W) + 1 as in packed_footprint, but computed as 4 * bx
// straight from the float (fma(fx, 4, 4) is exact: fx is a small integer): 4 * bx & 124 is the byte offset into the in-macro table,
// (4 * bx >> 5) & ~3 the byte offset into the macro table.  Both terms of an axis and the three axes add up to the offset
// packed_footprint computes with ~20 half-rate shift / mask / multiply instructions; the LDS pipe is otherwise idle in this loop.
template <bool SC = true>
__device__ __forceinline__ const uint8_t *packed_footprint_lut(const RayMarchArgs &A, float px, float py, float pz, float &wx, float &wy, float &wz)
{
	const float cx = __builtin_fmaf(px, (float) A.W, -0.5f), cy = __builtin_fmaf(py, (float) A.H, -0.5f), cz = __builtin_fmaf(pz, (float) A.D, -0.5f);
	const float fx = __builtin_floorf(cx), fy = __builtin_floorf(cy), fz = __builtin_floorf(cz);
	wx = cx - fx, wy = cy - fy, wz = cz - fz;
	const uint32_t tx = (uint32_t) (SC ? clamp0_i32((int) __builtin_fmaf(fx, 4.0f, 4.0f), 4 * (A.W + 1)) : med3_i32((int) __builtin_fmaf(fx, 4.0f, 4.0f), 0, 4 * (A.W + 1)));
	const uint32_t ty = (uint32_t) (SC ? clamp0_i32((int) __builtin_fmaf(fy, 4.0f, 4.0f), 4 * (A.H + 1)) : med3_i32((int) __builtin_fmaf(fy, 4.0f, 4.0f), 0, 4 * (A.H + 1)));
	const uint32_t tz = (uint32_t) (SC ? clamp0_i32((int) __builtin_fmaf(fz, 4.0f, 4.0f), 4 * (A.D + 1)) : med3_i32((int) __builtin_fmaf(fz, 4.0f, 4.0f), 0, 4 * (A.D + 1)));
	const char *   lut = reinterpret_cast<const char *>(s_addr_lut);
	const uint32_t xi = *reinterpret_cast<const uint32_t *>(lut + (tx & 124u)), xm = *reinterpret_cast<const uint32_t *>(lut + 4u * kLutXm + ((tx >> 5) & ~3u));
	const uint32_t yi = *reinterpret_cast<const uint32_t *>(lut + 128u + (ty & 124u)), ym = *reinterpret_cast<const uint32_t *>(lut + 4u * A.lut_y + ((ty >> 5) & ~3u));
	const uint32_t zi = *reinterpret_cast<const uint32_t *>(lut + 256u + (tz & 124u));
	const uint64_t zm = *reinterpret_cast<const uint64_t *>(lut + 4u * A.lut_z + ((tz >> 4) & ~7u));
	return reinterpret_cast<const uint8_t *>(zm + (((xi + xm) + (yi + ym)) + zi));
}

// kLeanFull — one table entry per padded voxel index and axis, so an axis costs one LDS read and no shift / mask / add: X[W + 2],
// Y[H + 2], Z[D + 2], 32-bit offsets in units of TWO bytes (every term is even; a packed image of up to 8 GiB).  Built by the workgroup
// from the two-level tables (each entry = in-macro term + macro term).  11.4 KB at 1024 x 1024 x 795: too much on top of the general
// transfer-function tables, so they are only used with the separable transfer function, whose tables end 4 112 bytes into RmLds: the
// full tables start there and run on past the end of RmLds (the launcher sizes the segment: lean_lds_bytes).  Small on purpose: a workgroup
// keeps its LDS until its longest wave is done, so the LDS per workgroup decides how many waves a CU holds on average.
constexpr uint32_t kFullLutWord  = 1028;         // = sizeof(RmLds::s) / 4, checked below
__host__ __device__ __forceinline__ bool map_fits_u24(uint32_t mw, uint32_t mh, uint32_t md) { return (uint64_t) mh * md < (1ull << 24) && mw < (1u << 24); }
constexpr size_t   kFullLdsLimit = 17920;        // LDS of a workgroup that still lets 9 workgroups share a CU

__host__ __device__ __forceinline__ size_t full_lut_bytes(int W, int H, int D) { return (size_t) (W + 2 + H + 2 + D + 2) * 4; }

// dynamic LDS of a lean kernel: kind 0 = RmLds alone, 1 = + two-level tables of lut_words, 2 = whichever of that and the per-voxel-index
// tables (which start kFullLutWord words into RmLds) ends later (a kLeanFull kernel falls back to the two-level tables when the
// transfer function is not separable)
__host__ __forceinline__ size_t lean_lds_bytes(int kind, uint32_t lut_words, int W, int H, int D)
{
	const size_t two_level = kLdsBase + sizeof(RmLds) + (size_t) lut_words * 4;
	if (kind == 0)
		return kLdsBase + sizeof(RmLds);
	if (kind == 1)
		return two_level;
	const size_t full_end = kLdsBase + (size_t) kFullLutWord * 4 + full_lut_bytes(W, H, D);
	return two_level > full_end ? two_level : full_end;
}

__device__ __forceinline__ uint32_t *full_lut_base(const RmLds &L) { return const_cast<uint32_t *>(reinterpret_cast<const uint32_t *>(&L)) + kFullLutWord; }

__device__ __forceinline__ void stage_full_lut(const RayMarchArgs &A, RmLds &L)
{
	static_assert(sizeof(((RmLds *) nullptr)->s) == kFullLutWord * 4, "full tables start behind the separable TF tables");
	const uint32_t  nx = (uint32_t) A.W + 2u, ny = (uint32_t) A.H + 2u, nz = (uint32_t) A.D + 2u;
	uint32_t *      fx = full_lut_base(L), *fy = fx + nx, *fz = fy + ny;
	const uint32_t *g  = A.addr_lut;
	for (uint32_t b = threadIdx.x; b < nx; b += blockDim.x)
		fx[b] = (g[b & 31u] + g[kLutXm + (b >> 5)]) >> 1;
	for (uint32_t b = threadIdx.x; b < ny; b += blockDim.x)
		fy[b] = (g[32u + (b & 31u)] + g[A.lut_y + (b >> 5)]) >> 1;
	for (uint32_t b = threadIdx.x; b < nz; b += blockDim.x)
		fz[b] = (uint32_t) ((((((uint64_t) g[A.lut_z + 2u * (b >> 5) + 1u]) << 32) | g[A.lut_z + 2u * (b >> 5)]) + g[64u + (b & 31u)]) >> 1);
}

// Loop-invariant operands of packed_footprint_full, worked out once per ray.  They pass through an empty asm so that the compiler keeps
// them in registers: in the batch kernel (arguments in memory) it otherwise re-loads W, H, D and converts them again in every iteration.
struct FullLutConsts
{
	float          w, h, d, oy, oz;
	const uint8_t *base;        // the packed image
};

template <bool SCALAR = false>
__device__ __forceinline__ FullLutConsts full_lut_consts(const RayMarchArgs &A)
{
	FullLutConsts c;
	c.w = (float) A.W, c.h = (float) A.H, c.d = (float) A.D;
	c.base = A.packed;
	c.oy = (float) (4 * (A.W + 2) + 4);
	c.oz = (float) (4 * (A.W + 2 + A.H + 2) + 4);
	if (SCALAR)        // kLeanFmt: its rows take four more vector registers, the uniform operands move to scalar ones
		c.w = uniform_f32(c.w), c.h = uniform_f32(c.h), c.d = uniform_f32(c.d), c.oy = uniform_f32(c.oy), c.oz = uniform_f32(c.oz);
	else
		asm volatile("" : "+v"(c.w), "+v"(c.h), "+v"(c.d), "+v"(c.oy), "+v"(c.oz));
	return c;
}

// LDS byte address of X[clamp(ix, -1, W) + 1] straight from the float: fma(clamped floor, 4, 4 + table offset) is exact (small integers)
// FREE: the caller knows -1 <= floor(c) <= extent on every axis (lean_march's clamp-free loop)
template <bool FREE = false>
__device__ __forceinline__ const uint8_t *packed_footprint_full(const FullLutConsts &C, const RmLds &L, float px, float py, float pz, float &wx, float &wy, float &wz)
{
	const float cx = __builtin_fmaf(px, C.w, -0.5f), cy = __builtin_fmaf(py, C.h, -0.5f), cz = __builtin_fmaf(pz, C.d, -0.5f);
	const float fx = __builtin_floorf(cx), fy = __builtin_floorf(cy), fz = __builtin_floorf(cz);
	wx = cx - fx, wy = cy - fy, wz = cz - fz;
	const int   tx = (int) __builtin_fmaf(FREE ? fx : __builtin_amdgcn_fmed3f(fx, -1.0f, C.w), 4.0f, 4.0f);
	const int   ty = (int) __builtin_fmaf(FREE ? fy : __builtin_amdgcn_fmed3f(fy, -1.0f, C.h), 4.0f, C.oy);
	const int   tz = (int) __builtin_fmaf(FREE ? fz : __builtin_amdgcn_fmed3f(fz, -1.0f, C.d), 4.0f, C.oz);
	const char *   lut = reinterpret_cast<const char *>(full_lut_base(L));
	const uint32_t xo = *reinterpret_cast<const uint32_t *>(lut + tx), yo = *reinterpret_cast<const uint32_t *>(lut + ty);
	const uint32_t zo = *reinterpret_cast<const uint32_t *>(lut + tz);
	return C.base + ((uint64_t) ((xo + yo) + zo) << 1);        // the terms are in units of two bytes
}
// A footprint address that comes out of the LDS address tables is an integer: tell the compiler it points to global memory, or it emits
// flat_load (which also counts on lgkmcnt, so every wait for an LDS read would wait for the footprint as well).
typedef const __attribute__((address_space(1))) u32_align2 *global_row_ptr;

// The distance-map pointer reaches the batch kernel through its argument block in memory, which hides its address space from the
// compiler: a flat load, which counts on lgkmcnt, so every wait for an address-table read would also wait for the probe byte.
__device__ __forceinline__ uint32_t load_u8_global(const uint8_t *base, uint32_t index)
{
	typedef const __attribute__((address_space(1))) uint8_t *global_u8_ptr;
	return ((global_u8_ptr) (uintptr_t) base)[index];
}

template <bool NT = false>
__device__ __forceinline__ uint32_t load_row(const uint8_t *p)
{
	global_row_ptr g = (global_row_ptr) (uintptr_t) p;
	if (NT)
		return __builtin_nontemporal_load(g);
	return *g;
}

// ---- the march loop's loads with hand-set wait counts (kLeanAsync) --------------------------------------------------------------
// The compiler waits for a load with s_waitcnt vmcnt(n), n = the number of YOUNGER loads that may still be outstanding.  The probe
// byte and the four footprint dwords of an iteration are issued under different EXEC masks, each behind a branch that skips it when
// no lane takes part, so at the probe outcome the compiler cannot know whether four younger loads follow the byte or none: it waits
// for everything (vmcnt(0)) - the probe outcome, a third of an iteration's instructions, then starts only when the slowest footprint
// gather has come back.  Loads return in order, so with the loads hidden from the compiler's bookkeeping (inline asm) and one
// wave-uniform test (does any lane sample in this iteration?) the wait can be exact: vmcnt(4) when footprints follow, vmcnt(0) when none
// do.  The waits take the loaded registers as in/out operands: every use of a loaded value is ordered behind its wait.
// (Other loads the compiler issues itself can only make these waits stricter, never weaker: a count includes them.)
__device__ __forceinline__ void load_u8_async(uint32_t &dst, const uint8_t *base_uniform, uint32_t index)
{
	asm volatile("global_load_ubyte %0, %1, %2" : "=v"(dst) : "v"(index), "s"(base_uniform));
}
__device__ __forceinline__ void load_u8_async_lane(uint32_t &dst, const uint8_t *base_lane, uint32_t index)
{
	const uint8_t *p = base_lane + index;
	asm volatile("global_load_ubyte %0, %1, off" : "=v"(dst) : "v"(p));
}
__device__ __forceinline__ void load_rows_async(uint32_t &q00, uint32_t &q10, uint32_t &q01, uint32_t &q11, const uint8_t *p)
{        // (early-clobber outputs: none of them may share a register with the address)
	asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %4, off offset:10\n\tglobal_load_dword %2, %4, off offset:50\n\tglobal_load_dword %3, %4, off offset:60"
	             : "=&v"(q00), "=&v"(q10), "=&v"(q01), "=&v"(q11)
	             : "v"(p));
}
// Wait for the oldest load in flight, inside the EXEC-masked probe side: four younger loads are behind it when some live lane of the
// iteration does NOT probe (EXEC here = the probing lanes, `live` = EXEC at the top of the iteration), none otherwise.  One asm statement
// with its own scalar branch: as two statements on the two sides of a C++ branch the loaded register comes out as two values the compiler
// merges with moves, and a flag worked out in C++ from a ballot costs two vector instructions.
__device__ __forceinline__ void wait_oldest_of_five_or_one(uint32_t &a, unsigned long long live)
{
	asm volatile("s_cmp_lg_u64 exec, %1\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(0)\n\ts_branch 2f\n1:\n\ts_waitcnt vmcnt(4)\n2:" : "+v"(a) : "s"(live) : "scc");
}
__device__ __forceinline__ void wait_rows(uint32_t &q00, uint32_t &q10, uint32_t &q01, uint32_t &q11)
{
	asm volatile("s_waitcnt vmcnt(0)" : "+v"(q00), "+v"(q10), "+v"(q01), "+v"(q11));
}

struct LeanStamp
{
	uint32_t sum[3], cnt[3];        // shader-clock cycles and iterations of this wave by kind: 0 probe lanes only, 1 sample lanes only, 2 both
	uint32_t clamped;               // clamp-free loop: iterations that took the rare branch (some lane outside its safe range)
};

// ---- clamp-free march loop (kLeanSafe) -------------------------------------------------------------------------------------------
// The frag clamps the map cell and the remainder r (frag:221) and - in the sampler - the texel index: nine half-rate instructions
// per iteration that do nothing on all but the first and the last step of a ray.  Loop position i -> pos = fma(i, step, entry) ->
// u = k * pos is a chain of correctly rounded, monotone operations, so every component of pos and of u is monotone in i, and a
// condition "lo <= value < hi" that holds at two loop positions holds at every position between them.
//   safe(i)  := 0 <= u < map extent on every axis.  Then int(u) = floor(u) IS the clamped cell coordinate and
//               r = float(u_i) - u = -(u - floor(u)) = -fract(u) (both exact; the clamp to [-1, 0] cannot act).
//   A ray is "free" when position 1 and its last position n - 1, or the one before it, are safe: it is then safe on [1, lhi], lhi = n - 1
//   or n - 2 (position 0 lies ON the face the ray enters through, position n - 1 on the one it leaves through, give or take a
//   rounding either way), and when pos stays within [-g, 1 + g] at positions 0 and n - 1, g = 0.25 / the longest extent: then
//   c = fma(pos, extent, -0.5) is within [-0.75, extent - 0.25] at EVERY position: the sampler's clamp (floor(c) to [-1, extent]) cannot act.
// A wave whose marching lanes are all free runs the loop without the nine clamps (lean_march<FREE = true>), any other wave the loop
// with them.  At the (at most two) positions of a free ray that are not safe - 0, and n - 1 when lhi = n - 2 - u is first clamped to
// [0, largest float below the extent] (one wave-uniform branch per iteration guards that): int(u) is then the frag's clamped cell.  r at
// those positions differs from the frag's (fract of the clamped u), and that cannot be observed: position 0 is never probed (the
// frag starts with voxel_occupied = true and i_min >= 1 afterwards), and a probe at the LAST position either finds the cell occupied (no
// skip) or skips by max(1, .) >= 1 past the end of the ray - the loop position a ray ends with is not an output.
__device__ __forceinline__ bool lean_safe_at(const RayMarchArgs &A, const Ray &R, float j, float kx, float ky, float kz)
{
	const float px = __builtin_fmaf(j, R.sx, R.ex), py = __builtin_fmaf(j, R.sy, R.ey), pz = __builtin_fmaf(j, R.sz, R.ez);
	const float vx = kx * px, vy = ky * py, vz = kz * pz;
	return __builtin_fminf(__builtin_fminf(vx, vy), vz) >= 0.0f && vx < A.mapf[0] && vy < A.mapf[1] && vz < A.mapf[2];
}

// wave-uniform: every marching lane's ray is free
__device__ __forceinline__ bool lean_free_wave(const RayMarchArgs &A, const Ray &R)
{
	if (A.clamp_always)
		return false;
	const float kx = (float) A.W / A.block_size[0], ky = (float) A.H / A.block_size[1], kz = (float) A.D / A.block_size[2];
	const float g  = 0.25f / (float) max(max(A.W, A.H), A.D);
	const float last = (float) R.n_steps - 1.0f;
	// pos at positions 0 (= the entry) and n - 1
	const float lx = __builtin_fmaf(last, R.sx, R.ex), ly = __builtin_fmaf(last, R.sy, R.ey), lz = __builtin_fmaf(last, R.sz, R.ez);
	const bool  inside = __builtin_fminf(__builtin_fminf(__builtin_fminf(R.ex, R.ey), R.ez), __builtin_fminf(__builtin_fminf(lx, ly), lz)) >= -g &&
	                    __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(R.ex, R.ey), R.ez), __builtin_fmaxf(__builtin_fmaxf(lx, ly), lz)) <= 1.0f + g;
	const bool free_ray = inside && lean_safe_at(A, R, 1.0f, kx, ky, kz) && (lean_safe_at(A, R, last, kx, ky, kz) || lean_safe_at(A, R, last - 1.0f, kx, ky, kz));
	return __builtin_amdgcn_ballot_w64(!free_ray) == 0ull;
}

// The march loop of one ray (frag:215-312).  SEP: the transfer function is the reference's separable greyscale product (alpha byte from
// the two LDS tables, one colour channel); otherwise the alpha > 0 bit table in LDS gates a dependent RGBA texel fetch.
// kHoist (packed image, gradient from the map or unused): footprint loads issued ahead of the outcome blocks; the other variants (linear
// volume, on-the-fly gradient: five trilinear taps) sample inside the sample block.
template <int SKIP, bool ERT, int GRAD, bool PACKED, bool SEP, uint32_t LF, bool FREE = false>
__device__ __forceinline__ void lean_march(const RayMarchArgs &A, Ray &R, const RmLds &L, uint32_t &iter, LeanStamp &stamp)
{
	constexpr bool kStamp = (LF & kLeanStamp) != 0, kCounts = (LF & kLeanNoCounts) == 0;
	constexpr bool kHoist = PACKED && GRAD != 2;
	constexpr bool kLut = (LF & kLeanLut) != 0, kFull = (LF & kLeanFull) != 0 && SEP, kTf = SEP && kHoist;
	constexpr bool kAsync = (LF & kLeanAsync) != 0 && kTf && SKIP != VKV_SKIP_NONE;        // (not the texture path: its texel fetch is a load the compiler issues)
	uint32_t       stamp_prev = 0, stamp_kind = 3;
	const int      W = A.W, H = A.H, D = A.D;
	const float    kx = SKIP != VKV_SKIP_NONE ? (float) W / A.block_size[0] : 0.0f, ky = SKIP != VKV_SKIP_NONE ? (float) H / A.block_size[1] : 0.0f,
	            kz = SKIP != VKV_SKIP_NONE ? (float) D / A.block_size[2] : 0.0f;
	const int   mw1 = A.mw - 1, mh1 = A.mh - 1, md1 = A.md - 1;
	// (the multipliers of the cell index, pinned in scalar registers: the batch kernel's arguments sit in memory and the compiler would
	// otherwise be free to load them again in every iteration)
	// (readfirstlane: an "s" operand must BE in a scalar register - the compiler reports "illegal VGPR to SGPR copy" when its own copy of a
	// uniform value happens to live in a vector register, which depends on optimisation flags)
	uint32_t amw = (uint32_t) __builtin_amdgcn_readfirstlane(A.mw), amh = (uint32_t) __builtin_amdgcn_readfirstlane(A.mh);
	asm volatile("" : "+s"(amw), "+s"(amh));
	float       grey = 0.0f;
	uint32_t    ul   = 0;
	bool        occ  = true;
	const float sgx = R.six > 0.0f ? 1.0f : -1.0f, sgy = R.siy > 0.0f ? 1.0f : -1.0f, sgz = R.siz > 0.0f ? 1.0f : -1.0f;
	const float ofx = R.six > 0.0f ? 0.0f : 1.0f, ofy = R.siy > 0.0f ? 0.0f : 1.0f, ofz = R.siz > 0.0f ? 0.0f : 1.0f;
	FullLutConsts fullc = {};
	if (kFull)
		fullc = full_lut_consts<FREE>(A);        // (the clamp-free loop keeps them in scalar registers: room for the per-ray map pointer of the anisotropic kernels)
	// loop position, its bounds and the first hit as floats (exact: n_steps <= 2^24)
	float       li = (float) R.i, li_min = (float) R.i_min, lfirst = (float) R.first_hit;
	const float ln = (float) R.n_steps, lback = (float) A.back;
	// the distance map of a launch without the anisotropic maps is the same for every ray: a scalar base for the probe's load
	const uint8_t *const dmap = SKIP == VKV_SKIP_ANISOTROPIC_DISTANCE ? R.dmap : reinterpret_cast<const uint8_t *>(uniform_u64(reinterpret_cast<uintptr_t>(A.maps[0])));
	// FREE: the clamp-free form of the loop, for waves of free rays (lean_free_wave above)
	float lhi_now = -1.0f;        // the last safe position of this lane's ray: -1 until the first iteration (i = 0) is done, then lhi
	static_assert(!FREE || ((LF & kLeanSafe) != 0 && kFull && kHoist && SKIP != VKV_SKIP_NONE && !kStamp), "the clamp-free loop exists for kLeanSafe kernels only");
	if (kAsync)        // nothing of the set-up is in flight when the loop starts: the compiler then carries no vmcnt waits for it round the loop
		__builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) (expcnt, lgkmcnt untouched)
	// (a ray is over when its loop position reaches n_steps - early ray termination puts it there: no "done" flag of its own, whose
	// per-lane state the compiler keeps as wave masks at six scalar operations and a second comparison per iteration)
	while (li < ln)
	{
		const float i    = li;
		const float posx = __builtin_fmaf(i, R.sx, R.ex), posy = __builtin_fmaf(i, R.sy, R.ey), posz = __builtin_fmaf(i, R.sz, R.ez);
		int         uix = 0, uiy = 0, uiz = 0;
		float       ux = 0, uy = 0, uz = 0;
		uint32_t    cell = 0;
		if (SKIP != VKV_SKIP_NONE)
		{        // frag:192, 220-221
			ux = kx * posx, uy = ky * posy, uz = kz * posz;
			if (FREE)
			{
				if (__builtin_expect(__builtin_amdgcn_ballot_w64(i > lhi_now) != 0ull, 0))
				{        // the first iteration of every wave, and the iterations in which a ray with lhi = n - 2 stands at its last position
					ux = clamp0_f32_cold(ux, A.mapb[0]), uy = clamp0_f32_cold(uy, A.mapb[1]), uz = clamp0_f32_cold(uz, A.mapb[2]);        // to [0, largest float below the extent]
#ifdef VKV_TRACE_CLAMPED        // diagnostic build (tools/wave_trace_batch.py prints the share of such iterations: 2.8 % on C3); in the product the
					++stamp.clamped;        // count would cost every iteration a per-lane copy of it (a value that leaves a loop whose lanes exit at different times)
#endif
					// lhi, worked out here, where it is rarely needed, instead of being kept in a register across the loop (the empty asm keeps the
					// compiler from re-using - and spilling - what lean_free_wave computed from the same operands)
					float last = ln;
					asm volatile("" : "+v"(last));
					last -= 1.0f;
					lhi_now = lean_safe_at(A, R, last, kx, ky, kz) ? last : last - 1.0f;
				}
				uix = (int) ux, uiy = (int) uy, uiz = (int) uz;
				cell = mad_u24(mad_u24((uint32_t) uiz, amh, (uint32_t) uiy), amw, (uint32_t) uix);
			}
			else if (kLut)
			{
				uix = clamp0_i32((int) ux, mw1), uiy = clamp0_i32((int) uy, mh1), uiz = clamp0_i32((int) uz, md1);
				cell = mad_u24(mad_u24((uint32_t) uiz, amh, (uint32_t) uiy), amw, (uint32_t) uix);
			}
			else
			{
				uix = med3_i32((int) ux, 0, mw1), uiy = med3_i32((int) uy, 0, mh1), uiz = med3_i32((int) uz, 0, md1);
				cell = __umul24(__umul24((uint32_t) uiz, (uint32_t) A.mh) + (uint32_t) uiy, (uint32_t) A.mw) + (uint32_t) uix;
			}
		}
		const bool probe = SKIP != VKV_SKIP_NONE && !occ && cell != ul;        // frag:224
		if (kStamp)
		{
			const uint32_t now = (uint32_t) __builtin_amdgcn_s_memtime();
			const uint32_t dt  = __builtin_amdgcn_readfirstlane(now - stamp_prev);
			// (readfirstlane: wave-level values, not per-lane copies that stop when their lane's ray ends; no indexing by a variable)
			if (stamp_kind == 0u)
				stamp.sum[0] = __builtin_amdgcn_readfirstlane(stamp.sum[0] + dt), stamp.cnt[0] = __builtin_amdgcn_readfirstlane(stamp.cnt[0] + 1u);
			else if (stamp_kind == 1u)
				stamp.sum[1] = __builtin_amdgcn_readfirstlane(stamp.sum[1] + dt), stamp.cnt[1] = __builtin_amdgcn_readfirstlane(stamp.cnt[1] + 1u);
			else if (stamp_kind == 2u)
				stamp.sum[2] = __builtin_amdgcn_readfirstlane(stamp.sum[2] + dt), stamp.cnt[2] = __builtin_amdgcn_readfirstlane(stamp.cnt[2] + 1u);
			stamp_prev = now;
			const bool any_p = __builtin_amdgcn_ballot_w64(probe) != 0ull, any_s = __builtin_amdgcn_ballot_w64(!probe) != 0ull;
			stamp_kind       = __builtin_amdgcn_readfirstlane(any_p ? (any_s ? 2u : 0u) : 1u);
		}

		// ---- loads: probe byte first, then the footprint of the sampling lanes ----------------------------------------
		// every use sits under the predicate of its load: the values of the other lanes are left undefined (no moves)
		uint32_t dist = undefined_value<uint32_t>(), q00 = undefined_value<uint32_t>(), q10 = undefined_value<uint32_t>(), q01 = undefined_value<uint32_t>(),
		         q11 = undefined_value<uint32_t>();
		float    wx = undefined_value<float>(), wy = undefined_value<float>(), wz = undefined_value<float>();
		unsigned long long live = 0ull;
		if (kAsync)
			live = __builtin_amdgcn_read_exec();        // the lanes whose rays are alive in this iteration
		if (SKIP != VKV_SKIP_NONE && probe)
		{
			if (kAsync && SKIP == VKV_SKIP_ANISOTROPIC_DISTANCE)
				load_u8_async_lane(dist, dmap, cell);
			else if (kAsync)
				load_u8_async(dist, dmap, cell);
			else
				dist = load_u8_global(dmap, cell);
		}
		else if (kHoist)
		{
			const uint8_t *ba = kFull  ? packed_footprint_full<FREE>(fullc, L, posx, posy, posz, wx, wy, wz)
			                    : kLut ? packed_footprint_lut(A, posx, posy, posz, wx, wy, wz)
			                           : packed_footprint(A.packed, W, H, D, A.pmx, A.pmy, posx, posy, posz, wx, wy, wz);
			if (kAsync)
				load_rows_async(q00, q10, q01, q11, ba);
			else
			{
				q00 = load_row(ba);
				q10 = load_row(ba + 10);
				q01 = load_row(ba + 50);
				q11 = load_row(ba + 60);
			}
		}

		// ---- probe outcome (frag:234-247); needs the probe byte only ---------------------------------------------------
		float skip = 0.0f;
		auto  probe_outcome = [&]() {
			// frag:221 r = clamp(float(u_i) - u, -1, 0); FREE: r = -fract(u) (above the loop), and "+ r" becomes "- fract(u)"
			float bx, by, bz;        // frag:239 step(0, s) / frag:242 step(0, -s) + sign(s) * dist
			if (SKIP == VKV_SKIP_BLOCK)
				bx = (R.six < 0.0f) ? 0.0f : 1.0f, by = (R.siy < 0.0f) ? 0.0f : 1.0f, bz = (R.siz < 0.0f) ? 0.0f : 1.0f;
			else
			{        // = fd for s > 0, 1 - fd for s < 0: one fma with per-ray constants (exact: the product is exact)
				const float fd = (float) dist;
				bx = __builtin_fmaf(sgx, fd, ofx), by = __builtin_fmaf(sgy, fd, ofy), bz = __builtin_fmaf(sgz, fd, ofz);
			}
			float ax, ay, az;
			if (FREE)
			{
				ax = (bx - __builtin_amdgcn_fractf(ux)) * R.six;
				ay = (by - __builtin_amdgcn_fractf(uy)) * R.siy;
				az = (bz - __builtin_amdgcn_fractf(uz)) * R.siz;
			}
			else
			{
				ax = (bx + __builtin_amdgcn_fmed3f((float) uix - ux, -1.0f, 0.0f)) * R.six;
				ay = (by + __builtin_amdgcn_fmed3f((float) uiy - uy, -1.0f, 0.0f)) * R.siy;
				az = (bz + __builtin_amdgcn_fmed3f((float) uiz - uz, -1.0f, 0.0f)) * R.siz;
			}
			// a NaN component (0 * inf on an axis-parallel ray) counts as +inf: minNum ignores it; all three cannot be NaN (a direction has a
			// non-zero component)
			// (no cap on m: the oracle's int(ceil(m)) saturates for a huge m, here the position becomes huge or +inf - either way past the end of
			// the ray, and the position a ray ends with is not an output)
			const float m = __builtin_fminf(__builtin_fminf(ax, ay), az);
			skip          = __builtin_fmaxf(1.0f, __builtin_ceilf(m));        // a NaN m gives 1 as max(1, (int) NaN = 0) does
		};

		// ---- sample outcome (frag:272-284) ---------------------------------------------------------------------------
		float    intensity = 0.0f, gradient = 1.0f;
		uint32_t ab = 0, texel = 0;
		float    a = 0.0f, c = 0.0f;
		auto sample_outcome = [&]() {
			if (kTf)
			{        // separable transfer function, table addresses straight from the filtered values: index int(u * 1024) & ~3 without a clamp (u <= 1: a
				 // filter of bytes / 255; the tables have a 257th entry), alpha byte without a clamp (ai, ag <= 1 by construction of the tables)
				const char *ai_tab = reinterpret_cast<const char *>(L.s.ai), *ag_tab = reinterpret_cast<const char *>(L.s.ag);
				float       g_unused;
				if (GRAD == 1)
					packed_filter_cvt<true, true>(q00, q10, q01, q11, wx, wy, wz, intensity, gradient);
				else
					packed_filter_cvt<false, true>(q00, q10, q01, q11, wx, wy, wz, intensity, g_unused);
				const float ai = *reinterpret_cast<const float *>(ai_tab + ((uint32_t) (int) intensity & ~3u));        // intensity, gradient: * 1024 here
				float       ag = GRAD == 0 ? L.s.ag[255] : 0.0f;
				if (GRAD == 1)
					ag = *reinterpret_cast<const float *>(ag_tab + ((uint32_t) (int) gradient & ~3u));
				ab              = (uint32_t) ((ai * ag) * 255.0f);        // <= 255: ai, ag <= 1 (k_tf_tables_init)
				const float2 pr = L.s.pair[ab];
				a = pr.x, c = pr.y;
				return;
			}
			if (kHoist)
			{
				float unused;
				if (GRAD == 1)
					packed_filter_cvt<true>(q00, q10, q01, q11, wx, wy, wz, intensity, gradient);
				else
					packed_filter_cvt<false>(q00, q10, q01, q11, wx, wy, wz, intensity, unused);
			}
			else if (!probe)
			{
				float unused;
				if (PACKED)
					sample_packed<false>(A.packed, W, H, D, A.pmx, A.pmy, posx, posy, posz, intensity, unused);
				else
				{
					intensity = sample_linear(A.vol, W, H, D, posx, posy, posz);
					if (GRAD == 1)
						gradient = sample_linear(A.grad, W, H, D, posx, posy, posz);
				}
				if (GRAD == 2)
				{        // frag:92-97
					const float dix = 1.0f / (float) W, diy = 1.0f / (float) H, diz = 1.0f / (float) D;
					float       t1, t2, t3, t4;
					if (PACKED)
					{
						sample_packed<false>(A.packed, W, H, D, A.pmx, A.pmy, posx + dix, posy - diy, posz - diz, t1, unused);
						sample_packed<false>(A.packed, W, H, D, A.pmx, A.pmy, posx - dix, posy - diy, posz + diz, t2, unused);
						sample_packed<false>(A.packed, W, H, D, A.pmx, A.pmy, posx - dix, posy + diy, posz - diz, t3, unused);
						sample_packed<false>(A.packed, W, H, D, A.pmx, A.pmy, posx + dix, posy + diy, posz + diz, t4, unused);
					}
					else
					{
						t1 = sample_linear(A.vol, W, H, D, posx + dix, posy - diy, posz - diz);
						t2 = sample_linear(A.vol, W, H, D, posx - dix, posy - diy, posz + diz);
						t3 = sample_linear(A.vol, W, H, D, posx - dix, posy + diy, posz - diz);
						t4 = sample_linear(A.vol, W, H, D, posx + dix, posy + diy, posz + diz);
					}
					const float gx = (((t1 - t2) - t3) + t4) * 0.25f;
					const float gy = (((-t1 - t2) + t3) + t4) * 0.25f;
					const float gz = (((-t1 + t2) - t3) + t4) * 0.25f;
					const float len = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz);
					gradient = g_clamp(len * A.grad_modifier, 0.0f, 1.0f);
				}
			}
			// get_color (transfer_function.glsl:35-38), NEAREST: intensity and gradient are >= 0 and never NaN here (a filter of bytes;
			// a clamped length), so clamp(int(floor(u * 256)), 0, 255) is min(int(u * 256), 255)
			const uint32_t ti = (uint32_t) min((int) (intensity * 256.0f), 255), tg = GRAD == 0 ? 255u : (uint32_t) min((int) (gradient * 256.0f), 255);
			if (SEP)
			{
				ab              = tf_separable_alpha(L.s.ai[ti], L.s.ag[tg]);
				const float2 pr = L.s.pair[ab];
				a = pr.x, c = pr.y;
			}
			else
			{
				const uint32_t tidx = tg * 256u + ti;
				if (!probe)
				{
					if (A.tf_bits)
					{
						if ((L.g.bits[tidx >> 5] >> (tidx & 31u)) & 1u)
							texel = reinterpret_cast<const uint32_t *>(A.tf)[tidx];
					}
					else
						texel = reinterpret_cast<const uint32_t *>(A.tf)[tidx];
				}
				ab = texel >> 24;
				a  = L.g.alpha[ab];
			}
		};

		// ---- the frag's state update (frag:224-310) under EXEC: plain moves and adds instead of selects ------------------
		__builtin_amdgcn_wave_barrier();        // emits nothing; keeps the load blocks above apart from the blocks below (left to itself the
		                                        // compiler sinks each load into its block: the footprint would be requested after the probe outcome)
		// the probe side FIRST (its byte is the oldest load in flight, and its outcome runs while the footprint gathers are under way): the
		// compiler lays out the side it is told to expect first (without the hint it starts with the sample side; two separate ifs kept the
		// order too, at the price of a copy of the loop position and three mask operations per iteration).  tools/isa_march_loop.py --order
		// checks the order in a listing; the result does not depend on it (wait_rows waits for everything)
		if (__builtin_expect(probe, 1))
		{
			if (kAsync)
			{        // the byte is the oldest load in flight: behind it the four footprint dwords, or nothing
				wait_oldest_of_five_or_one(dist, live);
			}
			probe_outcome();        // an empty side is skipped by the branch the compiler puts around it (s_cbranch_execz)
			if (kCounts)
				++R.n_dist;
			// frag:244-247 (dist > 0: skip) and frag:253-261 (occupied cell: step back).  The step-back side is three instructions: as
			// selects next to the skip side's EXEC-masked block, not as a block of its own (one region and one branch less per probe)
			const float back_to = max_f32_raw(i - lback, li_min);
			const bool  hit     = dist == 0u;
			li                  = back_to;
			if (!hit)
				li = i + skip;
			occ  = hit ? true : occ;
			ul   = hit ? cell : ul;
		}
		else
		{
			if (kAsync)
				wait_rows(q00, q10, q01, q11);
			sample_outcome();
			if (kCounts)
				++R.n_vol;
			occ        = ab > 0u;        // frag:276
			bool ended = false;
			// kTf: ONE wave-uniform branch around the blend instead of an EXEC-masked block (two EXEC writes less per sample iteration).  For a
			// lane whose sample is empty the blend is a no-op by its arithmetic: alpha byte 0 reads pair[0] = {0, 0} (stage_tables_er), and
			// fma(om, 0, x) = x exactly; a > 0 and alpha > 0.99 are false for it (alpha only passes 0.99 when the ray ends there)
			const bool blend = kTf ? __builtin_amdgcn_ballot_w64(occ) != 0ull : occ;
			if (blend)
			{
				if (SKIP != VKV_SKIP_NONE)
					ul = (kTf && !occ) ? ul : cell;
				const float om = 1.0f - R.a;        // frag:287
				if (SEP)
					grey = __builtin_fmaf(om, c, grey);
				else
				{
					const float r_ = L.g.unorm[texel & 255u] * a, g_ = L.g.unorm[(texel >> 8) & 255u] * a, b_ = L.g.unorm[(texel >> 16) & 255u] * a;
					R.r = __builtin_fmaf(om, r_, R.r), R.g = __builtin_fmaf(om, g_, R.g), R.b = __builtin_fmaf(om, b_, R.b);
				}
				R.a = __builtin_fmaf(om, a, R.a);
				if (a > 0.0f)
					lfirst = i;
				if (ERT && !kTf)        // frag:293-299; the frag's "alpha = 1" of a terminated ray happens behind the loop (alpha only passes 0.99 here, and then the ray ends)
					ended = R.a > 0.99f;
			}
			if (ERT && kTf)        // (behind the uniform branch: no value to merge at its end, so no jump around the rare side)
				ended = R.a > 0.99f;
			if (kCounts && !occ)
				++R.n_empty;
			// frag:308-309 ++i, i_min = i; a terminated ray leaves the loop through its position (the position a ray ends with is not an
			// output, nor is i_min)
			li_min = i + 1.0f;
			li     = ended ? ln : li_min;
		}
		// the frame time is the critical path of the wave with the longest ray: once a wave has run 48 iterations it is one of
		// those, so let it win instruction arbitration against the younger waves on its SIMD
		if (__builtin_amdgcn_readfirstlane(++iter) == 48u)
			__builtin_amdgcn_s_setprio(3);
	}
	if (kStamp)
	{        // (all lanes of the wave that entered are back here, each with the stamp of ITS last iteration: per lane, no readfirstlane)
		const uint32_t dt = (uint32_t) __builtin_amdgcn_s_memtime() - stamp_prev;
		if (stamp_kind == 0u)
			stamp.sum[0] += dt, ++stamp.cnt[0];
		else if (stamp_kind == 1u)
			stamp.sum[1] += dt, ++stamp.cnt[1];
		else if (stamp_kind == 2u)
			stamp.sum[2] += dt, ++stamp.cnt[2];
	}
	R.i = (int) li, R.i_min = (int) li_min, R.first_hit = (int) lfirst;
	if (ERT)
		R.a = R.a > 0.99f ? 1.0f : R.a;
	if (SEP)
		R.r = grey, R.g = grey, R.b = grey;
}

// Computed start order of a schedule over a tile rectangle (RayMarchArgs.order_h; wave-uniform, once per workgroup, no table): which schedule
// entry is started r-th.  Ring by ring from the innermost ring of the w x h rectangle to its border (ring j = the tiles with j tiles between them
// and the nearest border; the rectangle inside ring j holds (w - 2j)(h - 2j) tiles, so the ring of rank r follows from a square root), inside a
// ring clockwise from its top left tile.  The volume's silhouette and the empty corners of the rectangle start last, as with the centre-first
// table of a whole-image schedule - but a rectangle's size changes with the camera, and a table per size does not pay (a camera that moves
// gives every frame in flight its own: profiles/r6_rect_schedules.txt).  Launches of 2 - 4 frames on one stream: -11 % against the plain order.
__device__ __forceinline__ uint32_t start_entry(const RayMarchArgs &A, uint32_t r)
{
	const int w = (int) A.tiles_x, h = (int) A.order_h, rings = (min(w, h) + 1) >> 1;
	auto      inner = [&](int j) { return (w - 2 * j > 0 && h - 2 * j > 0) ? (uint32_t) ((w - 2 * j) * (h - 2 * j)) : 0u; };        // tiles inside ring j - 1
	const float d = (float) (w - h);
	int         j = (int) (((float) (w + h) - __builtin_sqrtf(d * d + 4.0f * (float) r)) * 0.25f);
	j             = max(0, min(j, rings - 1));
#pragma unroll
	for (int it = 0; it < 2; ++it)        // the float estimate is off by one at most
	{
		if (j + 1 < rings && inner(j + 1) > r)
			++j;
		if (j > 0 && inner(j) <= r)
			--j;
	}
	const int wj = w - 2 * j, hj = h - 2 * j;
	int       q = (int) (r - inner(j + 1)), tx, ty;
	if (hj == 1 || q < wj)
		tx = j + q, ty = j;        // top edge, left to right (all of a one-row ring)
	else if ((q -= wj) < hj - 1)
		tx = j + wj - 1, ty = j + 1 + q;        // right edge, downwards
	else if ((q -= hj - 1) < wj - 1)
		tx = j + wj - 2 - q, ty = j + hj - 1;        // bottom edge, right to left
	else
		q -= wj - 1, tx = j, ty = j + hj - 2 - q;        // left edge, upwards
	return (uint32_t) __builtin_amdgcn_readfirstlane(ty * w + tx);
}

// the body of one workgroup: 16x16 pixels of the frame described by A; `bid` is the workgroup's id inside that frame's grid
// FILL: the batch kernel's workgroups also fill the tiles outside a fill_outside schedule's rectangle (the single-frame kernel takes its argument
// block by value, and this code in it made the compiler keep the whole 1.8 KB block in scratch: its launcher renders the whole-image schedule instead)
template <int SKIP, bool ERT, int GRAD, bool PACKED, uint32_t LF, int WPB = 4, bool FILL = false>
__device__ __forceinline__ void lean_block(const RayMarchArgs &A, uint32_t bid, RmLds &L)
{
	// Hardware deals workgroup ids round-robin over the 8 XCDs (own L2 each).  XCD x = id & 7 marches the schedule's tiles
	// k = x, x + 8, x + 16, ... one after the other: neighbouring workgroups of an XCD share a tile (L2 locality) while the tiles
	// of the frame are spread evenly over the XCDs (ESS makes screen regions differ >10x in cost).
	// WPB = waves per workgroup (4: a workgroup is a 16x16 block; 2 / 1: a half / a quarter of it, its parts stay on one XCD)
	constexpr uint32_t kParts = 4 / WPB;
	const uint32_t x = bid & 7u, idx = (bid >> 3) / kParts, part = (bid >> 3) % kParts;
	const uint32_t rank = (idx / A.blocks_per_tile) * 8u + x, sb = idx % A.blocks_per_tile;
	if (rank >= A.tile_count)
		return;
	// Tiles are STARTED centre of the image first (tile_order, built by the launcher): the volume sits there, so the tiles with the
	// long rays — the critical path of the launch — start at once and the cheap border tiles fill the tail.  Any order gives the same
	// frame; on C3 this one shortens a single frame's launch from 0.324 to 0.306 ms and the tail of an 8-frame launch from 0.18 to
	// 0.03 ms (bench.py).  Several single-frame launches in flight on their own streams prefer the plain order
	// (VKV_RAYMARCH_TILE_ORDER=linear: 0.157 vs 0.167 ms per frame with three in flight) - their heavy centres then do not coincide.
	const uint32_t k = A.tile_order ? A.tile_order[rank] : (A.order_h ? start_entry(A, rank) : rank);
	if (k >= A.tile_count)
		return;        // never with a well-formed order; keeps a damaged one (a target shared by two streams without an event) from becoming a wild address
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t       px, py, o;
	const uint32_t rb = (part * WPB + wave) * 64u + lane;        // ray of the block this lane marches: the lane's own pixel of its 8x8 quadrant
	if (FILL && A.fill_tiles != 0u)
	{
		// VkvTileSchedule.fill_outside: the tiles of the image outside the schedule's rectangle get the no-fragment result from the workgroups that
		// render (no workgroup per empty tile: half of a C3 frame's tiles, 4 % of its time).  Outside tile j, row-major over the image with the
		// rectangle left out: the rows above it, the parts left and right of it, the rows below.  Wave-uniform arithmetic, once per workgroup.
		// (at most kFillPerTile outside tiles per rendering tile: the launcher renders the whole-image schedule when the rectangle is smaller than that;
		// a loop of unknown length here keeps the single-frame kernels' by-value argument block in scratch)
#pragma unroll
		for (uint32_t it = 0; it < kFillPerTile; ++it)
		{
			const uint32_t j = k + it * A.tile_count;
			if (j >= A.fill_tiles)
				break;
			const uint32_t top = A.rect_ty0 * A.img_tiles_x, per_row = A.img_tiles_x - A.tiles_x, mid = A.rect_th * per_row;
			uint32_t       tx, ty;
			if (j < top)
				ty = j / A.img_tiles_x, tx = j % A.img_tiles_x;
			else if (j - top < mid)
			{
				const uint32_t q = j - top, c = q % per_row;
				ty = A.rect_ty0 + q / per_row, tx = c < A.rect_tx0 ? c : c + A.tiles_x;
			}
			else
			{
				const uint32_t q = j - top - mid;
				ty = A.rect_ty0 + A.rect_th + q / A.img_tiles_x, tx = q % A.img_tiles_x;
			}
			const uint32_t bx0 = tx * A.tile_w + (sb % A.blocks_per_tile_x) * 16u, by0 = ty * A.tile_h + (sb / A.blocks_per_tile_x) * 16u;        // the block's first pixel
			if (A.fill_rgba8_rows != 0u && bx0 + 16u <= A.img_w)
			{
				// the production frame (RGBA8 only, no blend state, rows of the image 16-byte aligned: the launcher's flag): a 16-pixel row of the block
				// is four 16-byte stores, the block one store instruction of its first wave instead of four
				if (rb < 64u && by0 + (rb >> 2) < A.img_h)
				{
					typedef uint32_t v4u __attribute__((ext_vector_type(4)));
					const v4u zero = {0u, 0u, 0u, 0u};
					__builtin_nontemporal_store(zero, reinterpret_cast<v4u *>(A.out_rgba8 + ((size_t) (by0 + (rb >> 2)) * A.img_w + bx0 + (rb & 3u) * 4u) * 4u));
				}
			}
			else
			{
				const uint32_t fx = bx0 + (rb & 15u), fy = by0 + (rb >> 4);
				if (fx < A.img_w && fy < A.img_h)
				{
					Ray F;
					ray_clear(F);
					F.o = fy * A.img_w + fx;
					ray_finish(A, F, false);
				}
			}
		}
	}
	const bool     inside = block_pixel<1>(A, k * A.blocks_per_tile + sb, rb, px, py, o);
	Ray R;
	R.o = o;
	const unsigned long long t_start = A.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
	const unsigned long long c_start = A.trace ? __builtin_amdgcn_s_memtime() : 0ull;        // shader clock (with t_start: the clock rate under load)
	bool marched = false;
	if (inside)
	{        // pixels outside the screen bound of the volume's box skip the ray set-up (a third of a C3 frame)
		if (px >= A.cull_x0 && px <= A.cull_x1 && py >= A.cull_y0 && py <= A.cull_y1)
			marched = ray_setup<SKIP>(A, px, py, R);
		else
			ray_clear(R);
	}
	uint32_t  iter  = 0;
	LeanStamp stamp = {};
	// 60 % of the workgroups of a frame hold no ray that enters the volume: they skip the LDS tables (and their barrier) altogether
	if (wg_any(marched))
	{
		if ((LF & kLeanLut) != 0 && PACKED && GRAD != 2)
		{        // before the barrier of stage_tables_er
			if ((LF & kLeanFull) != 0 && tf_is_separable(A))
				stage_full_lut(A, L);
			else
				stage_addr_lut(A);
		}
		const bool sep = stage_tables_er(A, L);
		if (marched)
		{
			if (sep)
			{
				constexpr bool kSafe = (LF & kLeanSafe) != 0 && (LF & kLeanFull) != 0 && PACKED && GRAD != 2 && SKIP != VKV_SKIP_NONE && (LF & kLeanStamp) == 0;
				if constexpr (kSafe)
				{
					if (lean_free_wave(A, R))
						lean_march<SKIP, ERT, GRAD, PACKED, true, LF, true>(A, R, L, iter, stamp);
					else
						lean_march<SKIP, ERT, GRAD, PACKED, true, LF>(A, R, L, iter, stamp);
				}
				else
					lean_march<SKIP, ERT, GRAD, PACKED, true, LF>(A, R, L, iter, stamp);
			}
			else
				lean_march<SKIP, ERT, GRAD, PACKED, false, LF>(A, R, L, iter, stamp);
		}
	}
	if (A.tile_cost)
	{        // the tile costs as much as its longest wave; a lane's `iter` stops counting when its ray ends
		uint32_t it = iter;
		for (int o2 = 32; o2 > 0; o2 >>= 1)
			it = max(it, (uint32_t) __shfl_xor((int) it, o2));
		if (it != 0u && lane == (uint32_t) __builtin_ctzll(__ballot(1)))
			atomicMax(&A.tile_cost[k], it);
	}
	if (!inside)
		return;
	ray_finish(A, R, marched);
	if (A.trace)
	{        // diagnostic only: per-wave timeline (100 MHz clock); the values never feed an output
		uint32_t it = iter;
		for (int o2 = 32; o2 > 0; o2 >>= 1)
			it = max(it, (uint32_t) __shfl_xor((int) it, o2));
		const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
		uint32_t clamped_max = stamp.clamped;        // a lane counts the rare-branch iterations it was alive in
		for (int o2 = 32; o2 > 0; o2 >>= 1)
			clamped_max = max(clamped_max, (uint32_t) __shfl_xor((int) clamped_max, o2));
		if ((LF & kLeanStamp) != 0 && iter == it && lane == (uint32_t) __builtin_ctzll(__ballot(iter == it)))
		{        // the stamps of the lane whose ray lived through every iteration of the wave (the others stop counting when their ray ends)
			unsigned long long *rec = A.trace + ((size_t) blockIdx.x * WPB + wave) * kTraceWords;
			rec[4] = ((unsigned long long) stamp.cnt[0] << 32) | stamp.sum[0], rec[5] = ((unsigned long long) stamp.cnt[1] << 32) | stamp.sum[1];
			rec[8] = ((unsigned long long) stamp.cnt[2] << 32) | stamp.sum[2];
		}
		else if ((LF & kLeanStamp) == 0 && lane == (uint32_t) __builtin_ctzll(__ballot(1)))
		{
			unsigned long long *rec = A.trace + ((size_t) blockIdx.x * WPB + wave) * kTraceWords;
			rec[4] = 0, rec[5] = 0, rec[8] = 0;
		}
		if (lane == (uint32_t) __builtin_ctzll(__ballot(1)))
		{
			unsigned long long *rec = A.trace + ((size_t) blockIdx.x * WPB + wave) * kTraceWords;        // per launch (a batch: all its frames)
			rec[0] = t_start, rec[1] = t_end, rec[2] = it, rec[3] = ((unsigned long long) __builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | (k * A.blocks_per_tile + sb);
			rec[6] = c_start, rec[7] = __builtin_amdgcn_s_memtime(), rec[9] = clamped_max;
		}
	}
}

// (the single-frame kernel is not held to 64 VGPRs like the batch kernel below: one frame at a time is bound by the longest wave's
// dependent chain, not by the number of resident waves, and at 64 the march loop of some instantiations spills)
// No packed fp32 instructions in the lean kernels (round 5).  Under plain -O3 the SLP vectoriser pairs the loop's isomorphic fp32 operations
// into v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32.  On gfx950 a packed fp32 instruction issues at half the rate of a plain one (nothing gained),
// needs its operands in aligned register pairs (v_mov) and a wait state when it follows its producer (s_nop): the bench kernel's march loop is
// 124 plain VALU + 36 SALU instructions without them, 109 + 44 with them - and 1.7 % faster without (profiles/r5_ab_no_packed_fp32.txt;
// MI355X_MICROARCH.md lists packed fp32 as an anti-lever for the same reason).  The target attribute switches the feature off for these
// kernels only; same IEEE operations, same bits.  (The batch kernel only: the single-frame kernel built this way keeps its 1.8 KB by-value
// argument block in scratch - its packed and its plain build measure the same for one frame alone.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VKV_LAB_PACKED_FP32)        // (-DVKV_LAB_PACKED_FP32: the A/B build of profiles/r5_ab_no_packed_fp32.txt)
#define VKV_NO_PACKED_FP32 __attribute__((target("no-packed-fp32-ops")))
#else        // (the host pass of the translation unit does not know the feature)
#define VKV_NO_PACKED_FP32
#endif

template <int SKIP, bool ERT, int GRAD, bool PACKED, uint32_t LF>
__global__ void __launch_bounds__(256) k_raymarch_lean(const RayMarchArgs A)
{
	lean_lds_check();
	RmLds &L = lean_lds();
	lean_block<SKIP, ERT, GRAD, PACKED, LF>(A, blockIdx.x, L);
}

// Several frames in one launch (vkv_render_batch): frame f = (id / 8) % n takes every n-th group of eight workgroup ids, so the
// frames advance side by side and the long tail of each (a few waves with hundreds of dependent events) is covered by the bulk
// of the others — what a renderer with frames in flight gets from several queues, without depending on queue scheduling.
// The argument blocks live in device memory (n x 1.7 KB does not fit the kernel-argument segment).
// held to 64 VGPRs = 8 waves per SIMD (the fast ray set-up first came out at 66: seven waves, 10 % slower with frames in flight: 0.1246
// against 0.1101 ms per C3 frame).  Not the on-the-fly gradient variant (five trilinear taps per sample, ~100 VGPRs: it would spill) and
// not the kernels without empty-space skipping (45 VGPRs anyway, but the occupancy target changes their schedule: dense sampling with 8
// frames per launch 2.63 against 2.14 ms per frame), and not the kernels without the LDS address tables (volumes without a packed image: at 64
// registers their loop spills a few dwords, and a kernel with scratch makes its first launch allocate)
// (Round 3, measured and dropped: workgroups of TWO waves - half a 16x16 block - with the two-level tables, so that wave slots are handed
// back in smaller pieces: three single-frame launches in flight gain 5 % in the lab (0.1181 against 0.1242 ms per frame, variants 112 / 21),
// this kernel loses 6 % (0.1167 against 0.1099; the two-level tables alone 0.1116): twice the workgroups to set up and to stage tables for.)
template <int SKIP, bool ERT, int GRAD, uint32_t LF>
__global__ void __launch_bounds__(256) VKV_NO_PACKED_FP32 __attribute__((amdgpu_waves_per_eu((GRAD == 2 || SKIP == VKV_SKIP_NONE || (LF & kLeanLut) == 0) ? 1 : 8))) k_raymarch_lean_batch(const RayMarchArgs *__restrict__ frames, uint32_t n, uint32_t groups_per_frame)
{
	lean_lds_check();
	RmLds &        L = lean_lds();
	const uint32_t g = blockIdx.x >> 3;
	// groups_per_frame == 0: frames interleaved in groups of eight workgroups; otherwise one frame after the other (A/B switch of the launcher)
	const uint32_t f = groups_per_frame == 0 ? g % n : g / groups_per_frame, gi = groups_per_frame == 0 ? g / n : g % groups_per_frame;
	// (the anisotropic kernels that keep the per-pixel counters have no room under the 64-VGPR cap for the second march loop: they would spill)
	constexpr uint32_t kLfBatch = (SKIP == VKV_SKIP_ANISOTROPIC_DISTANCE && (LF & kLeanNoCounts) == 0) ? (LF & ~kLeanSafe) : LF;
	lean_block<SKIP, ERT, GRAD, true, kLfBatch, 4, true>(frames[f], (gi << 3) | (blockIdx.x & 7u), L);
}

// ---------------------------------------------------------------------------------------------------------------
// k_raymarch_lean_pull — the batch launch with resident workgroups whose WAVES pull 8x8 units from per-XCD ticket counters.
//   * A workgroup of the tile-per-workgroup kernels keeps its 20 KB of LDS tables until its longest wave is done, so a CU that could hold 32
//     waves holds 20 on average (wave trace of an 8-frame launch) and a launch ends with 150 us of declining occupancy.  Here a wave that
//     has finished its unit takes the next one: the workgroups stay, every wave slot stays busy until the tickets run out, the tables are
//     staged once per workgroup (2048 times per launch instead of 26 000), and workgroups without a marching ray cost nothing.
//   * Requires every frame of the launch to share the tables (same packed image, TF tables, opacity table: the launcher compares them).
//   * Ticket v of queue q (one queue per XCD, as the static kernels deal the tiles): unit w = v % upt of frame (v / upt) % n of the
//     queue's (v / (upt n))-th tile, i.e. the frames advance side by side through the centre-first start order.  A wave takes from its own
//     XCD's queue first and from the others once that is empty.  The eight counters sit in front of the argument blocks in the stream's
//     scratch buffer and are zeroed by the same upload.
// No lane compaction: a wave marches one 8x8 unit at a time with the loop of the tile kernels (lean_march).
// ---------------------------------------------------------------------------------------------------------------
// the pull loop of one wave, one copy per transfer-function path (each copy holds one march loop: half the register pressure)
template <int SKIP, bool ERT, int GRAD, uint32_t LF, bool SEP>
__device__ __forceinline__ void pull_units(const RayMarchArgs *__restrict__ frames, uint32_t n, uint32_t *__restrict__ heads, const RmLds &L)
{
	const RayMarchArgs &A0   = frames[0];
	const uint32_t      lane = threadIdx.x & 63u;
	const uint32_t      upt  = A0.blocks_per_tile * 4u;        // 8x8 units per tile
	uint32_t            q = blockIdx.x & 7u, open_queues = 0xffu;
	while (open_queues != 0u)
	{
		if (((open_queues >> q) & 1u) == 0u)
		{
			q = (q + 1u) & 7u;
			continue;
		}
		uint32_t v = 0;
		if (lane == 0)
			v = atomicAdd(&heads[q * vkv::kPullHeadStride], 1u);
		v = __builtin_amdgcn_readfirstlane(v);
		const uint32_t tiles_q = A0.tile_count > q ? (A0.tile_count - q + 7u) >> 3 : 0u;
		if (v >= tiles_q * upt * n)
		{
			open_queues &= ~(1u << q);
			q = (q + 1u) & 7u;
			continue;
		}
		const uint32_t      w = v % upt, f = (v / upt) % n, rank = (v / (upt * n)) * 8u + q;
		const RayMarchArgs &A = frames[f];
		const uint32_t      k = A.tile_order ? A.tile_order[rank] : rank;
		if (k >= A.tile_count)
			continue;
		uint32_t            px, py, o;
		const bool          inside = block_pixel<1>(A, k * A.blocks_per_tile + (w >> 2), (w & 3u) * 64u + lane, px, py, o);
		Ray                 R = {};        // every field defined per unit: nothing of the previous unit's ray is carried round the loop
		R.o = o;
		const unsigned long long t_start = A.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
		bool                     marched = false;
		if (inside)
		{
			if (px >= A.cull_x0 && px <= A.cull_x1 && py >= A.cull_y0 && py <= A.cull_y1)
				marched = ray_setup<SKIP>(A, px, py, R);
			else
				ray_clear(R);
		}
		uint32_t                 iter    = 0;
		const unsigned long long t_setup = A.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
		LeanStamp                unused_stamp = {};
		if (marched)
			lean_march<SKIP, ERT, GRAD, true, SEP, LF>(A, R, L, iter, unused_stamp);
		const unsigned long long t_march = A.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
		__builtin_amdgcn_s_setprio(0);        // lean_march raises the priority of a long wave: back to normal for the next unit
		if (A.tile_cost)
		{
			uint32_t it = iter;
			for (int o2 = 32; o2 > 0; o2 >>= 1)
				it = max(it, (uint32_t) __shfl_xor((int) it, o2));
			if (it != 0u && lane == 0u)
				atomicMax(&A.tile_cost[k], it);
		}
		if (inside)
			ray_finish(A, R, marched);
		if (A.trace)
		{        // diagnostic only: one record per unit
			uint32_t it = iter;
			for (int o2 = 32; o2 > 0; o2 >>= 1)
				it = max(it, (uint32_t) __shfl_xor((int) it, o2));
			const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
			if (lane == 0)
			{
				unsigned long long *rec = A.trace + ((size_t) ((v / upt) * 8u + q) * upt + w) * kTraceWords;
				rec[0] = t_start, rec[1] = t_end, rec[2] = it, rec[3] = ((unsigned long long) __builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | (rank * upt + w);
				rec[4] = t_setup, rec[5] = t_march;
				for (int x = 6; x < kTraceWords; ++x)
					rec[x] = 0;
			}
		}
	}
}

template <int SKIP, bool ERT, int GRAD, uint32_t LF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_raymarch_lean_pull(const RayMarchArgs *__restrict__ frames, uint32_t n, uint32_t *__restrict__ heads)
{
	lean_lds_check();
	RmLds &             L  = lean_lds();
	const RayMarchArgs &A0 = frames[0];
	if ((LF & kLeanLut) != 0 && GRAD != 2)
	{        // before the barrier of stage_tables_er
		if ((LF & kLeanFull) != 0 && tf_is_separable(A0))
			stage_full_lut(A0, L);
		else
			stage_addr_lut(A0);
	}
	if (stage_tables_er(A0, L))
		pull_units<SKIP, ERT, GRAD, LF, true>(frames, n, heads, L);
	else
		pull_units<SKIP, ERT, GRAD, LF, false>(frames, n, heads, L);
}

namespace vkv
{
// raymarch.hip: VkvRenderParams -> kernel arguments (shared with tools/lab)
int fill_render_args(vkv_ctx *ctx, const VkvRenderParams *P, const float *alpha_lut, RayMarchArgs &a, hipStream_t s, const VkvTuning &T, bool setup, bool batch);
}        // namespace vkv
