/*
 * vkv_oracle.c — CPU restatement of the reference's ray-caster hot path (see vkv_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY — never linked into, called from, or shipped with the product.
 * PARITY UNPINNED — the reference has no golden vectors and cannot run here (vkv_oracle.h).
 *
 * Every function cites the reference file:line it follows.  Where the reference relies on
 * Vulkan fixed-function behaviour with implementation latitude (trilinear filtering, UNORM
 * conversion, rasteriser interpolation) this file PINS one behaviour; the pins are listed in
 * DESIGN.md §"Pinned numerics" and repeated next to the code.  Build: -O2 -ffp-contract=off,
 * no fast-math; fused multiply-adds appear only where fmaf() is written out.
 */
#define _GNU_SOURCE
#include "vkv_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* GLSL built-ins with their specification semantics                                           */
/* ------------------------------------------------------------------------------------------- */
static inline float g_min(float x, float y) { return (y < x) ? y : x; }        /* GLSL min */
static inline float g_max(float x, float y) { return (x < y) ? y : x; }        /* GLSL max */
static inline float g_clamp(float x, float lo, float hi) { return g_min(g_max(x, lo), hi); }
static inline float g_step(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }
static inline float g_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
static inline int   i_min(int a, int b) { return a < b ? a : b; }
static inline int   i_max(int a, int b) { return a > b ? a : b; }
static inline int   i_clamp(int x, int lo, int hi) { return i_min(i_max(x, lo), hi); }

#define VKV_INV255 (1.0f / 255.0f)

static inline size_t vidx(int x, int y, int z, int W, int H) { return ((size_t) z * (size_t) H + (size_t) y) * (size_t) W + (size_t) x; }

/* ------------------------------------------------------------------------------------------- */
/* Transfer function                                                                           */
/* ------------------------------------------------------------------------------------------- */

/* src/volume_component.cpp:226-240 */
void vkvo_transfer_function_uniform(const VkvVolumeOptions *o, VkvTransferFunctionUniform *u)
{
	u->sampling_factor         = o->sampling_factor;
	u->voxel_alpha_factor      = o->voxel_alpha_factor;
	u->grad_magnitude_modifier = 1.0f;
	u->use_gradient            = (o->gradient_max != o->gradient_min) ? 1u : 0u;
	u->intensity_min           = o->intensity_min;
	u->intensity_range_inv     = 1.0f / (o->intensity_max - o->intensity_min);
	u->gradient_min            = o->gradient_min;
	u->gradient_range_inv      = 1.0f / (o->gradient_max - o->gradient_min);
}

/* src/volume_component.cpp:242-261: greyscale LUT, row = gradient, column = intensity, all four
 * channels = truncated alpha. */
void vkvo_transfer_function_texture(const VkvVolumeOptions *o, uint8_t *tex)
{
	const float i_inv        = 1.0f / (o->intensity_max - o->intensity_min);
	const float g_inv        = 1.0f / (o->gradient_max - o->gradient_min);
	const int   use_gradient = o->gradient_max != o->gradient_min;
	size_t      idx          = 0;
	for (float g = 0; g < 256; ++g)
	{
		for (float i = 0; i < 256; ++i, ++idx)
		{
			/* the reference's lambda is std::min(std::max(x, lo), hi) */
			float   t       = ((i / 255.0f) - o->intensity_min) * i_inv;
			float   alpha_i = fminf(fmaxf(t, 0.0f), 1.0f);
			float   alpha_g = 1.0f;
			if (use_gradient)
			{
				float tg = ((g / 255.0f) - o->gradient_min) * g_inv;
				alpha_g  = fminf(fmaxf(tg, 0.0f), 1.0f);
			}
			float   a     = fminf(fmaxf(alpha_i * alpha_g * 255, 0.0f), 255.0f);
			uint8_t alpha = (uint8_t) a; /* truncation, volume_component.cpp:259 */
			tex[idx * 4 + 0] = tex[idx * 4 + 1] = tex[idx * 4 + 2] = tex[idx * 4 + 3] = alpha;
		}
	}
}

/* ------------------------------------------------------------------------------------------- */
/* Gradient map — shaders/gradient_map.comp:35-41, shaders/get_gradient_compute.glsl:5-23       */
/* ------------------------------------------------------------------------------------------- */

/* imageLoad of an R8_UNORM texel: exactly byte / 255 */
static inline float unorm8(uint8_t b) { return (float) b / 255.0f; }

/* get_gradient() of get_gradient_compute.glsl, on-the-fly branch (:12-20). Returns the float in
 * [0,1] BEFORE the UNORM store. PIN: sums evaluated left to right in the shader's term order
 * k.xyy, k.yyx, k.yxy, k.xxx with k = (1,-1). */
static float gradient_on_the_fly(const uint8_t *vol, int W, int H, int D, int x, int y, int z, float modifier)
{
	const int xm = i_clamp(x - 1, 0, W - 1), xp = i_clamp(x + 1, 0, W - 1);
	const int ym = i_clamp(y - 1, 0, H - 1), yp = i_clamp(y + 1, 0, H - 1);
	const int zm = i_clamp(z - 1, 0, D - 1), zp = i_clamp(z + 1, 0, D - 1);
	const float v1 = unorm8(vol[vidx(xp, ym, zm, W, H)]); /* k.xyy = ( 1,-1,-1) */
	const float v2 = unorm8(vol[vidx(xm, ym, zp, W, H)]); /* k.yyx = (-1,-1, 1) */
	const float v3 = unorm8(vol[vidx(xm, yp, zm, W, H)]); /* k.yxy = (-1, 1,-1) */
	const float v4 = unorm8(vol[vidx(xp, yp, zp, W, H)]); /* k.xxx = ( 1, 1, 1) */
	const float gx = 0.25f * (((v1 - v2) - v3) + v4);
	const float gy = 0.25f * (((-v1 - v2) + v3) + v4);
	const float gz = 0.25f * (((-v1 + v2) - v3) + v4);
	const float len = sqrtf((gx * gx + gy * gy) + gz * gz);
	return g_clamp(len * modifier, 0.0f, 1.0f);
}

/* PIN: the R8_UNORM imageStore rounds to nearest even: (uint8) rintf(g * 255). */
static inline uint8_t store_unorm8(float g) { return (uint8_t) rintf(g * 255.0f); }

void vkvo_gradient_map(const uint8_t *vol, uint8_t *grad, VkvExtent3D e, const VkvTransferFunctionUniform *tf)
{
	const int W = (int) e.width, H = (int) e.height, D = (int) e.depth;
	for (int z = 0; z < D; ++z)
		for (int y = 0; y < H; ++y)
			for (int x = 0; x < W; ++x)
			{
				float g = 1.0f; /* get_gradient_compute.glsl:6-7 */
				if (tf->use_gradient)
					g = gradient_on_the_fly(vol, W, H, D, x, y, z, tf->grad_magnitude_modifier);
				grad[vidx(x, y, z, W, H)] = store_unorm8(g);
			}
}

/* ------------------------------------------------------------------------------------------- */
/* Occupancy map — shaders/occupancy_map.comp:45-73                                            */
/* ------------------------------------------------------------------------------------------- */

/* NEAREST lookup of the 256x256 TF texture (sampler at src/volume_component.cpp:149-151):
 * PIN texel = clamp(int(floor(u * 256)), 0, 255). */
static inline int tf_texel(float u) { return i_clamp((int) floorf(u * 256.0f), 0, 255); }

void vkvo_occupancy_map(const uint8_t *vol, const uint8_t *grad, const uint8_t *tf_rgba8, const VkvTransferFunctionUniform *tf,
                        VkvExtent3D e, uint8_t *map, VkvExtent3D me)
{
	const int W = (int) e.width, H = (int) e.height, D = (int) e.depth;
	const int mw = (int) me.width, mh = (int) me.height, md = (int) me.depth;
	/* src/compute_distance_map.cpp:110-113 */
	const int bx = (W + mw - 1) / mw, by = (H + mh - 1) / mh, bz = (D + md - 1) / md;
	for (int cz = 0; cz < md; ++cz)
		for (int cy = 0; cy < mh; ++cy)
			for (int cx = 0; cx < mw; ++cx)
			{
				const int sx = cx * bx, sy = cy * by, sz = cz * bz;
				const int ex = i_min(sx + bx, W), ey = i_min(sy + by, H), ez = i_min(sz + bz, D);
				uint8_t   cell = 255; /* EMPTY */
				for (int z = sz; z < ez && cell; ++z)
					for (int y = sy; y < ey && cell; ++y)
						for (int x = sx; x < ex; ++x)
						{
							const float intensity = unorm8(vol[vidx(x, y, z, W, H)]);
							float       gradient  = 1.0f;
							if (tf->use_gradient)
								gradient = grad ? unorm8(grad[vidx(x, y, z, W, H)]) :
								                  gradient_on_the_fly(vol, W, H, D, x, y, z, tf->grad_magnitude_modifier);
							const uint8_t alpha = tf_rgba8[((size_t) tf_texel(gradient) * 256 + (size_t) tf_texel(intensity)) * 4 + 3];
							if (alpha > 0)
							{
								cell = 0; /* OCCUPIED */
								break;
							}
						}
				map[vidx(cx, cy, cz, mw, mh)] = cell;
			}
}

/* ------------------------------------------------------------------------------------------- */
/* Occupied-voxel count — shaders/occupied_voxel_count.comp:28-56 (+ _reduce.comp: a plain sum)  */
/* ------------------------------------------------------------------------------------------- */

/* analytic get_color, shaders/transfer_function.glsl:41-43 — this statistic does NOT use the TF texture
 * (occupied_voxel_count.comp:12-15 leaves TRANSFER_FUNCTION_BINDING_TEXTURE undefined) */
static inline float analytic_alpha(const VkvTransferFunctionUniform *tf, float intensity, float gradient)
{
	const float ai = g_clamp((intensity - tf->intensity_min) * tf->intensity_range_inv, 0.0f, 1.0f);
	const float ag = g_clamp((gradient - tf->gradient_min) * tf->gradient_range_inv, 0.0f, 1.0f);
	return ai * ag;
}

uint64_t vkvo_occupied_voxel_count(const uint8_t *vol, const uint8_t *grad, const VkvTransferFunctionUniform *tf, VkvExtent3D e)
{
	const int W = (int) e.width, H = (int) e.height, D = (int) e.depth;
	uint64_t  n = 0;
	for (int z = 0; z < D; ++z)
		for (int y = 0; y < H; ++y)
			for (int x = 0; x < W; ++x)
			{
				const float intensity = unorm8(vol[vidx(x, y, z, W, H)]);
				float       gradient  = 1.0f;
				if (tf->use_gradient)
					gradient = grad ? unorm8(grad[vidx(x, y, z, W, H)]) : gradient_on_the_fly(vol, W, H, D, x, y, z, tf->grad_magnitude_modifier);
				if (analytic_alpha(tf, intensity, gradient) > 0.0f)
					++n;
			}
	return n;
}

/* ------------------------------------------------------------------------------------------- */
/* Chebyshev distance transform — shaders/distance_map.comp:44-109                              */
/* ------------------------------------------------------------------------------------------- */

static void dm_stage(int stage, uint8_t *dist, uint8_t *dist_swap, int W, int H, int D)
{
	if (stage == 0)
	{ /* "Transformation 1" (:57-71); dist and dist_swap alias (compute_distance_map.cpp:156-157) */
		for (int z = 0; z < D; ++z)
			for (int y = 0; y < H; ++y)
			{
				uint32_t g1 = dist_swap[vidx(0, y, z, W, H)];
				for (int x = 1; x < W; ++x)
				{
					uint32_t v = dist_swap[vidx(x, y, z, W, H)];
					uint32_t g = (g1 + 1 < v) ? g1 + 1 : v;
					dist[vidx(x, y, z, W, H)] = (uint8_t) g;
					g1 = g;
				}
				for (int x = W - 2; x >= 0; --x)
				{
					uint32_t v = dist[vidx(x, y, z, W, H)];
					uint32_t g = (g1 + 1 < v) ? g1 + 1 : v;
					dist[vidx(x, y, z, W, H)] = (uint8_t) g;
					g1 = g;
				}
			}
	}
	else if (stage == 1)
	{ /* "Transformation 2" (:72-90): dist -> dist_swap along y */
		for (int z = 0; z < D; ++z)
			for (int x = 0; x < W; ++x)
				for (int y = 0; y < H; ++y)
				{
					uint32_t Dm = dist[vidx(x, y, z, W, H)];
					for (int n = 1; (uint32_t) n < Dm; ++n)
					{
						if (y >= n)
						{
							uint32_t dn = dist[vidx(x, y - n, z, W, H)];
							uint32_t m  = (uint32_t) n > dn ? (uint32_t) n : dn;
							Dm          = Dm < m ? Dm : m;
						}
						if ((y + n) < H && (uint32_t) n < Dm)
						{
							uint32_t dn = dist[vidx(x, y + n, z, W, H)];
							uint32_t m  = (uint32_t) n > dn ? (uint32_t) n : dn;
							Dm          = Dm < m ? Dm : m;
						}
					}
					dist_swap[vidx(x, y, z, W, H)] = (uint8_t) Dm;
				}
	}
	else
	{ /* "Transformation 3" (:91-107): dist_swap -> dist along z */
		for (int y = 0; y < H; ++y)
			for (int x = 0; x < W; ++x)
				for (int z = 0; z < D; ++z)
				{
					uint32_t Dm = dist_swap[vidx(x, y, z, W, H)];
					for (int n = 1; (uint32_t) n < Dm; ++n)
					{
						if (z >= n)
						{
							uint32_t dn = dist_swap[vidx(x, y, z - n, W, H)];
							uint32_t m  = (uint32_t) n > dn ? (uint32_t) n : dn;
							Dm          = Dm < m ? Dm : m;
						}
						if ((z + n) < D && (uint32_t) n < Dm)
						{
							uint32_t dn = dist_swap[vidx(x, y, z + n, W, H)];
							uint32_t m  = (uint32_t) n > dn ? (uint32_t) n : dn;
							Dm          = Dm < m ? Dm : m;
						}
					}
					dist[vidx(x, y, z, W, H)] = (uint8_t) Dm;
				}
	}
}

/* src/compute_distance_map.cpp:142-175 */
void vkvo_distance_map(uint8_t *map, uint8_t *swap, VkvExtent3D me)
{
	const int W = (int) me.width, H = (int) me.height, D = (int) me.depth;
	dm_stage(0, map, map, W, H, D);  /* both bindings = distance image */
	dm_stage(1, map, swap, W, H, D); /* binding 1 = swap */
	dm_stage(2, map, swap, W, H, D);
}

/* shaders/distance_map_anisotropic.comp:31-92; `dist` is binding 0, `dist_swap` binding 1 */
static void dma_stage(int stage, int dir, uint8_t *dist, uint8_t *dist_swap, int W, int H, int D)
{
	if (stage == 0)
	{ /* :44-53 — writes dist from dist_swap (the occupancy map) */
		const int start = dir > 0 ? W - 1 : 0;
		const int end   = dir > 0 ? -1 : W;
		for (int z = 0; z < D; ++z)
			for (int y = 0; y < H; ++y)
			{
				uint32_t g1 = dist_swap[vidx(start, y, z, W, H)];
				for (int x = start; x != end; x -= dir)
				{
					uint32_t v = dist_swap[vidx(x, y, z, W, H)];
					uint32_t g = (g1 + 1 < v) ? g1 + 1 : v;
					dist[vidx(x, y, z, W, H)] = (uint8_t) g;
					g1 = g;
				}
			}
	}
	else if (stage == 1)
	{ /* :55-72 — dist -> dist_swap along y */
		for (int z = 0; z < D; ++z)
			for (int x = 0; x < W; ++x)
				for (int y = 0; y < H; ++y)
				{
					uint32_t m_min = dist[vidx(x, y, z, W, H)];
					for (int n = 1; (uint32_t) n < m_min && n < 255; ++n)
					{
						int yt = y + dir * n;
						if (yt < 0 || yt >= H)
							break;
						uint32_t gn = dist[vidx(x, yt, z, W, H)];
						uint32_t m  = (uint32_t) n > gn ? (uint32_t) n : gn;
						if (m < m_min)
							m_min = m;
					}
					dist_swap[vidx(x, y, z, W, H)] = (uint8_t) m_min;
				}
	}
	else
	{ /* :73-91 — dist_swap -> dist along z */
		for (int y = 0; y < H; ++y)
			for (int x = 0; x < W; ++x)
				for (int z = 0; z < D; ++z)
				{
					uint32_t m_min = dist_swap[vidx(x, y, z, W, H)];
					for (int n = 1; (uint32_t) n < m_min && n < 255; ++n)
					{
						int zt = z + dir * n;
						if (zt < 0 || zt >= D)
							break;
						uint32_t gn = dist_swap[vidx(x, y, zt, W, H)];
						uint32_t m  = (uint32_t) n > gn ? (uint32_t) n : gn;
						if (m < m_min)
							m_min = m;
					}
					dist[vidx(x, y, z, W, H)] = (uint8_t) m_min;
				}
	}
}

/* src/compute_distance_map.cpp:201-252 — occupancy lives in maps[7] */
void vkvo_distance_map_anisotropic(uint8_t *const maps[8], uint8_t *swap, VkvExtent3D me)
{
	const int W = (int) me.width, H = (int) me.height, D = (int) me.depth;
	uint8_t * occ = maps[7];
	/* stage1(idx,dir): dist = maps[idx], dist_swap = occupancy; stage2: dist = maps[idx] (read), dist_swap = swap
	 * (write); stage3: dist = maps[idx] (write), dist_swap = swap (read). */
	dma_stage(0, 1, maps[3], occ, W, H, D);
	dma_stage(1, 1, maps[3], swap, W, H, D);
	dma_stage(2, 1, maps[0], swap, W, H, D);
	dma_stage(2, -1, maps[1], swap, W, H, D);
	dma_stage(1, -1, maps[3], swap, W, H, D);
	dma_stage(2, 1, maps[2], swap, W, H, D);
	dma_stage(2, -1, maps[3], swap, W, H, D);

	dma_stage(0, -1, maps[7], occ, W, H, D);
	dma_stage(1, 1, maps[7], swap, W, H, D);
	dma_stage(2, 1, maps[4], swap, W, H, D);
	dma_stage(2, -1, maps[5], swap, W, H, D);
	dma_stage(1, -1, maps[7], swap, W, H, D);
	dma_stage(2, 1, maps[6], swap, W, H, D);
	dma_stage(2, -1, maps[7], swap, W, H, D);
}

/* src/compute_distance_map.cpp:65-101 */
void vkvo_compute_distance_map(const uint8_t *vol, const uint8_t *grad, const uint8_t *tf_rgba8, const VkvTransferFunctionUniform *tf,
                               VkvExtent3D e, uint8_t *const maps[8], uint8_t *swap, VkvExtent3D me, int32_t skipping_type)
{
	const int aniso = skipping_type == VKV_SKIP_ANISOTROPIC_DISTANCE;
	const int n     = aniso ? 8 : 1;
	vkvo_occupancy_map(vol, grad, tf_rgba8, tf, e, maps[n - 1], me);
	if (aniso)
		vkvo_distance_map_anisotropic(maps, swap, me);
	else if (skipping_type == VKV_SKIP_DISTANCE)
		vkvo_distance_map(maps[0], swap, me);
	/* None / Block: the raw 0/255 occupancy map is used as distance_map[0] (:96-99) */
}

/* ------------------------------------------------------------------------------------------- */
/* Ray-march integrator — shaders/volume_render.frag                                           */
/* ------------------------------------------------------------------------------------------- */

typedef struct
{
	float x, y, z;
} v3;

/* PIN (linear filter, clamp-to-edge, src/volume_component.cpp:139-148): unnormalised coordinate
 * c = pos*dim - 0.5 (one fma), i0 = floor(c), w = c - i0, both texel indices clamped to the
 * image, the eight BYTE values blended in fp32 as fma(w, b - a, a) along x, then y, then z, and
 * the result scaled by 1/255 once.  (Hardware filters with ~8-bit weights; Vulkan leaves the
 * precision to the implementation.) */
static inline float sample_linear(const uint8_t *tex, int W, int H, int D, float px, float py, float pz)
{
	const float cx = fmaf(px, (float) W, -0.5f), cy = fmaf(py, (float) H, -0.5f), cz = fmaf(pz, (float) D, -0.5f);
	const float fx = floorf(cx), fy = floorf(cy), fz = floorf(cz);
	const float wx = cx - fx, wy = cy - fy, wz = cz - fz;
	const int   ix = (int) fx, iy = (int) fy, iz = (int) fz;
	const int   x0 = i_clamp(ix, 0, W - 1), x1 = i_clamp(ix + 1, 0, W - 1);
	const int   y0 = i_clamp(iy, 0, H - 1), y1 = i_clamp(iy + 1, 0, H - 1);
	const int   z0 = i_clamp(iz, 0, D - 1), z1 = i_clamp(iz + 1, 0, D - 1);
	const float b000 = tex[vidx(x0, y0, z0, W, H)], b100 = tex[vidx(x1, y0, z0, W, H)];
	const float b010 = tex[vidx(x0, y1, z0, W, H)], b110 = tex[vidx(x1, y1, z0, W, H)];
	const float b001 = tex[vidx(x0, y0, z1, W, H)], b101 = tex[vidx(x1, y0, z1, W, H)];
	const float b011 = tex[vidx(x0, y1, z1, W, H)], b111 = tex[vidx(x1, y1, z1, W, H)];
	const float c00 = fmaf(wx, b100 - b000, b000), c10 = fmaf(wx, b110 - b010, b010);
	const float c01 = fmaf(wx, b101 - b001, b001), c11 = fmaf(wx, b111 - b011, b011);
	const float c0 = fmaf(wy, c10 - c00, c00), c1 = fmaf(wy, c11 - c01, c01);
	return fmaf(wz, c1 - c0, c0) * VKV_INV255;
}

/* column-major mat4 * vec4, PIN: r = fma(m3,w, fma(m2,z, fma(m1,y, m0*x))) */
static inline void mat4_mul_vec4(const float *m, const float *v, float *r)
{
	for (int i = 0; i < 4; ++i)
		r[i] = fmaf(m[12 + i], v[3], fmaf(m[8 + i], v[2], fmaf(m[4 + i], v[1], m[i] * v[0])));
}

typedef struct
{
	float    rgba[4];
	uint32_t counts[3]; /* volume samples, distance probes, empty samples */
	float    depth;
	int      fragment; /* 0: the pixel has no fragment (not covered by the clipped box, or discarded by the depth test) */
} PixelOut;

/* One pixel: analytic ray setup (DESIGN.md "Ray generation", replaces the two vertex shaders and the
 * rasteriser) followed by main() of volume_render.frag from line 147 on. */
/* optional per-ray event trace (diagnostics for the scheduling experiments in tools/): 'P' = distance probe that skipped,
 * 'O' = probe that found an occupied cell, 'S' = empty volume sample, 'A' = volume sample with alpha > 0 */
static __thread uint8_t *g_trace;
static __thread int32_t *g_trace_steps; /* optional: the loop index i of every event */
static __thread uint32_t g_trace_cap, g_trace_len;
static inline void trace_event(uint8_t c, int i)
{
	if (g_trace && g_trace_len < g_trace_cap)
	{
		g_trace[g_trace_len] = c;
		if (g_trace_steps)
			g_trace_steps[g_trace_len] = i;
	}
	++g_trace_len;
}

static void march_pixel(const VkvRenderParams *P, const float *alpha_lut, int px, int py, const float *in_depth, PixelOut *out)
{
	memset(out, 0, sizeof(*out)); /* out_color = vec4(0) (frag:120); gl_FragDepth = 0 (frag:140) */

	const VkvTransferFunctionUniform *tfu = &P->transfer_function;
	const int W = (int) P->volume_extent.width, H = (int) P->volume_extent.height, D = (int) P->volume_extent.depth;

	/* ---- ray generation (own definition; volume_render_clipped.vert:50-65 + plane_intersection.vert) ---- */
	const float fx = (float) px + 0.5f, fy = (float) py + 0.5f;
	v3          d;
	d.x = fmaf(fy, P->ray_gen.ddy[0], fmaf(fx, P->ray_gen.ddx[0], P->ray_gen.dir00[0]));
	d.y = fmaf(fy, P->ray_gen.ddy[1], fmaf(fx, P->ray_gen.ddx[1], P->ray_gen.dir00[1]));
	d.z = fmaf(fy, P->ray_gen.ddy[2], fmaf(fx, P->ray_gen.ddx[2], P->ray_gen.dir00[2]));
	{
		const float len = sqrtf(fmaf(d.z, d.z, fmaf(d.y, d.y, d.x * d.x)));
		d.x /= len, d.y /= len, d.z /= len;
	}
	const float *o = P->ray_cast.camera_pos_tex;
	float        t_near = -INFINITY, t_far = INFINITY;
	{
		const float dv[3] = {d.x, d.y, d.z};
		for (int a = 0; a < 3; ++a)
		{
			if (dv[a] == 0.0f)
			{
				if (o[a] < 0.0f || o[a] > 1.0f)
					return; /* parallel to the slab and outside it */
			}
			else
			{
				const float inv = 1.0f / dv[a];
				const float ta = (0.0f - o[a]) * inv, tb = (1.0f - o[a]) * inv;
				t_near = g_max(t_near, g_min(ta, tb));
				t_far  = g_min(t_far, g_max(ta, tb));
			}
		}
	}
	/* clip plane (gl_ClipDistance = dot(plane, pos_world) >= 0 is kept, clipped.vert:56); in texture space the same
	 * half-space is dot(plane_tex, (p,1)) >= 0 (volume_render_subpass.cpp:239) */
	const float *pl = P->ray_cast.plane_tex;
	const float  A  = fmaf(pl[2], o[2], fmaf(pl[1], o[1], pl[0] * o[0])) + pl[3];
	const float  B  = fmaf(pl[2], d.z, fmaf(pl[1], d.y, pl[0] * d.x));
	if (!(B > 0.0f))
		return; /* ray never enters the kept half-space */
	const float t_plane = (0.0f - A) / B;
	const float t0      = g_max(t_near, t_plane);
	if (!(t0 < t_far))
		return; /* pixel not covered by the clipped box */
	v3 ray_entry;
	ray_entry.x = fmaf(t0, d.x, o[0]);
	ray_entry.y = fmaf(t0, d.y, o[1]);
	ray_entry.z = fmaf(t0, d.z, o[2]);
	out->fragment = 1;

	/* ---- DEPTH_ATTACHMENT, frag:122-136: manual z-test of the front face against the scene depth (reverse-Z) ---- */
	float frag_depth = 0.0f, frag_depth_front = 0.0f, position[4] = {0, 0, 0, 0};
	if (P->options.depth_attachment)
	{
		/* position = proj * view * model * vec4(ray_entry - 0.5, 1) (the vertex shaders' position_out, clipped.vert:62) */
		const float pm[4] = {ray_entry.x - 0.5f, ray_entry.y - 0.5f, ray_entry.z - 0.5f, 1.0f};
		float       a4[4], b4[4];
		mat4_mul_vec4(P->camera.model, pm, a4);
		mat4_mul_vec4(P->camera.camera_view, a4, b4);
		mat4_mul_vec4(P->camera.camera_proj, b4, position);
		frag_depth       = *in_depth;
		frag_depth_front = position[2] / position[3];
		if (frag_depth > frag_depth_front)
		{ /* discard: the front face is behind the scene */
			out->fragment = 0;
			return;
		}
		out->depth = frag_depth; /* gl_FragDepth = frag_depth (frag:135) */
	}

	/* ---- frag:147-149 ---- */
	v3 ray_dir;
	{
		const float ex = ray_entry.x - o[0], ey = ray_entry.y - o[1], ez = ray_entry.z - o[2];
		const float len = sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
		ray_dir.x = ex / len, ray_dir.y = ey / len, ray_dir.z = ez / len;
	}
	v3    ray_exit;
	float ray_distance;
	{ /* ray_caster_get_back, frag:71-83 */
		const float ix = 1.0f / ray_dir.x, iy = 1.0f / ray_dir.y, iz = 1.0f / ray_dir.z;
		const float tminx = -ray_entry.x * ix, tminy = -ray_entry.y * iy, tminz = -ray_entry.z * iz;
		const float tmaxx = (1.0f - ray_entry.x) * ix, tmaxy = (1.0f - ray_entry.y) * iy, tmaxz = (1.0f - ray_entry.z) * iz;
		const float t2x = g_max(tminx, tmaxx), t2y = g_max(tminy, tmaxy), t2z = g_max(tminz, tmaxz);
		const float tFar = g_min(g_min(t2x, t2y), t2z);
		ray_exit.x = fmaf(tFar, ray_dir.x, ray_entry.x);
		ray_exit.y = fmaf(tFar, ray_dir.y, ray_entry.y);
		ray_exit.z = fmaf(tFar, ray_dir.z, ray_entry.z);
		const float ex = ray_entry.x - ray_exit.x, ey = ray_entry.y - ray_exit.y, ez = ray_entry.z - ray_exit.z;
		ray_distance = sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
	}
	if (P->options.depth_attachment)
	{ /* frag:152-164: stop the ray where it meets the depth buffer */
		const float clip[4] = {(position[0] * frag_depth) / frag_depth_front, (position[1] * frag_depth) / frag_depth_front,
		                       (position[2] * frag_depth) / frag_depth_front, position[3]};
		float       w4[4], m4[4];
		mat4_mul_vec4(P->camera.camera_view_proj_inv, clip, w4);
		w4[0] /= w4[3], w4[1] /= w4[3], w4[2] /= w4[3], w4[3] /= w4[3];
		mat4_mul_vec4(P->camera.model_inv, w4, m4);
		const float ix = m4[0] + 0.5f, iy = m4[1] + 0.5f, iz = m4[2] + 0.5f;
		const float ex = ray_entry.x - ix, ey = ray_entry.y - iy, ez = ray_entry.z - iz;
		const float dd = sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
		if (dd < ray_distance)
		{
			ray_exit.x = ix, ray_exit.y = iy, ray_exit.z = iz;
			ray_distance = dd;
		}
	}

	/* ---- tests, frag:168-173 ---- */
	if (P->options.test == VKV_TEST_RAY_ENTRY)
	{
		out->rgba[0] = ray_entry.x, out->rgba[1] = ray_entry.y, out->rgba[2] = ray_entry.z, out->rgba[3] = 1.0f;
		return;
	}
	if (P->options.test == VKV_TEST_RAY_EXIT)
	{
		out->rgba[0] = ray_exit.x, out->rgba[1] = ray_exit.y, out->rgba[2] = ray_exit.z, out->rgba[3] = 1.0f;
		return;
	}

	/* ---- number of samples, frag:176-180 ---- */
	const int   dim_max = i_max(i_max(W, H), D);
	const float sf      = tfu->sampling_factor;
	const float nf      = ceilf((float) dim_max * ray_distance * sf);
	/* PIN: rays with fewer than two steps (or a NaN / absurd length) are treated like the grazing-ray early-out
	 * below; the shader would divide by zero there. */
	if (!(nf >= 2.0f && nf <= 16777216.0f))
		return;
	const int n_steps = (int) nf;
	v3        step;
	step.x = (ray_dir.x * ray_distance) / (nf - 1.0f);
	step.y = (ray_dir.y * ray_distance) / (nf - 1.0f);
	step.z = (ray_dir.z * ray_distance) / (nf - 1.0f);

	/* frag:184-187 */
	{
		const float ex = ray_entry.x + step.x, ey = ray_entry.y + step.y, ez = ray_entry.z + step.z;
		if (ex <= 0.0f || ey <= 0.0f || ez <= 0.0f || ex >= 1.0f || ey >= 1.0f || ez >= 1.0f)
			return;
	}

	const int skip_mode = P->options.skipping_type;
	const int mw = (int) P->map_extent.width, mh = (int) P->map_extent.height, md = (int) P->map_extent.depth;
	v3        k = {0, 0, 0}, s_inv = {0, 0, 0};
	if (skip_mode != VKV_SKIP_NONE)
	{ /* frag:191-195 */
		const float *bs = P->ray_cast.block_size;
		k.x = (float) W / bs[0], k.y = (float) H / bs[1], k.z = (float) D / bs[2];
		s_inv.x = 1.0f / ((step.x * (float) W) / bs[0]);
		s_inv.y = 1.0f / ((step.y * (float) H) / bs[1]);
		s_inv.z = 1.0f / ((step.z * (float) D) / bs[2]);
	}
	int i_min_ = 0;
	int ulx = 0, uly = 0, ulz = 0; /* u_last_alpha */

	uint32_t n_vol = 0, n_dist = 0, n_empty = 0;
	const v3 dim_inv = {1.0f / (float) W, 1.0f / (float) H, 1.0f / (float) D};
	const uint8_t *dmap = NULL;
	if (skip_mode == VKV_SKIP_ANISOTROPIC_DISTANCE) /* frag:209 */
		dmap = P->d_distance_maps[(ray_dir.z < 0 ? 1 : 0) + (ray_dir.y < 0 ? 2 : 0) + (ray_dir.x < 0 ? 4 : 0)];
	else if (skip_mode != VKV_SKIP_NONE)
		dmap = P->d_distance_maps[0];

	const int   ert      = P->options.early_ray_termination != 0;
	const int   back     = (int) ceilf(sf); /* frag:253 */
	int         occupied = 1;               /* frag:213 */
	int         i_first_hit = n_steps;
	float       cr = 0, cg = 0, cb = 0, ca = 0;

	for (int i = 0; i < n_steps;)
	{
		const float fi = (float) i;
		const float posx = fmaf(fi, step.x, ray_entry.x), posy = fmaf(fi, step.y, ray_entry.y), posz = fmaf(fi, step.z, ray_entry.z);
		int   uix = 0, uiy = 0, uiz = 0;
		float ux = 0, uy = 0, uz = 0;
		if (skip_mode != VKV_SKIP_NONE)
		{ /* frag:220-221 */
			ux = k.x * posx, uy = k.y * posy, uz = k.z * posz;
			uix = i_clamp((int) ux, 0, mw - 1), uiy = i_clamp((int) uy, 0, mh - 1), uiz = i_clamp((int) uz, 0, md - 1);
		}
		if (skip_mode != VKV_SKIP_NONE && !occupied && (uix != ulx || uiy != uly || uiz != ulz))
		{ /* frag:224-263 */
			++n_dist;
			const uint32_t dist = dmap[vidx(uix, uiy, uiz, mw, mh)];
			if (dist > 0u)
			{
				const float rx = g_clamp((float) uix - ux, -1.0f, 0.0f);
				const float ry = g_clamp((float) uiy - uy, -1.0f, 0.0f);
				const float rz = g_clamp((float) uiz - uz, -1.0f, 0.0f);
				float dx_, dy_, dz_;
				if (skip_mode == VKV_SKIP_BLOCK)
				{ /* frag:239 */
					dx_ = (g_step(0.0f, s_inv.x) + rx) * s_inv.x;
					dy_ = (g_step(0.0f, s_inv.y) + ry) * s_inv.y;
					dz_ = (g_step(0.0f, s_inv.z) + rz) * s_inv.z;
				}
				else
				{ /* frag:242 */
					const float fd = (float) dist;
					dx_ = ((g_step(0.0f, -s_inv.x) + g_sign(s_inv.x) * fd) + rx) * s_inv.x;
					dy_ = ((g_step(0.0f, -s_inv.y) + g_sign(s_inv.y) * fd) + ry) * s_inv.y;
					dz_ = ((g_step(0.0f, -s_inv.z) + g_sign(s_inv.z) * fd) + rz) * s_inv.z;
				}
				/* PIN: 0*inf on an axis-parallel component is +inf (that axis never limits the skip); the step count is
				 * capped at 2^30 before the float->int conversion. */
				if (dx_ != dx_) dx_ = INFINITY;
				if (dy_ != dy_) dy_ = INFINITY;
				if (dz_ != dz_) dz_ = INFINITY;
				float m = g_min(g_min(dx_, dy_), dz_);
				m       = (m < 1073741824.0f) ? m : 1073741824.0f;
				trace_event('P', i);
				i += i_max(1, (int) ceilf(m)); /* frag:244-247 */
			}
			else
			{ /* frag:253-261 */
				occupied = 1;
				ulx = uix, uly = uiy, ulz = uiz;
				trace_event('O', i);
				i = i_max(i - back, i_min_);
			}
		}
		else
		{ /* frag:266-310 */
			++n_vol;
			const float intensity = sample_linear(P->d_volume, W, H, D, posx, posy, posz);
			float       gradient  = 1.0f; /* frag:100-102 */
			if (tfu->use_gradient)
			{
				if (P->use_precomputed_gradient)
					gradient = sample_linear(P->d_gradient, W, H, D, posx, posy, posz); /* frag:89 */
				else
				{ /* frag:92-97, term order k.xyy, k.yyx, k.yxy, k.xxx */
					const float t1 = sample_linear(P->d_volume, W, H, D, posx + dim_inv.x, posy - dim_inv.y, posz - dim_inv.z);
					const float t2 = sample_linear(P->d_volume, W, H, D, posx - dim_inv.x, posy - dim_inv.y, posz + dim_inv.z);
					const float t3 = sample_linear(P->d_volume, W, H, D, posx - dim_inv.x, posy + dim_inv.y, posz - dim_inv.z);
					const float t4 = sample_linear(P->d_volume, W, H, D, posx + dim_inv.x, posy + dim_inv.y, posz + dim_inv.z);
					const float gx = (((t1 - t2) - t3) + t4) * 0.25f;
					const float gy = (((-t1 - t2) + t3) + t4) * 0.25f;
					const float gz = (((-t1 + t2) - t3) + t4) * 0.25f;
					const float len = sqrtf((gx * gx + gy * gy) + gz * gz);
					gradient = g_clamp(len * tfu->grad_magnitude_modifier, 0.0f, 1.0f);
				}
			}
			/* get_color, transfer_function.glsl:35-38: NEAREST texel of the RGBA8 LUT */
			const uint8_t *texel = P->d_transfer_function + ((size_t) tf_texel(gradient) * 256 + (size_t) tf_texel(intensity)) * 4;
			occupied = texel[3] > 0; /* frag:276 */
			trace_event(occupied ? 'A' : 'S', i);
			if (occupied)
			{
				if (skip_mode != VKV_SKIP_NONE)
					ulx = uix, uly = uiy, ulz = uiz;
				/* frag:283-284; alpha_lut[a] = clamp(alpha_factor * (1 - pow(1 - a/255, 1/sf)), 0, 1) */
				const float a  = alpha_lut[texel[3]];
				const float r_ = unorm8(texel[0]) * a, g_ = unorm8(texel[1]) * a, b_ = unorm8(texel[2]) * a;
				/* frag:287, PIN: one fma per channel */
				const float om = 1.0f - ca;
				cr = fmaf(om, r_, cr), cg = fmaf(om, g_, cg), cb = fmaf(om, b_, cb), ca = fmaf(om, a, ca);
				if (a > 0.0f)
					i_first_hit = i; /* frag:289-291 */
				if (ca > 0.99f && ert)
				{ /* frag:293-299 */
					ca = 1.0f;
					break;
				}
			}
			else
				++n_empty;
			++i;
			i_min_ = i;
		}
	}

	out->counts[0] = n_vol, out->counts[1] = n_dist, out->counts[2] = n_empty;

	/* frag:315-321 */
	if (ca > 0.0f && i_first_hit < n_steps)
	{
		const float fi   = (float) i_first_hit;
		const float p[4] = {fmaf(fi, step.x, ray_entry.x) - 0.5f, fmaf(fi, step.y, ray_entry.y) - 0.5f, fmaf(fi, step.z, ray_entry.z) - 0.5f, 1.0f};
		float       a[4], b[4], c[4];
		mat4_mul_vec4(P->camera.model, p, a);
		mat4_mul_vec4(P->camera.camera_view, a, b);
		mat4_mul_vec4(P->camera.camera_proj, b, c);
		out->depth = c[2] / c[3];
	}

	if (P->options.test == VKV_TEST_NUM_TEXTURE_SAMPLES)
	{ /* frag:324-334 */
		const uint32_t n_steps_max = (uint32_t) (ceilf((float) dim_max * sqrtf(3.0f)) * sf);
		const float    v           = (float) (n_vol + n_dist) / (float) n_steps_max;
		out->rgba[0] = out->rgba[1] = out->rgba[2] = v;
		out->rgba[3] = 1.0f;
	}
	else
	{
		out->rgba[0] = cr, out->rgba[1] = cg, out->rgba[2] = cb, out->rgba[3] = ca;
	}
}

/* diagnostics: event sequence of one pixel's ray; returns its length (may exceed cap) */
static void build_alpha_lut(const VkvTransferFunctionUniform *tf, float *lut);
uint32_t vkvo_trace_ray_steps(const VkvRenderParams *P, int px, int py, uint8_t *events, int32_t *steps, uint32_t cap)
{
	float lut[256];
	build_alpha_lut(&P->transfer_function, lut);
	PixelOut po;
	g_trace = events, g_trace_steps = steps, g_trace_cap = cap, g_trace_len = 0;
	float far_depth = 0.0f;
	march_pixel(P, lut, px, py, P->d_in_depth ? P->d_in_depth + ((size_t) py * P->image_width + px) : &far_depth, &po);
	g_trace = NULL, g_trace_steps = NULL;
	return g_trace_len;
}

uint32_t vkvo_trace_ray(const VkvRenderParams *P, int px, int py, uint8_t *events, uint32_t cap)
{
	return vkvo_trace_ray_steps(P, px, py, events, NULL, cap);
}

/* opacity-correction table keyed by the TF alpha byte (frag:283); shared definition with the product:
 * lut[a] = clamp(voxel_alpha_factor * (1 - powf(1 - a/255, 1/sampling_factor)), 0, 1). */
static void build_alpha_lut(const VkvTransferFunctionUniform *tf, float *lut)
{
	const float sf_inv = 1.0f / tf->sampling_factor;
	for (int a = 0; a < 256; ++a)
		lut[a] = g_clamp(tf->voxel_alpha_factor * (1.0f - powf(1.0f - unorm8((uint8_t) a), sf_inv)), 0.0f, 1.0f);
}

static inline uint8_t quantise_rgba8(float c) { return (uint8_t) rintf(g_clamp(c, 0.0f, 1.0f) * 255.0f); }

/* One render call = one job: the frame's work items (CHUNK_ROWS pixel rows of one tile) are claimed from an atomic counter by the
 * threads of a persistent pool, so that a frame whose cost is concentrated in a few tiles (empty-space skipping: 60 % of a bench frame
 * never enters the volume) keeps every core busy to the end, and a call costs no thread creation (round 3 created n - 1 threads per
 * call and dealt rows statically: 256 cores delivered 23 x the one-thread rate). */
#define CHUNK_ROWS 4u

typedef struct
{
	const VkvRenderParams *P;
	const float *          lut;
	uint32_t               stride;
	uint64_t               n_items;        /* tile_count * ceil(tile_height / CHUNK_ROWS) */
	uint64_t               grab;           /* items per claim */
	/* the two words the workers write, on a cache line of their own (the fields above are read per pixel by every worker) */
	__attribute__((aligned(64))) uint64_t next;        /* next unclaimed item (atomic) */
	uint64_t                              rays;        /* rays marched (atomic, added once per worker) */
	char                                  pad[48];
} RenderJob;

static void render_items(RenderJob *job)
{
	const VkvRenderParams *P   = job->P;
	const uint32_t         tw = P->tiles.tile_width, th = P->tiles.tile_height;
	/* the schedule's tile rectangle (VkvTileSchedule.rect; all zero = the whole image): tiles are numbered row-major inside it */
	const int              whole   = P->tiles.rect.w == 0 || P->tiles.rect.h == 0;
	const uint32_t         tiles_x = whole ? (P->image_width + tw - 1) / tw : P->tiles.rect.w;
	const uint32_t         org_x = whole ? 0u : P->tiles.rect.x0 * tw, org_y = whole ? 0u : P->tiles.rect.y0 * th;
	const uint32_t         chunks  = (th + CHUNK_ROWS - 1) / CHUNK_ROWS;
	const uint32_t         stride  = job->stride;
	const float *          lut     = job->lut;
	const uint64_t         n_items = job->n_items;
	uint64_t               rays    = 0;
	/* items are claimed `grab` at a time: ~32 claims per worker and frame (one contended atomic per item cost 256 threads more than the
	 * marching itself: 260 000 atomics on one cache line per 1920x1080 frame) */
	const uint64_t grab = job->grab;
	uint64_t       item = 0, item_end = 0;
	for (;;)
	{
		if (item == item_end)
		{
			item = __atomic_fetch_add(&job->next, grab, __ATOMIC_RELAXED);
			if (item >= n_items)
				break;
			item_end = item + grab < n_items ? item + grab : n_items;
		}
		const uint32_t k  = (uint32_t) (item / chunks), row0 = (uint32_t) (item % chunks) * CHUNK_ROWS;
		const uint32_t t  = P->tiles.tile_first + k * P->tiles.tile_stride;
		const uint32_t x0 = org_x + (t % tiles_x) * tw, y0 = org_y + (t / tiles_x) * th;
		for (uint32_t ly = row0; ly < th && ly < row0 + CHUNK_ROWS; ++ly)
		{
			const uint32_t y = y0 + ly;
			if (y >= P->image_height || (y % stride) != 0)
				continue;
			for (uint32_t lx = 0; lx < tw; ++lx)
			{
				const uint32_t x = x0 + lx;
				if (x >= P->image_width || (x % stride) != 0)
					continue;
				PixelOut     po;
				const size_t o = P->tiles.compact ? ((size_t) k * th + ly) * tw + lx : (size_t) y * P->image_width + x;
				const float  far_depth = 0.0f;
				march_pixel(P, lut, (int) x, (int) y, P->d_in_depth ? P->d_in_depth + o : &far_depth, &po);
				++rays;
				if (!po.fragment)
				{ /* no fragment: an existing target stays as it is, a fresh one holds the clear values */
					if (P->blend_over_target)
					{
						if (P->d_out_counts)
							memcpy(P->d_out_counts + o * 3, po.counts, sizeof(po.counts));
						continue;
					}
					if (P->options.depth_attachment && P->d_in_depth)
						po.depth = P->d_in_depth[o];
				}
				else if (P->blend_over_target)
				{ /* blend state of the subpass (src/volume_render_subpass.cpp:176-190) */
					const float om = 1.0f - po.rgba[3];
					if (P->d_out_color)
					{
						float *dst = P->d_out_color + o * 4;
						for (int c = 0; c < 3; ++c)
							dst[c] = fmaf(om, dst[c], po.rgba[c]);
						dst[3] = po.rgba[3] * om;
					}
					if (P->d_out_rgba8)
					{
						uint8_t *dst = P->d_out_rgba8 + o * 4;
						for (int c = 0; c < 3; ++c)
							dst[c] = quantise_rgba8(fmaf(om, unorm8(dst[c]), po.rgba[c]));
						dst[3] = quantise_rgba8(po.rgba[3] * om);
					}
					if (P->d_out_counts)
						memcpy(P->d_out_counts + o * 3, po.counts, sizeof(po.counts));
					if (P->d_out_depth)
						P->d_out_depth[o] = po.depth;
					continue;
				}
				if (P->d_out_color)
					memcpy(P->d_out_color + o * 4, po.rgba, sizeof(po.rgba));
				if (P->d_out_rgba8)
					for (int c = 0; c < 4; ++c)
						P->d_out_rgba8[o * 4 + c] = quantise_rgba8(po.rgba[c]);
				if (P->d_out_counts)
					memcpy(P->d_out_counts + o * 3, po.counts, sizeof(po.counts));
				if (P->d_out_depth)
					P->d_out_depth[o] = po.depth;
			}
		}
		++item;
	}
	__atomic_fetch_add(&job->rays, rays, __ATOMIC_RELAXED);
}

/* ---- the pool: threads are created on first use (and when a call asks for more than exist), then sleep on a condition variable
 * between jobs.  One job at a time (vkvo_render holds the job mutex); the calling thread works too. */
static struct
{
	pthread_mutex_t job_mutex;          /* one vkvo_render at a time */
	pthread_mutex_t m;
	pthread_cond_t  wake, done;
	RenderJob *     job;                /* the current job, NULL between jobs */
	uint64_t        generation;         /* bumped per job */
	int             participants;       /* pool threads with id < participants take part in the current job */
	int             running;            /* pool threads still inside the current job */
	int             n_threads;          /* pool threads created so far */
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0, 0, 0, 0};

/* a forked child inherits the pool's bookkeeping but none of its threads: start over (a parent that forks while a render is running is
 * on its own, as with any lock held across fork) */
static void pool_after_fork_in_child(void)
{
	pthread_mutex_init(&g_pool.job_mutex, NULL);
	pthread_mutex_init(&g_pool.m, NULL);
	pthread_cond_init(&g_pool.wake, NULL);
	pthread_cond_init(&g_pool.done, NULL);
	g_pool.job = NULL, g_pool.generation = 0, g_pool.participants = 0, g_pool.running = 0, g_pool.n_threads = 0;
}

static pthread_once_t g_pool_once = PTHREAD_ONCE_INIT;
static void           pool_init_once(void) { pthread_atfork(NULL, NULL, pool_after_fork_in_child); }

static void *pool_thread(void *arg)
{
	const int id   = (int) (intptr_t) arg;
	uint64_t  seen = 0;
	pthread_mutex_lock(&g_pool.m);
	for (;;)
	{
		while (g_pool.generation == seen)
			pthread_cond_wait(&g_pool.wake, &g_pool.m);
		seen           = g_pool.generation;
		RenderJob *job = id < g_pool.participants ? g_pool.job : NULL;
		if (!job)
			continue;
		pthread_mutex_unlock(&g_pool.m);
		render_items(job);
		pthread_mutex_lock(&g_pool.m);
		if (--g_pool.running == 0)
			pthread_cond_signal(&g_pool.done);
	}
	return NULL;
}

uint64_t vkvo_render(const VkvRenderParams *P_in, int n_threads, uint32_t pixel_stride)
{
	/* VkvTileSchedule.fill_outside: the launch completes the frame - its result is DEFINED as the whole-image schedule's, and that is what the
	 * oracle renders (every pixel through the frag, whatever rectangle the caller derived) */
	VkvRenderParams        whole;
	const VkvRenderParams *P = P_in;
	if (P_in->tiles.fill_outside && P_in->tiles.rect.w && P_in->tiles.rect.h && !P_in->tiles.compact)
	{
		whole = *P_in;
		memset(&whole.tiles.rect, 0, sizeof(whole.tiles.rect));
		whole.tiles.fill_outside = 0, whole.tiles.tile_first = 0, whole.tiles.tile_stride = 1;
		whole.tiles.tile_count   = ((P_in->image_width + P_in->tiles.tile_width - 1) / P_in->tiles.tile_width) *
		                         ((P_in->image_height + P_in->tiles.tile_height - 1) / P_in->tiles.tile_height);
		P = &whole;
	}
	float lut[256];
	build_alpha_lut(&P->transfer_function, lut);
	if (n_threads < 1)
		n_threads = 1;
	if (pixel_stride < 1)
		pixel_stride = 1;
	RenderJob job;
	job.P = P, job.lut = lut, job.stride = pixel_stride, job.next = 0, job.rays = 0;
	job.n_items = (uint64_t) P->tiles.tile_count * ((P->tiles.tile_height + CHUNK_ROWS - 1) / CHUNK_ROWS);
	job.grab = job.n_items / ((uint64_t) n_threads * 32u);
	if (job.grab < 1)
		job.grab = 1;
	if (n_threads == 1 || job.n_items < 2)
	{
		render_items(&job);
		return job.rays;
	}
	pthread_once(&g_pool_once, pool_init_once);
	pthread_mutex_lock(&g_pool.job_mutex);
	pthread_mutex_lock(&g_pool.m);
	while (g_pool.n_threads < n_threads - 1)
	{        /* (the caller is the n-th worker) a thread that cannot be created just leaves the job to the others */
		pthread_t      th;
		pthread_attr_t at;
		pthread_attr_init(&at);
		pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
		const int rc = pthread_create(&th, &at, pool_thread, (void *) (intptr_t) g_pool.n_threads);
		pthread_attr_destroy(&at);
		if (rc != 0)
			break;
		++g_pool.n_threads;
	}
	const int helpers  = g_pool.n_threads < n_threads - 1 ? g_pool.n_threads : n_threads - 1;
	g_pool.job          = &job;
	g_pool.participants = helpers;
	g_pool.running      = helpers;
	++g_pool.generation;
	pthread_cond_broadcast(&g_pool.wake);
	pthread_mutex_unlock(&g_pool.m);
	render_items(&job);
	pthread_mutex_lock(&g_pool.m);
	while (g_pool.running > 0)
		pthread_cond_wait(&g_pool.done, &g_pool.m);
	g_pool.job = NULL;
	pthread_mutex_unlock(&g_pool.m);
	pthread_mutex_unlock(&g_pool.job_mutex);
	return job.rays;
}

/* ------------------------------------------------------------------------------------------- */
/* Uniform maths — src/volume_render_subpass.cpp:221-249 in double precision                   */
/* ------------------------------------------------------------------------------------------- */

static void m4_mul_d(const double *a, const double *b, double *r)
{ /* column-major r = a*b */
	double t[16];
	for (int c = 0; c < 4; ++c)
		for (int rr = 0; rr < 4; ++rr)
		{
			double s = 0;
			for (int kk = 0; kk < 4; ++kk)
				s += a[kk * 4 + rr] * b[c * 4 + kk];
			t[c * 4 + rr] = s;
		}
	memcpy(r, t, sizeof(t));
}

static int m4_inv_d(const double *m, double *out)
{ /* Gauss-Jordan with partial pivoting on the column-major matrix */
	double a[4][8];
	for (int r = 0; r < 4; ++r)
		for (int c = 0; c < 4; ++c)
		{
			a[r][c]     = m[c * 4 + r];
			a[r][c + 4] = (r == c) ? 1.0 : 0.0;
		}
	for (int c = 0; c < 4; ++c)
	{
		int piv = c;
		for (int r = c + 1; r < 4; ++r)
			if (fabs(a[r][c]) > fabs(a[piv][c]))
				piv = r;
		if (a[piv][c] == 0.0)
			return -1;
		if (piv != c)
			for (int j = 0; j < 8; ++j)
			{
				double t  = a[c][j];
				a[c][j]   = a[piv][j];
				a[piv][j] = t;
			}
		const double inv = 1.0 / a[c][c];
		for (int j = 0; j < 8; ++j)
			a[c][j] *= inv;
		for (int r = 0; r < 4; ++r)
			if (r != c)
			{
				const double f = a[r][c];
				for (int j = 0; j < 8; ++j)
					a[r][j] -= f * a[c][j];
			}
	}
	for (int r = 0; r < 4; ++r)
		for (int c = 0; c < 4; ++c)
			out[c * 4 + r] = a[r][c + 4];
	return 0;
}

static void m4_mul_v_d(const double *m, const double *v, double *r)
{
	for (int i = 0; i < 4; ++i)
		r[i] = m[i] * v[0] + m[4 + i] * v[1] + m[8 + i] * v[2] + m[12 + i] * v[3];
}

void vkvo_build_uniforms(const float *view, const float *proj, const float *node_transform, const float *image_transform,
                         float clip_distance, uint32_t image_width, uint32_t image_height, VkvExtent3D ve, VkvExtent3D me,
                         VkvCameraUniform *cam, VkvRayCastUniform *rc, VkvRayGen *rg)
{
	double V[16], Pm[16], N[16], I[16], M[16], Minv[16], PV[16], PVinv[16], Vinv[16];
	for (int i = 0; i < 16; ++i)
		V[i] = view[i], Pm[i] = proj[i], N[i] = node_transform[i], I[i] = image_transform[i];
	m4_mul_d(N, I, M); /* :227 */
	m4_inv_d(M, Minv);
	m4_mul_d(Pm, V, PV);
	m4_inv_d(PV, PVinv);
	m4_inv_d(V, Vinv);
	for (int i = 0; i < 16; ++i)
	{
		cam->camera_view[i]          = (float) V[i];
		cam->camera_proj[i]          = (float) Pm[i];
		cam->camera_view_proj_inv[i] = (float) PVinv[i];
		cam->model[i]                = (float) M[i];
		cam->model_inv[i]            = (float) Minv[i];
	}
	/* global_to_tex = translate(0.5) * model_inv (:231-232) */
	double T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.5, 0.5, 0.5, 1};
	double G2T[16];
	m4_mul_d(T, Minv, G2T);
	const double cam_pos[4] = {Vinv[12], Vinv[13], Vinv[14], 1.0}; /* viewInv[3] (:234) */
	double       cam_tex[4];
	m4_mul_v_d(G2T, cam_pos, cam_tex); /* :235-236 */
	const double fwd[4] = {0, 0, -1, 0};
	double       cam_dir[4];
	m4_mul_v_d(Vinv, fwd, cam_dir); /* :237 */
	const double plane[4] = {cam_dir[0], cam_dir[1], cam_dir[2],
	                         -(double) clip_distance - (cam_pos[0] * cam_dir[0] + cam_pos[1] * cam_dir[1] + cam_pos[2] * cam_dir[2])}; /* :238 */
	/* plane_tex = inverseTranspose(global_to_tex) * plane (:239); inverse(global_to_tex) = M * translate(-0.5) */
	double Tm[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -0.5, -0.5, -0.5, 1};
	double T2G[16];
	m4_mul_d(M, Tm, T2G);
	double plane_tex[4];
	for (int i = 0; i < 4; ++i) /* transpose(T2G) * plane */
		plane_tex[i] = T2G[i * 4 + 0] * plane[0] + T2G[i * 4 + 1] * plane[1] + T2G[i * 4 + 2] * plane[2] + T2G[i * 4 + 3] * plane[3];
	for (int i = 0; i < 4; ++i)
	{
		rc->plane[i]          = (float) plane[i];
		rc->plane_tex[i]      = (float) plane_tex[i];
		rc->camera_pos_tex[i] = (float) cam_tex[i];
	}
	rc->front_index   = (rc->plane_tex[0] < 0 ? 1 : 0) + (rc->plane_tex[1] < 0 ? 2 : 0) + (rc->plane_tex[2] < 0 ? 4 : 0); /* :240-242 */
	rc->block_size[0] = (float) ((ve.width + me.width - 1) / me.width);                                                     /* :245-249 */
	rc->block_size[1] = (float) ((ve.height + me.height - 1) / me.height);
	rc->block_size[2] = (float) ((ve.depth + me.depth - 1) / me.depth);
	rc->block_size[3] = 0.0f;

	/* ray generator: texture-space direction through three pixel-space points, from two depths of the unprojection */
	double dirs[3][3];
	const double pts[3][2] = {{0, 0}, {1, 0}, {0, 1}};
	for (int p = 0; p < 3; ++p)
	{
		const double ndc_x = 2.0 * pts[p][0] / (double) image_width - 1.0;
		const double ndc_y = 2.0 * pts[p][1] / (double) image_height - 1.0;
		const double c1[4] = {ndc_x, ndc_y, 1.0, 1.0}, c2[4] = {ndc_x, ndc_y, 0.25, 1.0};
		double       w1[4], w2[4], t1[4], t2[4];
		m4_mul_v_d(PVinv, c1, w1);
		m4_mul_v_d(PVinv, c2, w2);
		const double iw1 = 1.0 / w1[3], iw2 = 1.0 / w2[3];
		w1[0] *= iw1, w1[1] *= iw1, w1[2] *= iw1, w1[3] = 1.0;
		w2[0] *= iw2, w2[1] *= iw2, w2[2] *= iw2, w2[3] = 1.0;
		m4_mul_v_d(G2T, w1, t1);
		m4_mul_v_d(G2T, w2, t2);
		const double dx = t2[0] - t1[0], dy = t2[1] - t1[1], dz = t2[2] - t1[2];
		/* plane_tex.xyz is the texture-space covector of "distance along the view direction"; scaling every direction to
		 * unit distance along it puts the three of them on one image plane, hence affine in pixel coordinates */
		const double along = plane_tex[0] * dx + plane_tex[1] * dy + plane_tex[2] * dz;
		dirs[p][0] = dx / along, dirs[p][1] = dy / along, dirs[p][2] = dz / along;
	}
	for (int i = 0; i < 3; ++i)
	{
		rg->dir00[i] = (float) dirs[0][i];
		rg->ddx[i]   = (float) (dirs[1][i] - dirs[0][i]);
		rg->ddy[i]   = (float) (dirs[2][i] - dirs[0][i]);
	}
	rg->dir00[3] = rg->ddx[3] = rg->ddy[3] = 0.0f;
}

/* ------------------------------------------------------------------------------------------- */
/* Synthetic volumes (definition in DESIGN.md "Synthetic inputs")               */
/* ------------------------------------------------------------------------------------------- */

static uint64_t splitmix64(uint64_t *s)
{
	uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
	z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static float u01(uint64_t *s) { return (float) (splitmix64(s) >> 40) * (1.0f / 16777216.0f); }

#define SYNTH_SHELLS 40
typedef struct
{
	float cx, cy, cz, irx, iry, irz, slope, amp, lo2, hi2;
} Shell;

/* kind = 1 | shells << 8 | thickness << 16 | noise << 28 (include/vkvolume_amd.h, vkv_synth_volume): the first `shells` (0 = all 40) of the seed's
 * shells, their thickness scaled by thickness / 256 (12 bits, 0 = 1), hash noise 0 .. noise (4 bits, 0 = the default 0 .. 20) - the knobs
 * tools/benchmark_sweep.py tunes the occupied share of its scenes with */
static void synth_shells(VkvExtent3D e, uint32_t seed, uint32_t kind, Shell *sh)
{
	uint64_t    s  = (0x5EEDull << 32) | (uint64_t) seed;
	const float Wf = (float) e.width, Hf = (float) e.height, Df = (float) e.depth;
	const float dm = fmaxf(fmaxf(Wf, Hf), Df);
	const uint32_t tq = (kind >> 16) & 0xfffu;
	const float th = (0.001f * dm + 1.0f) * (tq ? (float) tq * (1.0f / 256.0f) : 1.0f);
	for (int k = 0; k < SYNTH_SHELLS; ++k)
	{
		sh[k].cx = (0.15f + 0.70f * u01(&s)) * Wf;
		sh[k].cy = (0.15f + 0.70f * u01(&s)) * Hf;
		sh[k].cz = (0.15f + 0.70f * u01(&s)) * Df;
		const float r  = (0.05f + 0.13f * u01(&s)) * dm;
		const float rx = r * (0.7f + 0.6f * u01(&s));
		const float ry = r * (0.7f + 0.6f * u01(&s));
		const float rz = r * (0.7f + 0.6f * u01(&s));
		sh[k].irx = 1.0f / rx, sh[k].iry = 1.0f / ry, sh[k].irz = 1.0f / rz;
		sh[k].slope = fminf(fminf(rx, ry), rz) / th;
		sh[k].amp   = 110.0f + 145.0f * u01(&s);
		const float w  = 1.0f / sh[k].slope + 0.001f;
		const float lo = 1.0f - w, hi = 1.0f + w;
		sh[k].lo2 = lo > 0.0f ? lo * lo : 0.0f;
		sh[k].hi2 = hi * hi;
	}
}

static inline uint32_t synth_hash(uint32_t seed, uint32_t x, uint32_t y, uint32_t z)
{
	uint32_t h = seed ^ (x * 0x8da6b343u) ^ (y * 0xd8163841u) ^ (z * 0xcb1ab31fu);
	h ^= h >> 16;
	h *= 0x7feb352du;
	h ^= h >> 15;
	h *= 0x846ca68bu;
	h ^= h >> 16;
	return h;
}

void vkvo_synth_volume(uint8_t *vol, VkvExtent3D e, uint32_t kind, uint32_t seed)
{
	const int W = (int) e.width, H = (int) e.height, D = (int) e.depth;
	if ((kind & 255u) == 0)
	{ /* C1 soft sphere: v = round(255 * clamp((R0 - r) / (R0 - R1), 0, 1)), R0 = 0.375 dim, R1 = 0.25 dim */
		const float dm = (float) i_max(i_max(W, H), D);
		const float R0 = 0.375f * dm, R1 = 0.25f * dm;
		const float cx = ((float) W - 1.0f) * 0.5f, cy = ((float) H - 1.0f) * 0.5f, cz = ((float) D - 1.0f) * 0.5f;
		for (int z = 0; z < D; ++z)
			for (int y = 0; y < H; ++y)
				for (int x = 0; x < W; ++x)
				{
					const float dx = (float) x - cx, dy = (float) y - cy, dz = (float) z - cz;
					const float r  = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
					const float t  = g_clamp((R0 - r) / (R0 - R1), 0.0f, 1.0f);
					vol[vidx(x, y, z, W, H)] = (uint8_t) rintf(255.0f * t);
				}
		return;
	}
	Shell sh[SYNTH_SHELLS];
	synth_shells(e, seed, kind, sh);
	const uint32_t nq = (kind >> 8) & 255u;
	const int      n_shells = nq && nq < SYNTH_SHELLS ? (int) nq : SYNTH_SHELLS;
	const uint32_t noise_mod = (kind >> 28) ? (kind >> 28) + 1u : 21u;
	for (int z = 0; z < D; ++z)
		for (int y = 0; y < H; ++y)
			for (int x = 0; x < W; ++x)
			{
				float best = 0.0f;
				for (int k = 0; k < n_shells; ++k)
				{
					const float dx = ((float) x - sh[k].cx) * sh[k].irx;
					const float dy = ((float) y - sh[k].cy) * sh[k].iry;
					const float dz = ((float) z - sh[k].cz) * sh[k].irz;
					const float q2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
					if (q2 > sh[k].lo2 && q2 < sh[k].hi2)
					{
						const float q   = sqrtf(q2);
						const float val = sh[k].amp * (1.0f - fabsf(q - 1.0f) * sh[k].slope);
						if (val > best)
							best = val;
					}
				}
				const uint32_t noise = synth_hash(seed, (uint32_t) x, (uint32_t) y, (uint32_t) z) % noise_mod;
				const uint32_t v     = (uint32_t) best + noise;
				vol[vidx(x, y, z, W, H)] = (uint8_t) (v > 255u ? 255u : v);
			}
}

/* ------------------------------------------------------------------------------------------- */
/* Loader — src/load_volume.cpp                                                                */
/* ------------------------------------------------------------------------------------------- */

/* glm::rotate(angle, axis) * glm::scale(size), column-major (load_volume.cpp:82-83) */
static void rotate_scale(float angle_deg, const float *axis, const float *size, float *m)
{
	const float a = angle_deg * 0.01745329251994329576923690768489f; /* glm::radians */
	const float c = cosf(a), s = sinf(a);
	float       n[3] = {axis[0], axis[1], axis[2]};
	const float len  = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
	if (len > 0)
		n[0] /= len, n[1] /= len, n[2] /= len;
	const float t[3] = {(1 - c) * n[0], (1 - c) * n[1], (1 - c) * n[2]};
	float       R[16] = {0};
	R[0] = c + t[0] * n[0], R[1] = t[0] * n[1] + s * n[2], R[2] = t[0] * n[2] - s * n[1];
	R[4] = t[1] * n[0] - s * n[2], R[5] = c + t[1] * n[1], R[6] = t[1] * n[2] + s * n[0];
	R[8] = t[2] * n[0] + s * n[1], R[9] = t[2] * n[1] - s * n[0], R[10] = c + t[2] * n[2];
	R[15] = 1;
	for (int col = 0; col < 3; ++col)
		for (int r = 0; r < 4; ++r)
			m[col * 4 + r] = R[col * 4 + r] * size[col];
	m[12] = m[13] = m[14] = 0, m[15] = 1;
}

int vkvo_load_header(const char *fn, VkvoHeader *h)
{
	FILE *f = fopen(fn, "r");
	if (!f)
		return VKV_E_IO; /* load_volume.cpp:36-39 */
	char  line[512];
	float axis_angle[4] = {0, 0, 0, 0};
	memset(h, 0, sizeof(*h));
	if (fgets(line, sizeof line, f)) sscanf(line, "%u %u %u", &h->extent.width, &h->extent.height, &h->extent.depth);
	if (fgets(line, sizeof line, f)) sscanf(line, "%f %f %f", &h->voxel_size[0], &h->voxel_size[1], &h->voxel_size[2]);
	if (fgets(line, sizeof line, f)) sscanf(line, "%f %f", &h->normalisation_range[0], &h->normalisation_range[1]);
	if (fgets(line, sizeof line, f)) sscanf(line, "%15s %15s", h->type, h->endianness);
	if (fgets(line, sizeof line, f)) sscanf(line, "%f %f %f %f", &axis_angle[0], &axis_angle[1], &axis_angle[2], &axis_angle[3]);
	fclose(f);
	const float size[3] = {h->voxel_size[0] * (float) h->extent.width, h->voxel_size[1] * (float) h->extent.height,
	                       h->voxel_size[2] * (float) h->extent.depth};
	rotate_scale(axis_angle[3], axis_angle, size, h->image_transform);
	return 0;
}

int vkvo_load_data(const char *fn, const VkvoHeader *h, uint8_t *out)
{
	int bytes, is_signed;
	if (!strcmp(h->type, "uint8_t")) bytes = 1, is_signed = 0;
	else if (!strcmp(h->type, "int8_t")) bytes = 1, is_signed = 1;
	else if (!strcmp(h->type, "uint16_t")) bytes = 2, is_signed = 0;
	else if (!strcmp(h->type, "int16_t")) bytes = 2, is_signed = 1;
	else return VKV_E_INVALID_ARGUMENT; /* load_volume.cpp:106-109 */
	const size_t n = (size_t) h->extent.width * h->extent.height * h->extent.depth;
	FILE *       f = fopen(fn, "rb");
	if (!f)
		return VKV_E_IO;
	fseek(f, 0, SEEK_END);
	const long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	if ((size_t) sz != n * (size_t) bytes)
	{ /* load_volume.cpp:128-131 */
		fclose(f);
		return VKV_E_IO;
	}
	uint8_t *raw = (uint8_t *) malloc(n * (size_t) bytes);
	if (fread(raw, (size_t) bytes, n, f) != n)
	{
		free(raw);
		fclose(f);
		return VKV_E_IO;
	}
	fclose(f);
	const int   big = !strcmp(h->endianness, "big"); /* :152 — anything else is little */
	const float mn = h->normalisation_range[0], mx = h->normalisation_range[1];
	for (size_t i = 0; i < n; ++i)
	{
		float v;
		if (bytes == 1)
			v = is_signed ? (float) (int8_t) raw[i] : (float) raw[i];
		else
		{
			const uint16_t u = big ? (uint16_t) ((raw[2 * i] << 8) | raw[2 * i + 1]) : (uint16_t) ((raw[2 * i + 1] << 8) | raw[2 * i]);
			v                = is_signed ? (float) (int16_t) u : (float) u;
		}
		/* :165-169: (uint8) (255 * max(0, min(1, (v - min) / (max - min)))), truncating */
		const float t = fmaxf(0.0f, fminf(1.0f, (v - mn) / (mx - mn)));
		out[i]        = (uint8_t) (255 * t);
	}
	free(raw);
	return 0;
}
