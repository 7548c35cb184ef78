// raymarch.hip — the ray-march integrator with block / Chebyshev / anisotropic-Chebyshev empty-space skipping
// and early ray termination, for gfx950.
//
// Replaces VolumeRenderSubpass::draw (src/volume_render_subpass.cpp:159-294) and the shaders it binds:
// shaders/volume_render.frag (integrator), shaders/transfer_function.glsl (get_color) and the two vertex
// shaders (ray entry; here an analytic per-pixel box / clip-plane intersection, there is no rasteriser on CDNA).
//
// Mapping: one lane per ray, one wave per 8x8 pixel tile (rays of a wave stay spatially coherent so the byte
// gathers of a wave fall into a few cache lines), one 256-thread workgroup per 16x16 pixels, workgroups remapped
// so each XCD (own L2) owns a contiguous run of screen tiles.  The frag's compile-time variants
// (volume_render_subpass.cpp:57-92) are template parameters here.
#include "vkv_device.hpp"

using namespace vkv;

struct RayMarchArgs
{
	// ray generator + RayCastUniform
	float dir00[3], ddx[3], ddy[3];
	float cam[3];
	float plane_tex[4];
	float block_size[3];
	// CameraUniform matrices needed for gl_FragDepth (frag:319)
	float model[16], view[16], proj[16];
	// TransferFunctionUniform
	float sampling_factor, grad_modifier;
	// extents
	int W, H, D, mw, mh, md;
	const uint8_t *vol, *grad, *tf;
	const uint8_t *packed;        // vkv_pack_volume image (PACKED variants) or null
	int            pmx, pmy;      // macro-bricks per axis of the packed image
	const uint32_t *tf_bits;      // 2048-word alpha>0 bit table or null
	const uint8_t *maps[8];
	float *        out_color;
	uint8_t *      out_rgba8;
	uint32_t *     out_counts;
	float *        out_depth;
	uint32_t       img_w, img_h, tile_w, tile_h, tiles_x, tile_first, tile_stride, tile_count, compact;
	uint32_t       blocks_per_tile_x, blocks_per_tile, nblocks;
	int            test;
	float          alpha_lut[256];        // opacity correction keyed by the TF alpha byte (frag:283)
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

template <bool WANT_G>
__device__ __forceinline__ void sample_packed(const uint8_t *__restrict__ P, int W, int H, int D, int pmx, int pmy, float px, float py, float pz,
                                              float &out_v, float &out_g)
{
	const float cx = __builtin_fmaf(px, (float) W, -0.5f), cy = __builtin_fmaf(py, (float) H, -0.5f), cz = __builtin_fmaf(pz, (float) D, -0.5f);
	const float fx = __builtin_floorf(cx), fy = __builtin_floorf(cy), fz = __builtin_floorf(cz);
	const float wx = cx - fx, wy = cy - fy, wz = cz - fz;
	const int   bx = i_clamp((int) fx, -1, W) + 1, by = i_clamp((int) fy, -1, H) + 1, bz = i_clamp((int) fz, -1, D) + 1;
	const uint8_t *b = P + packed_brick_offset(bx >> 2, by >> 2, bz >> 2, pmx, pmy) + (uint32_t) ((((bz & 3) * 5 + (by & 3)) * 5 + (bx & 3)) * 2);
	const uint32_t q00 = *reinterpret_cast<const u32_align2 *>(b);
	const uint32_t q10 = *reinterpret_cast<const u32_align2 *>(b + 10);
	const uint32_t q01 = *reinterpret_cast<const u32_align2 *>(b + 50);
	const uint32_t q11 = *reinterpret_cast<const u32_align2 *>(b + 60);
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

__device__ __forceinline__ void mat4_mul_vec4(const float *m, const float *v, float *r)
{
#pragma unroll
	for (int i = 0; i < 4; ++i)
		r[i] = __builtin_fmaf(m[12 + i], v[3], __builtin_fmaf(m[8 + i], v[2], __builtin_fmaf(m[4 + i], v[1], m[i] * v[0])));
}

__device__ __forceinline__ uint8_t quantise_rgba8(float c) { return (uint8_t) __builtin_rintf(g_clamp(c, 0.0f, 1.0f) * 255.0f); }

// SKIP: VkvSkippingType; ERT: early ray termination; GRAD: 0 = use_gradient false, 1 = precomputed map, 2 = on the fly.
// PACKED: sample the vkv_pack_volume image instead of the linear buffers.
template <int SKIP, bool ERT, int GRAD, bool PACKED>
__global__ void __launch_bounds__(256) k_raymarch(const RayMarchArgs A)
{
	__shared__ float    s_alpha[256];
	__shared__ uint32_t s_bits[2048];
	s_alpha[threadIdx.x] = A.alpha_lut[threadIdx.x];
	const bool tf_bits = A.tf_bits != nullptr;        // wave-uniform
	if (tf_bits)
		for (int i = threadIdx.x; i < 2048; i += 256)
			s_bits[i] = A.tf_bits[i];
	__syncthreads();

	// ---- workgroup -> 16x16 pixel block of one scheduled tile -------------------------------------------------
	const uint32_t b  = xcd_remap(blockIdx.x, A.nblocks);
	const uint32_t k  = b / A.blocks_per_tile;        // index into this launch's tile list
	const uint32_t sb = b % A.blocks_per_tile;
	const uint32_t t  = A.tile_first + k * A.tile_stride;
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t lx = (sb % A.blocks_per_tile_x) * 16 + (wave & 1) * 8 + (lane & 7);        // pixel inside the tile
	const uint32_t ly = (sb / A.blocks_per_tile_x) * 16 + (wave >> 1) * 8 + (lane >> 3);
	const uint32_t px = (t % A.tiles_x) * A.tile_w + lx, py = (t / A.tiles_x) * A.tile_h + ly;
	if (px >= A.img_w || py >= A.img_h)
		return;
	const size_t o = A.compact ? ((size_t) k * A.tile_h + ly) * A.tile_w + lx : (size_t) py * A.img_w + px;

	float    out_r = 0.0f, out_g = 0.0f, out_b = 0.0f, out_a = 0.0f, out_depth = 0.0f;        // frag:120, :140
	uint32_t n_vol = 0, n_dist = 0, n_empty = 0;

	const int W = A.W, H = A.H, D = A.D;
	// `do { ... } while (0)` so every early-out of the shader funnels into the single store block below
	do
	{
		// ---- ray generation (replaces volume_render_clipped.vert + volume_render_plane_intersection.vert) ------
		const float fx = (float) px + 0.5f, fy = (float) py + 0.5f;
		float       dx = __builtin_fmaf(fy, A.ddy[0], __builtin_fmaf(fx, A.ddx[0], A.dir00[0]));
		float       dy = __builtin_fmaf(fy, A.ddy[1], __builtin_fmaf(fx, A.ddx[1], A.dir00[1]));
		float       dz = __builtin_fmaf(fy, A.ddy[2], __builtin_fmaf(fx, A.ddx[2], A.dir00[2]));
		{
			const float len = __builtin_sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
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
					const float inv = 1.0f / dv[a];
					const float ta = (0.0f - ov[a]) * inv, tb = (1.0f - ov[a]) * inv;
					t_near = g_max(t_near, g_min(ta, tb));
					t_far  = g_min(t_far, g_max(ta, tb));
				}
			}
		}
		if (miss)
			break;
		const float Ap = __builtin_fmaf(A.plane_tex[2], oz, __builtin_fmaf(A.plane_tex[1], oy, A.plane_tex[0] * ox)) + A.plane_tex[3];
		const float Bp = __builtin_fmaf(A.plane_tex[2], dz, __builtin_fmaf(A.plane_tex[1], dy, A.plane_tex[0] * dx));
		if (!(Bp > 0.0f))
			break;
		const float t_plane = (0.0f - Ap) / Bp;
		const float t0      = g_max(t_near, t_plane);
		if (!(t0 < t_far))
			break;
		const float ex = __builtin_fmaf(t0, dx, ox), ey = __builtin_fmaf(t0, dy, oy), ez = __builtin_fmaf(t0, dz, oz);        // ray_entry

		// ---- frag:147-149 ------------------------------------------------------------------------------------
		float rdx, rdy, rdz;
		{
			const float vx = ex - ox, vy = ey - oy, vz = ez - oz;
			const float len = __builtin_sqrtf(__builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx)));
			rdx = vx / len, rdy = vy / len, rdz = vz / len;
		}
		float xx, xy, xz, ray_distance;        // ray_exit
		{
			const float ix = 1.0f / rdx, iy = 1.0f / rdy, iz = 1.0f / rdz;
			const float tminx = -ex * ix, tminy = -ey * iy, tminz = -ez * iz;
			const float tmaxx = (1.0f - ex) * ix, tmaxy = (1.0f - ey) * iy, tmaxz = (1.0f - ez) * iz;
			const float t2x = g_max(tminx, tmaxx), t2y = g_max(tminy, tmaxy), t2z = g_max(tminz, tmaxz);
			const float tFar = g_min(g_min(t2x, t2y), t2z);
			xx = __builtin_fmaf(tFar, rdx, ex), xy = __builtin_fmaf(tFar, rdy, ey), xz = __builtin_fmaf(tFar, rdz, ez);
			const float vx = ex - xx, vy = ey - xy, vz = ez - xz;
			ray_distance = __builtin_sqrtf(__builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx)));
		}
		if (A.test == VKV_TEST_RAY_ENTRY)
		{
			out_r = ex, out_g = ey, out_b = ez, out_a = 1.0f;
			break;
		}
		if (A.test == VKV_TEST_RAY_EXIT)
		{
			out_r = xx, out_g = xy, out_b = xz, out_a = 1.0f;
			break;
		}

		// ---- frag:176-187 ------------------------------------------------------------------------------------
		const int   dim_max = max(max(W, H), D);
		const float sf      = A.sampling_factor;
		const float nf      = __builtin_ceilf((float) dim_max * ray_distance * sf);
		if (!(nf >= 2.0f && nf <= 16777216.0f))
			break;
		const int   n_steps = (int) nf;
		const float sx = (rdx * ray_distance) / (nf - 1.0f), sy = (rdy * ray_distance) / (nf - 1.0f), sz = (rdz * ray_distance) / (nf - 1.0f);
		{
			const float qx = ex + sx, qy = ey + sy, qz = ez + sz;
			if (qx <= 0.0f || qy <= 0.0f || qz <= 0.0f || qx >= 1.0f || qy >= 1.0f || qz >= 1.0f)
				break;
		}

		// ---- frag:191-210 ------------------------------------------------------------------------------------
		float kx = 0, ky = 0, kz = 0, six = 0, siy = 0, siz = 0;
		if (SKIP != VKV_SKIP_NONE)
		{
			kx = (float) W / A.block_size[0], ky = (float) H / A.block_size[1], kz = (float) D / A.block_size[2];
			six = 1.0f / ((sx * (float) W) / A.block_size[0]);
			siy = 1.0f / ((sy * (float) H) / A.block_size[1]);
			siz = 1.0f / ((sz * (float) D) / A.block_size[2]);
		}
		const int      mw = A.mw, mh = A.mh, md = A.md;
		const uint8_t *dmap = nullptr;
		if (SKIP == VKV_SKIP_ANISOTROPIC_DISTANCE)
			dmap = A.maps[(rdz < 0 ? 1 : 0) + (rdy < 0 ? 2 : 0) + (rdx < 0 ? 4 : 0)];
		else if (SKIP != VKV_SKIP_NONE)
			dmap = A.maps[0];
		const float dix = 1.0f / (float) W, diy = 1.0f / (float) H, diz = 1.0f / (float) D;
		const int   back = (int) __builtin_ceilf(sf);

		int  i_min_ = 0, ulx = 0, uly = 0, ulz = 0;
		bool occupied = true;
		int  i_first_hit = n_steps;

		// ---- frag:215-312 ------------------------------------------------------------------------------------
		for (int i = 0; i < n_steps;)
		{
			const float fi = (float) i;
			const float posx = __builtin_fmaf(fi, sx, ex), posy = __builtin_fmaf(fi, sy, ey), posz = __builtin_fmaf(fi, sz, ez);
			int   uix = 0, uiy = 0, uiz = 0;
			float ux = 0, uy = 0, uz = 0;
			if (SKIP != VKV_SKIP_NONE)
			{
				ux = kx * posx, uy = ky * posy, uz = kz * posz;
				uix = i_clamp((int) ux, 0, mw - 1), uiy = i_clamp((int) uy, 0, mh - 1), uiz = i_clamp((int) uz, 0, md - 1);
			}
			if (SKIP != VKV_SKIP_NONE && !occupied && (uix != ulx || uiy != uly || uiz != ulz))
			{
				++n_dist;
				const uint32_t dist = dmap[vidx(uix, uiy, uiz, mw, mh)];
				if (dist > 0u)
				{
					const float rx = g_clamp((float) uix - ux, -1.0f, 0.0f);
					const float ry = g_clamp((float) uiy - uy, -1.0f, 0.0f);
					const float rz = g_clamp((float) uiz - uz, -1.0f, 0.0f);
					float       ax, ay, az;
					if (SKIP == VKV_SKIP_BLOCK)
					{
						ax = (g_step(0.0f, six) + rx) * six;
						ay = (g_step(0.0f, siy) + ry) * siy;
						az = (g_step(0.0f, siz) + rz) * siz;
					}
					else
					{
						const float fd = (float) dist;
						ax = ((g_step(0.0f, -six) + g_sign(six) * fd) + rx) * six;
						ay = ((g_step(0.0f, -siy) + g_sign(siy) * fd) + ry) * siy;
						az = ((g_step(0.0f, -siz) + g_sign(siz) * fd) + rz) * siz;
					}
					if (ax != ax) ax = INFINITY;
					if (ay != ay) ay = INFINITY;
					if (az != az) az = INFINITY;
					float m = g_min(g_min(ax, ay), az);
					m       = (m < 1073741824.0f) ? m : 1073741824.0f;
					i += max(1, (int) __builtin_ceilf(m));
				}
				else
				{
					occupied = true;
					ulx = uix, uly = uiy, ulz = uiz;
					i = max(i - back, i_min_);
				}
			}
			else
			{
				++n_vol;
				float intensity, gradient = 1.0f;
				if (PACKED)
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
				{
					float t1, t2, t3, t4, unused;
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
				// get_color (transfer_function.glsl:35-38): NEAREST texel.  With the bit table the occupied test (frag:276)
				// comes from LDS and only occupied samples pay the dependent RGBA fetch.
				const uint32_t tidx  = (uint32_t) tf_texel(gradient) * 256u + (uint32_t) tf_texel(intensity);
				uint32_t       texel = 0;
				if (tf_bits)
				{
					if ((s_bits[tidx >> 5] >> (tidx & 31u)) & 1u)
						texel = reinterpret_cast<const uint32_t *>(A.tf)[tidx];
				}
				else
					texel = reinterpret_cast<const uint32_t *>(A.tf)[tidx];
				const uint32_t ab = texel >> 24;
				occupied          = ab > 0;
				if (occupied)
				{
					if (SKIP != VKV_SKIP_NONE)
						ulx = uix, uly = uiy, ulz = uiz;
					const float a  = s_alpha[ab];
					const float r_ = unorm8(texel & 255u) * a, g_ = unorm8((texel >> 8) & 255u) * a, b_ = unorm8((texel >> 16) & 255u) * a;
					const float om = 1.0f - out_a;
					out_r = __builtin_fmaf(om, r_, out_r), out_g = __builtin_fmaf(om, g_, out_g), out_b = __builtin_fmaf(om, b_, out_b);
					out_a = __builtin_fmaf(om, a, out_a);
					if (a > 0.0f)
						i_first_hit = i;
					if (ERT && out_a > 0.99f)
					{
						out_a = 1.0f;
						break;
					}
				}
				else
					++n_empty;
				++i;
				i_min_ = i;
			}
		}

		// ---- frag:315-321 ------------------------------------------------------------------------------------
		if (out_a > 0.0f && i_first_hit < n_steps)
		{
			const float fi   = (float) i_first_hit;
			const float p[4] = {__builtin_fmaf(fi, sx, ex) - 0.5f, __builtin_fmaf(fi, sy, ey) - 0.5f, __builtin_fmaf(fi, sz, ez) - 0.5f, 1.0f};
			float       a4[4], b4[4], c4[4];
			mat4_mul_vec4(A.model, p, a4);
			mat4_mul_vec4(A.view, a4, b4);
			mat4_mul_vec4(A.proj, b4, c4);
			out_depth = c4[2] / c4[3];
		}
		if (A.test == VKV_TEST_NUM_TEXTURE_SAMPLES)
		{        // frag:324-334
			const uint32_t n_steps_max = (uint32_t) (__builtin_ceilf((float) dim_max * __builtin_sqrtf(3.0f)) * sf);
			const float    v           = (float) (n_vol + n_dist) / (float) n_steps_max;
			out_r = out_g = out_b = v;
			out_a                 = 1.0f;
		}
	} while (0);

	if (A.out_color)
		reinterpret_cast<float4 *>(A.out_color)[o] = make_float4(out_r, out_g, out_b, out_a);
	if (A.out_rgba8)
		reinterpret_cast<uint32_t *>(A.out_rgba8)[o] = (uint32_t) quantise_rgba8(out_r) | ((uint32_t) quantise_rgba8(out_g) << 8) |
		                                               ((uint32_t) quantise_rgba8(out_b) << 16) | ((uint32_t) quantise_rgba8(out_a) << 24);
	if (A.out_counts)
	{
		A.out_counts[o * 3 + 0] = n_vol;
		A.out_counts[o * 3 + 1] = n_dist;
		A.out_counts[o * 3 + 2] = n_empty;
	}
	if (A.out_depth)
		A.out_depth[o] = out_depth;
}

namespace vkv
{

template <int SKIP, bool ERT, bool PACKED>
static void launch_grad(int grad, const RayMarchArgs &a, hipStream_t s)
{
	const dim3 grid(a.nblocks), block(256);
	if (grad == 0)
		hipLaunchKernelGGL((k_raymarch<SKIP, ERT, 0, PACKED>), grid, block, 0, s, a);
	else if (grad == 1)
		hipLaunchKernelGGL((k_raymarch<SKIP, ERT, 1, PACKED>), grid, block, 0, s, a);
	else
		hipLaunchKernelGGL((k_raymarch<SKIP, ERT, 2, PACKED>), grid, block, 0, s, a);
}

template <int SKIP>
static void launch_ert(bool ert, int grad, const RayMarchArgs &a, hipStream_t s)
{
	if (a.packed)
	{
		if (ert)
			launch_grad<SKIP, true, true>(grad, a, s);
		else
			launch_grad<SKIP, false, true>(grad, a, s);
	}
	else
	{
		if (ert)
			launch_grad<SKIP, true, false>(grad, a, s);
		else
			launch_grad<SKIP, false, false>(grad, a, s);
	}
}

int launch_render(vkv_ctx *ctx, const VkvRenderParams *P, const float *alpha_lut, hipStream_t s)
{
	RayMarchArgs a;
	for (int i = 0; i < 3; ++i)
	{
		a.dir00[i] = P->ray_gen.dir00[i], a.ddx[i] = P->ray_gen.ddx[i], a.ddy[i] = P->ray_gen.ddy[i];
		a.cam[i]        = P->ray_cast.camera_pos_tex[i];
		a.block_size[i] = P->ray_cast.block_size[i];
	}
	for (int i = 0; i < 4; ++i)
		a.plane_tex[i] = P->ray_cast.plane_tex[i];
	for (int i = 0; i < 16; ++i)
		a.model[i] = P->camera.model[i], a.view[i] = P->camera.camera_view[i], a.proj[i] = P->camera.camera_proj[i];
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
	a.img_w = P->image_width, a.img_h = P->image_height;
	a.tile_w = P->tiles.tile_width, a.tile_h = P->tiles.tile_height;
	a.tiles_x    = (a.img_w + a.tile_w - 1) / a.tile_w;
	a.tile_first = P->tiles.tile_first, a.tile_stride = P->tiles.tile_stride, a.tile_count = P->tiles.tile_count, a.compact = P->tiles.compact;
	a.blocks_per_tile_x = a.tile_w / 16;
	a.blocks_per_tile   = a.blocks_per_tile_x * (a.tile_h / 16);
	const uint64_t nb   = (uint64_t) a.blocks_per_tile * a.tile_count;
	if (nb == 0)
		return VKV_OK;
	if (nb > 0x7fffffffull)
		return set_error(ctx, VKV_E_UNSUPPORTED, "render: too many tiles for one launch");
	a.nblocks = (uint32_t) nb;
	a.test    = P->options.test;
	for (int i = 0; i < 256; ++i)
		a.alpha_lut[i] = alpha_lut[i];

	const bool ert  = P->options.early_ray_termination != 0;
	const int  grad = !P->transfer_function.use_gradient ? 0 : (P->use_precomputed_gradient ? 1 : 2);
	switch (P->options.skipping_type)
	{
		case VKV_SKIP_NONE: launch_ert<VKV_SKIP_NONE>(ert, grad, a, s); break;
		case VKV_SKIP_BLOCK: launch_ert<VKV_SKIP_BLOCK>(ert, grad, a, s); break;
		case VKV_SKIP_DISTANCE: launch_ert<VKV_SKIP_DISTANCE>(ert, grad, a, s); break;
		case VKV_SKIP_ANISOTROPIC_DISTANCE: launch_ert<VKV_SKIP_ANISOTROPIC_DISTANCE>(ert, grad, a, s); break;
		default: return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: bad skipping_type %d", P->options.skipping_type);
	}
	return check_launch(ctx, "render");
}

}        // namespace vkv
