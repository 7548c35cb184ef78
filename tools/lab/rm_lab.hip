// rm_lab.hip — experimental instantiations of the ray-march kernels (vkvolume_amd/csrc/raymarch_core.hpp), A/B-ed on the GPU by
// tools/lab/run_lab.py against the product's vkv_render before a variant moves into the product.  Not part of the product.
#include <algorithm>
#include <cmath>
#include <cstdio>

#include "raymarch_lab.hpp"

namespace
{
template <int SKIP, bool ERT, int GRAD, bool PACKED, int W, uint32_t FLAGS>
int launch_er(const RayMarchArgs &a, hipStream_t s)
{
	const uint32_t grid = ((a.tile_count + 7u) / 8u) * 8u * a.blocks_per_tile * W;
	hipLaunchKernelGGL((k_raymarch_er<SKIP, ERT, GRAD, PACKED, W, FLAGS>), dim3(grid), dim3(256), 0, s, a);
	return (int) hipGetLastError();
}

template <int SKIP, bool ERT, int GRAD, bool PACKED, uint32_t LF>
__global__ void __launch_bounds__(256) k_lab_lean(const RayMarchArgs A)
{
	lean_lds_check();
	lab_lean_block<SKIP, ERT, GRAD, PACKED, LF>(A, blockIdx.x, lean_lds());
}

// the PRODUCT's kernel with flags the product does not instantiate (the stamp build): cycles per iteration by kind of the shipped loop
template <int SKIP, bool ERT, int GRAD, uint32_t LF>
int launch_product_lean(const RayMarchArgs &a, hipStream_t s)
{
	const uint32_t grid = ((a.tile_count + 7u) / 8u) * 8u * a.blocks_per_tile;
	if ((LF & kLeanLut) && (!a.addr_lut || (size_t) a.lut_words * 4 > kMaxLutBytes || !map_fits_u24((uint32_t) a.mw, (uint32_t) a.mh, (uint32_t) a.md)))
		return -103;
	if ((LF & kLeanFull) != 0 && kFullLutWord * 4 + full_lut_bytes(a.W, a.H, a.D) > 48 * 1024)
		return -104;
	const size_t lds = lean_lds_bytes((LF & kLeanFull) ? 2 : ((LF & kLeanLut) ? 1 : 0), a.lut_words, a.W, a.H, a.D);
	hipLaunchKernelGGL((k_raymarch_lean<SKIP, ERT, GRAD, true, LF>), dim3(grid), dim3(256), lds, s, a);
	return (int) hipGetLastError();
}

template <int SKIP, bool ERT, int GRAD, bool PACKED, uint32_t LF>
int launch_lean(const RayMarchArgs &a, hipStream_t s)
{
	const uint32_t grid = ((a.tile_count + 7u) / 8u) * 8u * a.blocks_per_tile;
	if ((LF & kLabLut) && (!a.addr_lut || (size_t) a.lut_words * 4 > 48 * 1024))
		return -103;
	if ((LF & kLabFull) != 0 && kFullLutWord * 4 + full_lut_bytes(a.W, a.H, a.D) > 48 * 1024)
		return -104;
	const size_t lds = lean_lds_bytes((LF & kLabFull) ? 2 : ((LF & kLabLut) ? 1 : 0), a.lut_words, a.W, a.H, a.D);
	if ((LF & kLabScalar) != 0 && !map_fits_u24((uint32_t) a.mw, (uint32_t) a.mh, (uint32_t) a.md))
		return -105;
	hipLaunchKernelGGL((k_lab_lean<SKIP, ERT, GRAD, PACKED, LF>), dim3(grid), dim3(256), lds, s, a);
	return (int) hipGetLastError();
}

template <int SKIP, bool ERT, int GRAD, uint32_t LF, int WPB>
__global__ void __launch_bounds__(WPB * 64) k_lab_lean_wpb(const RayMarchArgs A)
{
	lean_lds_check();
	lab_lean_block<SKIP, ERT, GRAD, true, LF, WPB>(A, blockIdx.x, lean_lds());
}

template <int SKIP, bool ERT, int GRAD, int WPB, uint32_t LF = kLabDefault | kLabLut>
int launch_wpb(const RayMarchArgs &a, hipStream_t s)
{
	const uint32_t grid = ((a.tile_count + 7u) / 8u) * 8u * a.blocks_per_tile * (4 / WPB);
	if (!a.addr_lut || (size_t) a.lut_words * 4 > 48 * 1024)
		return -103;
	hipLaunchKernelGGL((k_lab_lean_wpb<SKIP, ERT, GRAD, LF, WPB>), dim3(grid), dim3(WPB * 64), lean_lds_bytes((LF & kLabFull) ? 2 : 1, a.lut_words, a.W, a.H, a.D), s, a);
	return (int) hipGetLastError();
}
// the product's flag sets (vkvolume_amd/csrc/raymarch.hip)
constexpr uint32_t kLabSetLut  = kLabDefault | kLabNest | kLabKeep | kLabTf | kLabWb | kLabFloatI | kLabScalar | kLabLut;
constexpr uint32_t kLabSetFull = kLabSetLut | kLabFull;

template <int SKIP, bool ERT, int GRAD>
int dispatch(const RayMarchArgs &a, int variant, hipStream_t s)
{
	switch (variant)
	{
#ifdef LAB_ALL
		case 102: return launch_wpb<SKIP, ERT, GRAD, 2>(a, s);
		case 101: return launch_wpb<SKIP, ERT, GRAD, 1>(a, s);
		case 104: return launch_wpb<SKIP, ERT, GRAD, 4>(a, s);
		case 9:        // the round 1 kernel (divergent probe / sample branches, texel fetch from memory)
		{
			const uint32_t grid = ((a.tile_count + 7u) / 8u) * 8u * a.blocks_per_tile;
			hipLaunchKernelGGL((k_raymarch_tiles<SKIP, ERT, GRAD, true, 4>), dim3(grid), dim3(256), 0, s, a);
			return (int) hipGetLastError();
		}
#endif
		case 111: return launch_wpb<SKIP, ERT, GRAD, 1, kLabSetLut>(a, s);        // workgroups of 1 / 2 / 4 waves with the two-level tables (5.3 KB of LDS per workgroup)
		case 112: return launch_wpb<SKIP, ERT, GRAD, 2, kLabSetLut>(a, s);
		case 114: return launch_wpb<SKIP, ERT, GRAD, 4, kLabSetLut>(a, s);
		case 122: return launch_wpb<SKIP, ERT, GRAD, 2, kLabSetFull>(a, s);       // ... with the per-index tables (15.5 KB: 10 workgroups per CU)
		case 124: return launch_wpb<SKIP, ERT, GRAD, 4, kLabSetFull>(a, s);
		case 8: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar>(a, s);
		case 10: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest>(a, s);
		case 13: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep>(a, s);
		case 15: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabFull>(a, s);
		case 16: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull>(a, s);
		case 18: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf>(a, s);
		case 19: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabGradSkip>(a, s);
		case 20: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb>(a, s);
		case 21: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb | kLabFloatI>(a, s);
		case 22: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb | kLabFloatI | kLabFloatCell>(a, s);
		case 24: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb | kLabFloatI | kLabFmt>(a, s);
		case 25: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb | kLabFloatI | kLabFmt | kLabFmtVec>(a, s);
		case 26: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb | kLabFloatI | kLabStamp>(a, s);
		// round 4: bricked distance map (the caller passes maps laid out by vkv_lab_brick_map), and the probes-only timing variants
		case 40: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb | kLabFloatI | kLabBrickMap>(a, s);
		case 41: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb | kLabFloatI | kLabProbeOnly>(a, s);
		case 42: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb | kLabFloatI | kLabProbeOnly | kLabBrickMap>(a, s);
		case 43: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb | kLabFloatI | kLabStamp | kLabBrickMap>(a, s);
		case 44: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb | kLabFloatI | kLabHalfRows>(a, s);
		case 23: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabScalar | kLabNest | kLabKeep | kLabFull | kLabTf | kLabWb | kLabFloatI | kLabPrefetch>(a, s);
		case 17: return launch_lean<SKIP, ERT, GRAD, true, kLabDefault | kLabLut | kLabFull>(a, s);
		// the product's lean_march (raymarch_core.hpp), instantiated here: 30 = kLeanLut | kLeanFull as shipped, 31 = + per-iteration stamps
		case 30: return launch_product_lean<SKIP, ERT, GRAD, kLeanLut | kLeanFull>(a, s);
		case 31: return launch_product_lean<SKIP, ERT, GRAD, kLeanLut | kLeanFull | kLeanStamp>(a, s);
		case 32: return launch_product_lean<SKIP, ERT, GRAD, kLeanLut | kLeanFull | kLeanSafe>(a, s);                     // round 5: + the clamp-free march loop
		case 33: return launch_product_lean<SKIP, ERT, GRAD, kLeanLut | kLeanFull | kLeanSafe | kLeanAsync>(a, s);        // + hand-set load waits = the product's kLfFull
		case 6: return launch_lean<SKIP, ERT, GRAD, true, kLabUniform | kLabLut | kLabBranch | kLabCvt>(a, s);
#ifdef LAB_ALL
		case 1: return launch_lean<SKIP, ERT, GRAD, true, 0>(a, s);
		case 2: return launch_lean<SKIP, ERT, GRAD, true, kLabUniform>(a, s);
		case 3: return launch_lean<SKIP, ERT, GRAD, true, kLabUniform | kLabLut>(a, s);
		case 4: return launch_lean<SKIP, ERT, GRAD, true, kLabUniform | kLabBranch>(a, s);
		case 5: return launch_lean<SKIP, ERT, GRAD, true, kLabUniform | kLabCvt>(a, s);
		case 7: return launch_lean<SKIP, ERT, GRAD, true, kLabUniform | kLabLut | kLabCvt>(a, s);
		case 211: return launch_er<SKIP, ERT, GRAD, true, 1, 0>(a, s);
		case 212: return launch_er<SKIP, ERT, GRAD, true, 2, 0>(a, s);
		case 214: return launch_er<SKIP, ERT, GRAD, true, 4, 0>(a, s);
		case 218: return launch_er<SKIP, ERT, GRAD, true, 8, 0>(a, s);
		case 221: return launch_er<SKIP, ERT, GRAD, true, 1, kErFull>(a, s);
		case 222: return launch_er<SKIP, ERT, GRAD, true, 2, kErFull>(a, s);
		case 224: return launch_er<SKIP, ERT, GRAD, true, 4, kErFull>(a, s);
		case 241: return launch_er<SKIP, ERT, GRAD, true, 1, kErMasked>(a, s);
		case 242: return launch_er<SKIP, ERT, GRAD, true, 2, kErMasked>(a, s);
		case 251: return launch_er<SKIP, ERT, GRAD, true, 1, kErMasked | kErStamp>(a, s);
		case 252: return launch_er<SKIP, ERT, GRAD, true, 2, kErMasked | kErStamp>(a, s);
		case 261: return launch_er<SKIP, ERT, GRAD, true, 1, kErCache16>(a, s);
		case 262: return launch_er<SKIP, ERT, GRAD, true, 2, kErCache16>(a, s);
		case 271: return launch_er<SKIP, ERT, GRAD, true, 1, kErCache32>(a, s);
		case 272: return launch_er<SKIP, ERT, GRAD, true, 2, kErCache32>(a, s);
		case 281: return launch_er<SKIP, ERT, GRAD, true, 1, kErCache32 | kErStamp>(a, s);
		case 231: return launch_er<SKIP, ERT, GRAD, true, 1, kErStamp>(a, s);
		case 232: return launch_er<SKIP, ERT, GRAD, true, 2, kErStamp>(a, s);
		case 234: return launch_er<SKIP, ERT, GRAD, true, 4, kErStamp>(a, s);
#endif
		default: return -100;
	}
}
}        // namespace

// dense [md][mh][mw] distance map -> 4x4x4-cell bricks of 64 bytes, bricks x fastest; cells beyond the map's edge are 0 (never read: the
// integrator clamps its cell coordinates).  dst holds 64 * ceil(mw / 4) * ceil(mh / 4) * ceil(md / 4) bytes.
__global__ void __launch_bounds__(256) k_lab_brick_map(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t mw, uint32_t mh, uint32_t md, uint64_t total)
{
	const uint64_t i = (uint64_t) blockIdx.x * 256u + threadIdx.x;
	if (i >= total)
		return;
	const uint32_t bw = (mw + 3u) >> 2, bh = (mh + 3u) >> 2;
	const uint32_t in = (uint32_t) (i & 63u);
	const uint64_t b  = i >> 6;
	const uint32_t bx = (uint32_t) (b % bw), by = (uint32_t) ((b / bw) % bh), bz = (uint32_t) (b / ((uint64_t) bw * bh));
	const uint32_t x = bx * 4u + (in & 3u), y = by * 4u + ((in >> 2) & 3u), z = bz * 4u + (in >> 4);
	dst[i] = (x < mw && y < mh && z < md) ? src[((size_t) z * mh + y) * mw + x] : (uint8_t) 0;
}

extern "C" int vkv_lab_brick_map(const uint8_t *d_src, uint8_t *d_dst, uint32_t mw, uint32_t mh, uint32_t md, void *stream)
{
	const uint64_t total = 64ull * ((mw + 3u) >> 2) * ((mh + 3u) >> 2) * ((md + 3u) >> 2);
	hipLaunchKernelGGL(k_lab_brick_map, dim3((uint32_t) ((total + 255) / 256)), dim3(256), 0, (hipStream_t) stream, d_src, d_dst, mw, mh, md, total);
	return (int) hipGetLastError();
}

// LAB_ALL variants = 200 + the ids of the first half of profiles/r2_lab_variants.txt: 210 + W evaluate+replay with W lanes per ray;
// 220 + W: every lane loads both kinds; 230 + W: stamped diagnostic build; 24x masked loads; 26x / 27x brick cache.
// Only the bench configurations are instantiated: (distance ESS, ERT, precomputed gradient, packed) and (no ESS, no ERT, ...).
extern "C" int vkv_lab_render(vkv_ctx *ctx, const VkvRenderParams *P, int variant, void *stream)
{
	float       lut[256];
	const float sf_inv = 1.0f / P->transfer_function.sampling_factor;
	for (int a = 0; a < 256; ++a)
	{
		const float v = P->transfer_function.voxel_alpha_factor * (1.0f - std::pow(1.0f - (float) a / 255.0f, sf_inv));
		lut[a]        = std::min(std::max(v, 0.0f), 1.0f);
	}
	RayMarchArgs a;
	const int    rc = vkv::fill_render_args(ctx, P, lut, a, (hipStream_t) stream, vkv::tuning_of(ctx), false, false);
	if (rc != VKV_OK || a.nblocks == 0)
		return rc;
	if (!a.packed || !P->transfer_function.use_gradient || !P->use_precomputed_gradient)
		return -101;
	const bool ert = P->options.early_ray_termination != 0;
	if (P->options.skipping_type == VKV_SKIP_DISTANCE && ert)
		return dispatch<VKV_SKIP_DISTANCE, true, 1>(a, variant, (hipStream_t) stream);
	if (P->options.skipping_type == VKV_SKIP_NONE && !ert)
		return dispatch<VKV_SKIP_NONE, false, 1>(a, variant, (hipStream_t) stream);
	if (P->options.skipping_type == VKV_SKIP_BLOCK && ert)
		return dispatch<VKV_SKIP_BLOCK, true, 1>(a, variant, (hipStream_t) stream);
	return -102;
}
