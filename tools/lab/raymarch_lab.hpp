// raymarch_lab.hpp — experimental and retired variants of the ray-march integrator, on top of the product's device code
// (vkvolume_amd/csrc/raymarch_core.hpp).  Included by rm_lab.hip only; NOTHING here is compiled into libvkvolume_amd.so.
//   k_raymarch_tiles    round 1 kernel: one lane per ray, divergent probe / sample branches (ray_event)
//   k_raymarch_er       "evaluate + replay": W lanes share one ray, with the wave-private LDS brick cache (BrickCache) as an option -
//                       measured slower in every regime (profiles/HISTORY.md)
//   lab_lean_march      round 3's lean_march with all twenty of its A/B flags (the product keeps the winning combination hard-wired);
//                       kLab* = the flags; the paths measured and rejected: non-temporal footprint loads, select-based state update,
//                       un-nested outcome blocks, kLabGradSkip, kLabFloatCell, kLabPrefetch, kLabFmt / kLabFmtVec (buffer FORMAT loads)
// Every variant must render the product's frames bit for bit (tools/lab/run_lab.py checks before it times).
#pragma once

#include "../../vkvolume_amd/csrc/raymarch_core.hpp"

// round 3's flag values (the product's kLean* are a different, smaller set)
constexpr uint32_t kLabUniform = 1u;        // wave-uniform branches around the probe-only and the sample-only work
constexpr uint32_t kLabNt      = 2u;        // non-temporal footprint loads (leave the caches to the distance map)
constexpr uint32_t kLabLut     = 4u;        // footprint address from per-axis tables in LDS
constexpr uint32_t kLabBranch  = 8u;        // state update as EXEC-masked branches instead of selects
constexpr uint32_t kLabCvt     = 16u;       // cost-aware instruction selection (v_cvt_f32_ubyteN, fma instead of select)
constexpr uint32_t kLabNest    = 32u;       // probe / sample arithmetic inside the state update's EXEC-masked blocks
constexpr uint32_t kLabKeep    = 64u;       // with kLabNest: keep the loads ahead of both blocks
constexpr uint32_t kLabScalar  = 128u;      // clamp bounds from scalar registers, 24-bit multiply-adds for the cell index
constexpr uint32_t kLabFull    = 256u;      // one address-table entry per voxel index and axis (separable transfer function)
constexpr uint32_t kLabTf      = 512u;      // separable transfer function: table index without a clamp
constexpr uint32_t kLabGradSkip = 1024u;    // gradient channel only filtered when some lane's intensity alpha is > 0
constexpr uint32_t kLabFloatI  = 2048u;     // loop position, bounds and first hit as floats
constexpr uint32_t kLabWb      = 4096u;     // kLabKeep through a wave barrier instead of a laundered predicate
constexpr uint32_t kLabFloatCell = 8192u;   // cell coordinates clamped in float, linear cell index from two fmas + one conversion
constexpr uint32_t kLabPrefetch = 16384u;   // footprint of position i + 1 requested one iteration ahead while the previous sample was occupied
constexpr uint32_t kLabFmt     = 32768u;    // x-pair rows through buffer FORMAT loads (f16 operands of v_fma_mix_f32); packed image below 4 GiB
constexpr uint32_t kLabFmtVec  = 65536u;    // kLabFmt with the uniform loop operands left in vector registers
constexpr uint32_t kLabStamp   = 131072u;   // s_memtime at the top of every iteration, summed per wave by the iteration's kind
constexpr uint32_t kLabNoCounts = 262144u;  // the three per-pixel counters are not kept
constexpr uint32_t kLabBrickMap = 524288u;   // (round 4) the distance map the probes read is BRICKED: 4x4x4 cells = one 64-byte block (bricks x fastest), as
                                             // vkv_lab_brick_map lays it out; the cell id compared with u_last_alpha is the bricked index (any injective id works)
constexpr uint32_t kLabProbeOnly = 1048576u; // (round 4, timing / cache-counter runs only: the frame is WRONG) the footprint gathers are not issued (the filter sees zeros),
                                             // so that the probes are the only vector memory traffic of the loop
constexpr uint32_t kLabHalfRows = 2097152u;  // (round 4, timing only: the frame is WRONG) two of the four footprint gathers are not issued (rows y1 reuse rows y0's
                                             // dwords): what a layout that delivers a footprint in two gathers could gain at unchanged VALU work
constexpr uint32_t kLabDefault = kLabUniform | kLabBranch | kLabCvt;

// ---------------------------------------------------------------------------------------------------------------
// Static scheduler: workgroup = 16x16 pixels, wave = 8x8 pixels.
// ---------------------------------------------------------------------------------------------------------------
// WPB = waves per workgroup: 4 (one workgroup = one 16x16 pixel block) or 2 (half a block: wave slots and LDS are handed back
// at a finer grain while the long rays of the other half are still running).
template <int SKIP, bool ERT, int GRAD, bool PACKED, int WPB>
__global__ void __launch_bounds__(WPB * 64) k_raymarch_tiles(const RayMarchArgs A)
{
	__shared__ float    s_alpha[256], s_unorm[256];
	__shared__ uint32_t s_bits[2048];
	stage_tables(A, s_alpha, s_unorm, s_bits);
	const bool     tf_bits = A.tf_bits != nullptr;
	// Hardware deals workgroup ids round-robin over the 8 XCDs (own L2 each).  XCD x = id & 7 marches the schedule's
	// tiles k = x, x + 8, x + 16, ... one after the other: neighbouring workgroups of an XCD share a tile (L2 locality)
	// while the tiles of the frame are spread evenly over the XCDs (ESS makes screen regions differ >10x in cost; a
	// contiguous band per XCD left most of the chip idle behind the XCD that owned the centre of the image).
	constexpr uint32_t kParts = 4 / WPB;        // workgroups per 16x16 block
	const uint32_t x = blockIdx.x & 7u, idx = blockIdx.x >> 3;
	const uint32_t k = (idx / (A.blocks_per_tile * kParts)) * 8u + x, sbp = idx % (A.blocks_per_tile * kParts);
	const uint32_t sb = sbp / kParts, part = sbp % kParts;
	if (k >= A.tile_count)
		return;
	uint32_t px, py, o;
	if (!unit_pixel(A, (k * A.blocks_per_tile + sb) * 4 + part * WPB + (threadIdx.x >> 6), threadIdx.x & 63, px, py, o))
		return;
	Ray R;
	R.o = o;
	const unsigned long long t_start = A.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
	const bool marched = ray_setup<SKIP>(A, px, py, R);
	uint32_t   iter    = 0;
	if (marched)
	{
		// The frame time is the critical path of the wave with the longest ray: once a wave has run 48 events it is one of
		// those, so let it win instruction arbitration against the younger waves on its SIMD.
		while (!ray_event<SKIP, ERT, GRAD, PACKED>(A, R, s_alpha, s_unorm, s_bits, tf_bits))
			if (__builtin_amdgcn_readfirstlane(++iter) == 48u)        // provably wave-uniform: a real scalar branch
				__builtin_amdgcn_s_setprio(3);
	}
	ray_finish(A, R, marched);
	if (A.trace)
	{        // diagnostic only: per-wave timeline (100 MHz clock); the values never feed an output
		uint32_t it = iter;
		for (int o2 = 32; o2 > 0; o2 >>= 1)
			it = max(it, (uint32_t) __shfl_xor((int) it, o2));
		const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
		if ((threadIdx.x & 63) == __builtin_ctzll(__ballot(1)))
		{
			unsigned long long *rec = A.trace + ((size_t) blockIdx.x * WPB + (threadIdx.x >> 6)) * kTraceWords;
			rec[0] = t_start, rec[1] = t_end, rec[2] = it, rec[3] = ((unsigned long long) __builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | (k * A.blocks_per_tile + sb);
		}
	}
}
// ---------------------------------------------------------------------------------------------------------------
// Wave-private brick cache in LDS (the reference samples through the texture unit, frag:272 / sampler
// src/volume_component.cpp:139-148; here the 256-byte bricks a wave's rays are inside are staged in LDS).
// A per-lane gather costs the CU's texture addresser 16-33 cycles per wave instruction whatever its width
// (tools/micro/gather_mask.hip), four of them per sample; a brick is fetched ONCE by 16 lanes x 16 bytes (four bricks per
// load instruction), serves every footprint of every ray of the wave that falls into it — typically for several
// iterations — and the gathers become ds_reads.  Direct-mapped, tag = brick index, no cross-wave sharing (no barriers).
// ---------------------------------------------------------------------------------------------------------------
template <int NS>
struct BrickCache
{
	uint32_t tags[4][NS];
	__attribute__((aligned(16))) uint8_t data[4][NS * 256];
};

// brick index, byte offset of the footprint's first dword inside the brick, cache slot and the three filter weights
template <int NS>
__device__ __forceinline__ void packed_footprint_ids(int W, int H, int D, int pmx, int pmy, float px, float py, float pz, float &wx, float &wy, float &wz,
                                                     uint32_t &brick, uint32_t &in, uint32_t &slot)
{
	const float cx = __builtin_fmaf(px, (float) W, -0.5f), cy = __builtin_fmaf(py, (float) H, -0.5f), cz = __builtin_fmaf(pz, (float) D, -0.5f);
	const float fx = __builtin_floorf(cx), fy = __builtin_floorf(cy), fz = __builtin_floorf(cz);
	wx = cx - fx, wy = cy - fy, wz = cz - fz;
	const int bx = i_clamp((int) fx, -1, W) + 1, by = i_clamp((int) fy, -1, H) + 1, bz = i_clamp((int) fz, -1, D) + 1;
	const uint32_t macro = ((uint32_t) (bz >> 5) * (uint32_t) pmy + (uint32_t) (by >> 5)) * (uint32_t) pmx + (uint32_t) (bx >> 5);
	const uint32_t sub   = (uint32_t) ((((bz >> 2) & 7) << 6) | (((by >> 2) & 7) << 3) | ((bx >> 2) & 7));
	in    = (uint32_t) ((((bz & 3) * 5 + (by & 3)) * 5 + (bx & 3)) * 2);
	brick = macro * 512u + sub;
	// a wave's rays span about three bricks per axis: 3 x 3 x 3 neighbourhoods map without conflicts
	slot = ((uint32_t) (bx >> 2) + 3u * (uint32_t) (by >> 2) + 9u * (uint32_t) (bz >> 2)) & (uint32_t) (NS - 1);
}

template <int NS>
__device__ __forceinline__ void cached_footprint(const uint8_t *__restrict__ packed, uint8_t *cache, uint32_t *tags, bool want, uint32_t brick, uint32_t slot,
                                                 uint32_t in, uint32_t &q00, uint32_t &q10, uint32_t &q01, uint32_t &q11)
{
	const uint32_t lane    = threadIdx.x & 63u;
	bool           pending = want;
	// every round serves at least the first missing lane, so 64 rounds are enough; the bound only keeps a logic error from hanging the GPU
	for (int round = 0; round < 66; ++round)
	{
		// hits take their four row dwords now: a fill further down may evict the slot (LDS operations of a wave execute in order)
		if (pending && tags[slot] == brick)
		{
			const uint8_t *b = cache + slot * 256u + in;
			q00 = *reinterpret_cast<const u32_align2 *>(b);
			q10 = *reinterpret_cast<const u32_align2 *>(b + 10);
			q01 = *reinterpret_cast<const u32_align2 *>(b + 50);
			q11 = *reinterpret_cast<const u32_align2 *>(b + 60);
			pending = false;
		}
		unsigned long long sel = __ballot(pending);
		if (sel == 0ull)
			break;
		// up to four missing bricks with four different slots, one per group of 16 lanes
		uint32_t fill_brick = 0xffffffffu, fill_slot = 0;
#pragma unroll
		for (uint32_t g = 0; g < 4; ++g)
		{
			if (sel != 0ull)
			{
				const int      leader = __builtin_ctzll(sel);
				const uint32_t b = (uint32_t) __builtin_amdgcn_readlane((int) brick, leader), sl = (uint32_t) __builtin_amdgcn_readlane((int) slot, leader);
				if ((lane >> 4) == g)
					fill_brick = b, fill_slot = sl;
				sel &= ~__ballot(slot == sl);
			}
		}
		if (fill_brick != 0xffffffffu)
		{
			const uint4 v = *reinterpret_cast<const uint4 *>(packed + ((uint64_t) fill_brick << 8) + (lane & 15u) * 16u);
			*reinterpret_cast<uint4 *>(cache + fill_slot * 256u + (lane & 15u) * 16u) = v;
			if ((lane & 15u) == 0u)
				tags[fill_slot] = fill_brick;
		}
	}
}

// What the evaluation of one loop position hands to the replay.
struct Entry
{
	uint32_t cell;        // linear index of the distance-map cell of the position (frag:220-221)
	int      skip;        // probe outcome: 0 = the cell is occupied (dist == 0), else the skip length max(1, ceil(...)) (frag:244-247)
	uint32_t tx;          // sample outcome: separable TF: the alpha byte; generic: the RGBA8 texel (0 when its alpha is 0)
	float    a, c;        // corrected opacity of the sample and (separable TF) its premultiplied grey value
};

// value of `v` in lane KK of this lane's group of W lanes
template <int W, int KK>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v)
{
	if (W == 1)
		return v;
	if (W == 2)
		return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, KK == 0 ? 0xA0 : 0xF5, 0xf, 0xf, true);        // quad_perm [0,0,2,2] / [1,1,3,3]
	if (W == 4)
		return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, KK * 0x55, 0xf, 0xf, true);        // quad_perm [KK,KK,KK,KK]
	return (uint32_t) __shfl((int) v, (int) ((threadIdx.x & 63u & ~(uint32_t) (W - 1)) + KK));
}

// ---------------------------------------------------------------------------------------------------------------
// Evaluate loop position p of ray R: everything the frag's loop body reads from memory at that position, for BOTH kinds
// of event (frag:220-247 probe, frag:266-284 sample).  want_dist / want_sample say which loads can be needed; the other
// kind's loads read a dummy address and its results are never used.
// ---------------------------------------------------------------------------------------------------------------
template <int SKIP, int GRAD, bool PACKED, bool STAMP, bool MASKED, int NS>
__device__ __forceinline__ void er_evaluate(const RayMarchArgs &A, const Ray &R, int p, int k, bool full, bool idle, bool sep, const RmLds &L, uint8_t *cache,
                                            uint32_t *tags, Entry &E, unsigned long long &t_addr, unsigned long long &t_issued, unsigned long long &t_returned)
{
	const int   W = A.W, H = A.H, D = A.D;
	const float fp = (float) p;
	const float posx = __builtin_fmaf(fp, R.sx, R.ex), posy = __builtin_fmaf(fp, R.sy, R.ey), posz = __builtin_fmaf(fp, R.sz, R.ez);
	int         uix = 0, uiy = 0, uiz = 0;
	float       ux = 0, uy = 0, uz = 0;
	uint32_t    cell = 0;
	if (SKIP != VKV_SKIP_NONE)
	{        // frag:192, 220-221
		const float kx = (float) W / A.block_size[0], ky = (float) H / A.block_size[1], kz = (float) D / A.block_size[2];
		ux = kx * posx, uy = ky * posy, uz = kz * posz;
		uix = i_clamp((int) ux, 0, A.mw - 1), uiy = i_clamp((int) uy, 0, A.mh - 1), uiz = i_clamp((int) uz, 0, A.md - 1);
		cell = ((uint32_t) uiz * (uint32_t) A.mh + (uint32_t) uiy) * (uint32_t) A.mw + (uint32_t) uix;
	}
	// the first position of the window is the ray's next event and its kind is known (frag:224); later positions get both
	bool want_dist = SKIP != VKV_SKIP_NONE, want_sample = true;
	if (SKIP != VKV_SKIP_NONE && k == 0 && !full)
	{
		const bool probe0 = !R.occupied && cell != R.ul;
		want_dist = probe0, want_sample = !probe0;
	}
	if (idle)        // the ray of this lane has ended (or never started): it only helps with the brick cache fills
		want_dist = want_sample = false;
	E.cell = cell;

	// ---- issue phase: probe byte and footprint of this position together --------------------------------------------
	constexpr bool kHoist = PACKED && GRAD != 2;
	uint32_t       dist = 0, q00 = 0, q10 = 0, q01 = 0, q11 = 0;
	float          wx = 0, wy = 0, wz = 0;
	if (NS > 0 && kHoist)
	{        // probe byte from memory (masked), footprint through the wave's brick cache
		if (SKIP != VKV_SKIP_NONE && want_dist)
			dist = R.dmap[cell];
		uint32_t brick, in, slot;
		packed_footprint_ids<(NS > 0 ? NS : 1)>(W, H, D, A.pmx, A.pmy, posx, posy, posz, wx, wy, wz, brick, in, slot);
		cached_footprint<(NS > 0 ? NS : 1)>(A.packed, cache, tags, want_sample, brick, slot, in, q00, q10, q01, q11);
	}
	else if (MASKED)
	{        // the texture addresser's time per load grows with the number of active lanes (tools/micro/gather_mask.hip): lanes that
		 // cannot need a kind of load sit it out under EXEC instead of reading a dummy address
		if (SKIP != VKV_SKIP_NONE && want_dist)
			dist = R.dmap[cell];
		if (kHoist && want_sample)
		{
			const uint8_t *ba = packed_footprint(A.packed, W, H, D, A.pmx, A.pmy, posx, posy, posz, wx, wy, wz);
			q00 = *reinterpret_cast<const u32_align2 *>(ba);
			q10 = *reinterpret_cast<const u32_align2 *>(ba + 10);
			q01 = *reinterpret_cast<const u32_align2 *>(ba + 50);
			q11 = *reinterpret_cast<const u32_align2 *>(ba + 60);
		}
	}
	else
	{
		uint32_t       dcell = want_dist ? cell : 0u;
		const uint8_t *ba    = A.packed;
		if (kHoist)
		{
			const uint8_t *fpa = packed_footprint(A.packed, W, H, D, A.pmx, A.pmy, posx, posy, posz, wx, wy, wz);
			ba                 = want_sample ? fpa : A.packed;
		}
		if (STAMP)
		{        // diagnostic build only: all addresses are computed, no load has been issued
			asm volatile("" : "+v"(dcell), "+v"(ba));
			t_addr = __builtin_amdgcn_s_memtime();
		}
		if (SKIP != VKV_SKIP_NONE)
			dist = R.dmap[dcell];
		if (kHoist)
		{
			q00 = *reinterpret_cast<const u32_align2 *>(ba);
			q10 = *reinterpret_cast<const u32_align2 *>(ba + 10);
			q01 = *reinterpret_cast<const u32_align2 *>(ba + 50);
			q11 = *reinterpret_cast<const u32_align2 *>(ba + 60);
		}
	}
	asm volatile("" : "+v"(dist), "+v"(q00), "+v"(q10), "+v"(q01), "+v"(q11));
	if (STAMP)
	{        // diagnostic build only: when were the loads issued, when had they all returned
		t_issued = __builtin_amdgcn_s_memtime();
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		asm volatile("" : "+v"(dist), "+v"(q00), "+v"(q10), "+v"(q01), "+v"(q11));
		t_returned = __builtin_amdgcn_s_memtime();
	}

	// ---- probe outcome (frag:234-247) ---------------------------------------------------------------------------------
	E.skip = 0;
	if (SKIP != VKV_SKIP_NONE)
	{
		const float rx = __builtin_amdgcn_fmed3f((float) uix - ux, -1.0f, 0.0f);
		const float ry = __builtin_amdgcn_fmed3f((float) uiy - uy, -1.0f, 0.0f);
		const float rz = __builtin_amdgcn_fmed3f((float) uiz - uz, -1.0f, 0.0f);
		float       ax, ay, az;
		if (SKIP == VKV_SKIP_BLOCK)
		{
			ax = (((R.six < 0.0f) ? 0.0f : 1.0f) + rx) * R.six;
			ay = (((R.siy < 0.0f) ? 0.0f : 1.0f) + ry) * R.siy;
			az = (((R.siz < 0.0f) ? 0.0f : 1.0f) + rz) * R.siz;
		}
		else
		{
			const float fd = (float) dist;
			ax = (((R.six > 0.0f) ? fd : 1.0f - fd) + rx) * R.six;
			ay = (((R.siy > 0.0f) ? fd : 1.0f - fd) + ry) * R.siy;
			az = (((R.siz > 0.0f) ? fd : 1.0f - fd) + rz) * R.siz;
		}
		// a NaN component (0 * inf on an axis-parallel ray) counts as +inf: minNum ignores it; all three cannot be NaN, and if they
		// were the comparison below caps the result exactly as the select chain of the oracle does
		float m = __builtin_fminf(__builtin_fminf(ax, ay), az);
		m       = (m < 1073741824.0f) ? m : 1073741824.0f;
		E.skip  = dist > 0u ? max(1, (int) __builtin_ceilf(m)) : 0;
	}

	// ---- sample outcome (frag:272-284) -------------------------------------------------------------------------------
	float intensity = 0.0f, gradient = 1.0f;
	if (kHoist)
	{
		float unused;
		if (GRAD == 1)
			packed_filter<true>(q00, q10, q01, q11, wx, wy, wz, intensity, gradient);
		else
			packed_filter<false>(q00, q10, q01, q11, wx, wy, wz, intensity, unused);
	}
	else if (want_sample)
	{
		float unused;
		if (PACKED)
		{
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
	// get_color (transfer_function.glsl:35-38): NEAREST texel
	const uint32_t ti = (uint32_t) tf_texel(intensity), tg = (uint32_t) tf_texel(gradient);
	if (sep)
	{
		const uint32_t ab = tf_separable_alpha(L.s.ai[ti], L.s.ag[tg]);
		const float2   pr = L.s.pair[ab];
		E.tx = ab, E.a = pr.x, E.c = pr.y;
	}
	else
	{
		const uint32_t tidx  = tg * 256u + ti;
		uint32_t       texel = 0;
		if (A.tf_bits)
		{
			if (want_sample && ((L.g.bits[tidx >> 5] >> (tidx & 31u)) & 1u))
				texel = reinterpret_cast<const uint32_t *>(A.tf)[tidx];
		}
		else if (want_sample)
			texel = reinterpret_cast<const uint32_t *>(A.tf)[tidx];
		texel = (texel >> 24) ? texel : 0u;
		E.tx = texel, E.a = L.g.alpha[texel >> 24], E.c = 0.0f;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Replay: one step of the frag's state machine (frag:215-312) for loop position q with the evaluated entry e.
// Nothing happens unless the ray's loop index IS q.  can_fuse: e holds the sample of q as well as the probe, so a probe
// that finds the cell occupied and steps back onto q itself (frag:253-261) goes on to take that sample in the same step.
// ---------------------------------------------------------------------------------------------------------------
template <int SKIP, bool ERT>
__device__ __forceinline__ void er_replay(const RayMarchArgs &A, Ray &R, float &grey, bool &done, const Entry &e, int q, bool can_fuse, bool sep, const RmLds &L)
{
	// Straight-line, predicated: the lanes of a wave disagree about almost every condition below, nested branches would
	// only serialise short pieces of arithmetic behind EXEC-mask bookkeeping.
	const bool act    = !done && R.i == q;
	const bool probe  = SKIP != VKV_SKIP_NONE && act && !R.occupied && e.cell != R.ul;        // frag:224
	const bool p_skip = probe && e.skip > 0;                                                     // frag:236-247
	const bool p_occ  = probe && e.skip == 0;                                                    // frag:248-262
	const int  jb     = max(q - A.back, R.i_min);
	const bool fuse   = p_occ && can_fuse && jb == q;
	const bool smp    = (act && !probe) || fuse;                                                 // frag:266-310
	const uint32_t ab = sep ? e.tx : (e.tx >> 24);
	const bool occ_s  = ab > 0u;                                                                 // frag:276
	const bool hit    = smp && occ_s;
	R.n_dist += probe ? 1u : 0u;
	R.n_vol += smp ? 1u : 0u;
	R.n_empty += (smp && !occ_s) ? 1u : 0u;
	// frag:283-299
	const float a  = e.a;
	const float om = 1.0f - R.a;
	const float na = __builtin_fmaf(om, a, R.a);
	if (sep)
		grey = hit ? __builtin_fmaf(om, e.c, grey) : grey;        // r = g = b: one channel is tracked
	else
	{
		const float r_ = L.g.unorm[e.tx & 255u] * a, g_ = L.g.unorm[(e.tx >> 8) & 255u] * a, b_ = L.g.unorm[(e.tx >> 16) & 255u] * a;
		const float nr = __builtin_fmaf(om, r_, R.r), ng = __builtin_fmaf(om, g_, R.g), nb = __builtin_fmaf(om, b_, R.b);
		R.r = hit ? nr : R.r, R.g = hit ? ng : R.g, R.b = hit ? nb : R.b;
	}
	const bool ended = ERT && hit && na > 0.99f;
	R.a              = hit ? (ended ? 1.0f : na) : R.a;
	R.first_hit      = (hit && a > 0.0f) ? q : R.first_hit;
	// state
	R.occupied = smp ? occ_s : (p_occ ? true : R.occupied);
	if (SKIP != VKV_SKIP_NONE)
		R.ul = (hit || p_occ) ? e.cell : R.ul;
	const bool adv = smp && !ended;
	int        ni  = R.i;
	ni             = p_skip ? q + e.skip : ni;
	ni             = (p_occ && !fuse) ? jb : ni;
	ni             = adv ? q + 1 : ni;
	R.i_min        = adv ? q + 1 : R.i_min;
	R.i            = ni;
	done           = done || ended || (act && ni >= R.n_steps);
}

template <int SKIP, bool ERT, int W, int KK>
__device__ __forceinline__ void er_replay_all(const RayMarchArgs &A, Ray &R, float &grey, bool &done, const Entry &mine, int j0, bool full, bool sep, const RmLds &L)
{
	if constexpr (KK < W)
	{
		Entry e;
		e.cell = group_bcast<W, KK>(mine.cell);
		e.skip = (int) group_bcast<W, KK>((uint32_t) mine.skip);
		e.tx   = group_bcast<W, KK>(mine.tx);
		e.a    = __uint_as_float(group_bcast<W, KK>(__float_as_uint(mine.a)));
		e.c    = __uint_as_float(group_bcast<W, KK>(__float_as_uint(mine.c)));
		er_replay<SKIP, ERT>(A, R, grey, done, e, j0 + KK, full || KK > 0, sep, L);
		er_replay_all<SKIP, ERT, W, KK + 1>(A, R, grey, done, mine, j0, full, sep, L);
	}
}
// FLAGS
constexpr uint32_t kErFull  = 1u;        // every lane loads probe byte AND footprint (the first position of a window too)
constexpr uint32_t kErStamp = 2u;        // diagnostic build: per-wave phase times into the trace buffer
constexpr uint32_t kErMasked = 4u;       // loads a lane cannot need are masked off instead of reading a dummy address
constexpr uint32_t kErCache16 = 8u, kErCache32 = 16u;        // wave-private LDS brick cache with 16 / 32 slots

// The march of one wave: evaluate + replay until every ray of the wave has ended.  `sep` arrives as a literal so each copy of the
// loop holds one transfer-function path only.
template <int SKIP, bool ERT, int GRAD, bool PACKED, int W, uint32_t FLAGS, int NS>
__device__ __forceinline__ void er_march(const RayMarchArgs &A, Ray &R, bool marched, const bool sep, const RmLds &L, uint8_t *cache_data, uint32_t *cache_tags,
                                         uint32_t lane, uint32_t &iter, uint32_t &pha, uint32_t &ph0, uint32_t &ph1, uint32_t &ph2, uint32_t &ph3)
{
	constexpr bool kFull = (FLAGS & kErFull) != 0, kStamp = (FLAGS & kErStamp) != 0;
	{
		const int kk   = (int) (lane % W);
		bool      done = !marched;
		float     grey = 0.0f;        // separable greyscale TF: the one colour channel
		if (done)
			R.i = 0, R.n_steps = 0, R.occupied = true, R.sx = R.sy = R.sz = R.ex = R.ey = R.ez = 0.0f, R.six = R.siy = R.siz = 1.0f, R.dmap = A.maps[0];
		while (__ballot(!done) != 0ull)        // wave-uniform loop
		{
			unsigned long long t0 = 0, ta = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
			if (kStamp)
				t0 = __builtin_amdgcn_s_memtime();
			const int j0 = R.i;
			Entry     E;
			er_evaluate<SKIP, GRAD, PACKED, kStamp, (FLAGS & kErMasked) != 0, NS>(A, R, j0 + kk, kk, kFull, done, sep, L, cache_data, cache_tags, E, ta, t1, t2);
			if (kStamp)
			{
				asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
				asm volatile("" : "+v"(E.cell), "+v"(E.skip), "+v"(E.tx), "+v"(E.a), "+v"(E.c));
				t3 = __builtin_amdgcn_s_memtime();
			}
			er_replay_all<SKIP, ERT, W, 0>(A, R, grey, done, E, j0, kFull, sep, L);
			if (kStamp)
			{
				asm volatile("" : "+v"(R.i), "+v"(R.a));
				t4 = __builtin_amdgcn_s_memtime();
				// top -> addresses ready -> loads issued -> loads returned -> entry ready -> replay done
				if (ta == 0)
					ta = t0;
				pha += (uint32_t) (ta - t0), ph0 += (uint32_t) (t1 - ta), ph1 += (uint32_t) (t2 - t1), ph2 += (uint32_t) (t3 - t2), ph3 += (uint32_t) (t4 - t3);
			}
			// the frame time is the critical path of the wave with the longest ray: once a wave has run 48 iterations it is one of
			// those, so let it win instruction arbitration against the younger waves on its SIMD
			if (++iter == 48u)
				__builtin_amdgcn_s_setprio(3);
		}
		if (sep && marched)
			R.r = grey, R.g = grey, R.b = grey;
	}
}

template <int SKIP, bool ERT, int GRAD, bool PACKED, int W, uint32_t FLAGS>
__global__ void __launch_bounds__(256) k_raymarch_er(const RayMarchArgs A)
{
	constexpr int NS = (FLAGS & kErCache32) ? 32 : ((FLAGS & kErCache16) ? 16 : 0);
	__shared__ RmLds                         L;
	__shared__ BrickCache<(NS > 0 ? NS : 1)> BC;
	if (NS > 0)
		for (int i = threadIdx.x; i < 4 * NS; i += blockDim.x)
			BC.tags[i / NS][i % NS] = 0xffffffffu;
	const bool     sep = stage_tables_er(A, L);
	// workgroup -> (tile of the schedule, 16x16 block of the tile, part of the block): XCD x = id & 7 marches the tiles
	// x, x + 8, ... (see k_raymarch_tiles); a block takes W workgroups of four waves
	const uint32_t x = blockIdx.x & 7u, idx = blockIdx.x >> 3;
	const uint32_t k = (idx / (A.blocks_per_tile * W)) * 8u + x, sbp = idx % (A.blocks_per_tile * W);
	const uint32_t sb = sbp / W, part = sbp % W;
	if (k >= A.tile_count)
		return;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t       px, py, o;
	// no lane leaves before the end: the brick cache is filled by all 64 lanes of a wave whatever their rays are doing
	const bool inside = block_pixel<W>(A, k * A.blocks_per_tile + sb, (part * 4u + wave) * (64u / W) + lane / W, px, py, o);
	Ray        R;
	R.o = o;
	const unsigned long long t_start = A.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
	bool       marched = false;
	if (inside)
		marched = ray_setup<SKIP>(A, px, py, R);
	uint32_t   iter    = 0;
	uint32_t   pha = 0, ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0;
	if (sep)
		er_march<SKIP, ERT, GRAD, PACKED, W, FLAGS, NS>(A, R, marched, true, L, BC.data[wave], BC.tags[wave], lane, iter, pha, ph0, ph1, ph2, ph3);
	else
		er_march<SKIP, ERT, GRAD, PACKED, W, FLAGS, NS>(A, R, marched, false, L, BC.data[wave], BC.tags[wave], lane, iter, pha, ph0, ph1, ph2, ph3);
	if (inside)
		ray_finish(A, R, marched);
	if (A.trace)
	{        // diagnostic only: per-wave timeline (100 MHz clock); the values never feed an output
		uint32_t it = iter;
		for (int o2 = 32; o2 > 0; o2 >>= 1)
		{        // the lane that marched longest carries the wave's sums
			it  = max(it, (uint32_t) __shfl_xor((int) it, o2));
			ph0 = max(ph0, (uint32_t) __shfl_xor((int) ph0, o2)), ph1 = max(ph1, (uint32_t) __shfl_xor((int) ph1, o2));
			ph2 = max(ph2, (uint32_t) __shfl_xor((int) ph2, o2)), ph3 = max(ph3, (uint32_t) __shfl_xor((int) ph3, o2));
			pha = max(pha, (uint32_t) __shfl_xor((int) pha, o2));
		}
		const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
		if (lane == (uint32_t) __builtin_ctzll(__ballot(1)))
		{
			unsigned long long *rec = A.trace + ((size_t) blockIdx.x * 4 + wave) * kTraceWords;
			rec[0] = t_start, rec[1] = t_end, rec[2] = it, rec[3] = ((unsigned long long) __builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | (k * A.blocks_per_tile + sb);
			rec[4] = ph0, rec[5] = ph1, rec[6] = ph2, rec[7] = ph3, rec[8] = pha, rec[9] = 0;
		}
	}
}
// the gradient channel alone (bytes 1 and 3 of the x-pair dwords)
template <bool TABLE = false>
__device__ __forceinline__ void packed_filter_g(uint32_t q00, uint32_t q10, uint32_t q01, uint32_t q11, float wx, float wy, float wz, float &out_g)
{
	constexpr float kScale = TABLE ? kInv255 * 1024.0f : kInv255;
	const float b000 = cvt_ubyte1(q00), b100 = cvt_ubyte3(q00), b010 = cvt_ubyte1(q10), b110 = cvt_ubyte3(q10);
	const float b001 = cvt_ubyte1(q01), b101 = cvt_ubyte3(q01), b011 = cvt_ubyte1(q11), b111 = cvt_ubyte3(q11);
	const float c00 = __builtin_fmaf(wx, b100 - b000, b000), c10 = __builtin_fmaf(wx, b110 - b010, b010);
	const float c01 = __builtin_fmaf(wx, b101 - b001, b001), c11 = __builtin_fmaf(wx, b111 - b011, b011);
	const float c0 = __builtin_fmaf(wy, c10 - c00, c00), c1 = __builtin_fmaf(wy, c11 - c01, c01);
	out_g = __builtin_fmaf(wz, c1 - c0, c0) * kScale;
}

// ---- kLabFmt: rows as f16 quadruples (v0, g0, v1, g1) from a buffer FORMAT load ---------------------------------------------------
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef int      int4v __attribute__((ext_vector_type(4)));
__device__ half4v vkv_buffer_load_format_h4(int4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f16");

// buffer resource of the packed image: raw (stride 0), 4 GiB window, dst_sel RGBA, USCALED 8_8_8_8 (gfx9 V# word 3)
__device__ __forceinline__ int4v packed_rows_rsrc(const uint8_t *base)
{
	const uint64_t a = reinterpret_cast<uint64_t>(base);
	int4v          r;
	r.x = __builtin_amdgcn_readfirstlane((int) (uint32_t) a);
	r.y = __builtin_amdgcn_readfirstlane((int) (uint32_t) ((a >> 32) & 0xffffu));
	r.z = -1;
	r.w = (int) (0xFACu | (2u << 12) | (10u << 15));
	return r;
}

// One channel pair of the x stage: d = (v1 - v0, g1 - g0) in f16 (integers below 256: exact), then c = fma(wx, d, b) with the f16
// operands widened inside the instruction - the same real numbers the fp32 path multiplies and adds, rounded once: bit-identical.
__device__ __forceinline__ float fma_mix_lo(float w, half2v d, half2v b)
{        // fma(w, float(d.x), float(b.x)); the compiler forms it from the plain expression for the low halves only
	float r;
	asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(w), "v"(d), "v"(b));
	return r;
}
__device__ __forceinline__ float fma_mix_hi(float w, half2v d, half2v b)
{        // fma(w, float(d.y), float(b.y))
	float r;
	asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(w), "v"(d), "v"(b));
	return r;
}

template <bool WANT_V, bool WANT_G, bool TABLE>
__device__ __forceinline__ void packed_filter_fmt(half4v h00, half4v h10, half4v h01, half4v h11, float wx, float wy, float wz, float &out_v, float &out_g)
{
	constexpr float kScale = TABLE ? kInv255 * 1024.0f : kInv255;
	const half2v    b00 = {h00.x, h00.y}, b10 = {h10.x, h10.y}, b01 = {h01.x, h01.y}, b11 = {h11.x, h11.y};
	const half2v    d00 = half2v{h00.z, h00.w} - b00, d10 = half2v{h10.z, h10.w} - b10, d01 = half2v{h01.z, h01.w} - b01, d11 = half2v{h11.z, h11.w} - b11;
	if (WANT_V)
	{
		const float c00 = fma_mix_lo(wx, d00, b00), c10 = fma_mix_lo(wx, d10, b10), c01 = fma_mix_lo(wx, d01, b01), c11 = fma_mix_lo(wx, d11, b11);
		const float c0 = __builtin_fmaf(wy, c10 - c00, c00), c1 = __builtin_fmaf(wy, c11 - c01, c01);
		out_v = __builtin_fmaf(wz, c1 - c0, c0) * kScale;
	}
	if (WANT_G)
	{
		const float c00 = fma_mix_hi(wx, d00, b00), c10 = fma_mix_hi(wx, d10, b10), c01 = fma_mix_hi(wx, d01, b01), c11 = fma_mix_hi(wx, d11, b11);
		const float c0 = __builtin_fmaf(wy, c10 - c00, c00), c1 = __builtin_fmaf(wy, c11 - c01, c01);
		out_g = __builtin_fmaf(wz, c1 - c0, c0) * kScale;
	}
}
// the same footprint as a 32-bit byte offset into the packed image (kLabFmt: images below 4 GiB)
__device__ __forceinline__ uint32_t packed_footprint_full_offset(const FullLutConsts &C, const RmLds &L, float px, float py, float pz, float &wx, float &wy, float &wz)
{
	const float cx = __builtin_fmaf(px, C.w, -0.5f), cy = __builtin_fmaf(py, C.h, -0.5f), cz = __builtin_fmaf(pz, C.d, -0.5f);
	const float fx = __builtin_floorf(cx), fy = __builtin_floorf(cy), fz = __builtin_floorf(cz);
	wx = cx - fx, wy = cy - fy, wz = cz - fz;
	const int      tx = (int) __builtin_fmaf(__builtin_amdgcn_fmed3f(fx, -1.0f, C.w), 4.0f, 4.0f);
	const int      ty = (int) __builtin_fmaf(__builtin_amdgcn_fmed3f(fy, -1.0f, C.h), 4.0f, C.oy);
	const int      tz = (int) __builtin_fmaf(__builtin_amdgcn_fmed3f(fz, -1.0f, C.d), 4.0f, C.oz);
	const char *   lut = reinterpret_cast<const char *>(full_lut_base(L));
	const uint32_t xo = *reinterpret_cast<const uint32_t *>(lut + tx), yo = *reinterpret_cast<const uint32_t *>(lut + ty);
	const uint32_t zo = *reinterpret_cast<const uint32_t *>(lut + tz);
	return ((xo + yo) + zo) << 1;
}

template <int SKIP, bool ERT, int GRAD, bool PACKED, bool SEP, uint32_t LF>
__device__ __forceinline__ void lab_lean_march(const RayMarchArgs &A, Ray &R, const RmLds &L, uint32_t &iter, LeanStamp &stamp)
{
	constexpr bool kStamp = (LF & kLabStamp) != 0, kCounts = (LF & kLabNoCounts) == 0;
	uint32_t       stamp_prev = 0, stamp_kind = 3;
	constexpr bool kHoist = PACKED && GRAD != 2;
	constexpr bool kUni = (LF & kLabUniform) != 0, kNt = (LF & kLabNt) != 0, kLut = (LF & kLabLut) != 0, kBranch = (LF & kLabBranch) != 0,
	               kCvt = (LF & kLabCvt) != 0, kNest = (LF & kLabNest) != 0 && kBranch, kKeep = (LF & kLabKeep) != 0, kScalar = (LF & kLabScalar) != 0, kFull = (LF & kLabFull) != 0 && SEP, kTf = (LF & kLabTf) != 0 && SEP && kHoist && kCvt,
	               kGradSkip = (LF & kLabGradSkip) != 0 && kTf && GRAD == 1, kFloatI = (LF & kLabFloatI) != 0 && kBranch, kWb = (LF & kLabWb) != 0,
	               kFloatCell = (LF & kLabFloatCell) != 0 && SKIP != VKV_SKIP_NONE, kPrefetch = (LF & kLabPrefetch) != 0 && kHoist && kNest,
	               kFmt = (LF & kLabFmt) != 0 && kFull && kTf && !kPrefetch, kFmtScalar = kFmt && (LF & kLabFmtVec) == 0,
	               kBrick = (LF & kLabBrickMap) != 0 && kScalar && !kFloatCell, kProbeOnly = (LF & kLabProbeOnly) != 0, kHalfRows = (LF & kLabHalfRows) != 0;
	// bricked map: index = x + 4 y + 16 z + 60 (x >> 2) + (64 bw - 16) (y >> 2) + (64 bw bh - 64) (z >> 2), bw / bh = bricks per row / column
	const uint32_t brick_cy = 64u * (((uint32_t) A.mw + 3u) >> 2) - 16u, brick_cz = 64u * (((uint32_t) A.mw + 3u) >> 2) * (((uint32_t) A.mh + 3u) >> 2) - 64u;
	using idx_t = std::conditional_t<kFloatI, float, int>;
	const int   W = A.W, H = A.H, D = A.D;
	float kx = SKIP != VKV_SKIP_NONE ? (float) W / A.block_size[0] : 0.0f, ky = SKIP != VKV_SKIP_NONE ? (float) H / A.block_size[1] : 0.0f,
	      kz = SKIP != VKV_SKIP_NONE ? (float) D / A.block_size[2] : 0.0f;
	if (kFmtScalar && SKIP != VKV_SKIP_NONE)
		kx = uniform_f32(kx), ky = uniform_f32(ky), kz = uniform_f32(kz);
	const int   mw1 = A.mw - 1, mh1 = A.mh - 1, md1 = A.md - 1;
	const float fmw1 = (float) mw1, fmh1 = (float) mh1, fmd1 = (float) md1, fmw = (float) A.mw, fmh = (float) A.mh;
	float    grey = 0.0f;
	uint32_t ul   = 0;
	bool     occ  = true, done = false;
	const float sgx = R.six > 0.0f ? 1.0f : -1.0f, sgy = R.siy > 0.0f ? 1.0f : -1.0f, sgz = R.siz > 0.0f ? 1.0f : -1.0f;
	const float ofx = R.six > 0.0f ? 0.0f : 1.0f, ofy = R.siy > 0.0f ? 0.0f : 1.0f, ofz = R.siz > 0.0f ? 0.0f : 1.0f;
	FullLutConsts fullc = {};
	if (kFull)
		fullc = full_lut_consts<kFmtScalar>(A);
	int4v rows = {};
	if (kFmt)
		rows = packed_rows_rsrc(A.packed);
	idx_t       li = (idx_t) R.i, li_min = (idx_t) R.i_min, lfirst = (idx_t) R.first_hit;
	const idx_t ln = (idx_t) R.n_steps;
	idx_t       lback = (idx_t) A.back;
	if constexpr (kFloatI && kFmtScalar)
		lback = uniform_f32(lback);
	uint32_t pq00 = 0, pq10 = 0, pq01 = 0, pq11 = 0;        // kPrefetch: the footprint of position pf_i, requested one iteration ahead
	float    pwx = 0, pwy = 0, pwz = 0;
	idx_t    pf_i = (idx_t) -1;
	while (!done)
	{
		const idx_t i  = li;
		const float fi = (float) i;
		const float posx = __builtin_fmaf(fi, R.sx, R.ex), posy = __builtin_fmaf(fi, R.sy, R.ey), posz = __builtin_fmaf(fi, R.sz, R.ez);
		int         uix = 0, uiy = 0, uiz = 0;
		float       ux = 0, uy = 0, uz = 0, fuix = 0, fuiy = 0, fuiz = 0;
		uint32_t    cell = 0;
		if (SKIP != VKV_SKIP_NONE)
		{        // frag:192, 220-221
			ux = kx * posx, uy = ky * posy, uz = kz * posz;
			if (kFloatCell)
			{        // clamp(trunc(u), 0, m - 1) == clamp(floor(u), 0, m - 1): they differ only for u in (-1, 0), where both clamp to 0
				fuix = __builtin_amdgcn_fmed3f(__builtin_floorf(ux), 0.0f, fmw1), fuiy = __builtin_amdgcn_fmed3f(__builtin_floorf(uy), 0.0f, fmh1);
				fuiz = __builtin_amdgcn_fmed3f(__builtin_floorf(uz), 0.0f, fmd1);
				cell = (uint32_t) __builtin_fmaf(__builtin_fmaf(fuiz, fmh, fuiy), fmw, fuix);        // integers below 2^24: exact
			}
			else if (kScalar)
			{
				uix = clamp0_i32((int) ux, mw1), uiy = clamp0_i32((int) uy, mh1), uiz = clamp0_i32((int) uz, md1);
				if (kBrick)
				{
					const uint32_t low = (((uint32_t) uiz << 4) + ((uint32_t) uiy << 2)) + (uint32_t) uix;
					cell = mad_u24((uint32_t) uiz >> 2, brick_cz, mad_u24((uint32_t) uiy >> 2, brick_cy, mad_u24((uint32_t) uix >> 2, 60u, low)));
				}
				else
					cell = mad_u24(mad_u24((uint32_t) uiz, (uint32_t) A.mh, (uint32_t) uiy), (uint32_t) A.mw, (uint32_t) uix);
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
		uint32_t dist = 0, q00 = 0, q10 = 0, q01 = 0, q11 = 0;
		half4v   h00 = {}, h10 = {}, h01 = {}, h11 = {};
		if (kFmt)
			h00 = undefined_value<half4v>(), h10 = undefined_value<half4v>(), h01 = undefined_value<half4v>(), h11 = undefined_value<half4v>();
		float    wx = 0, wy = 0, wz = 0;
		if (kNest)
		{        // every use sits under the predicate of its load: the values of the other lanes are left undefined (no moves)
			dist = undefined_value<uint32_t>(), q00 = undefined_value<uint32_t>(), q10 = undefined_value<uint32_t>(), q01 = undefined_value<uint32_t>();
			q11 = undefined_value<uint32_t>(), wx = undefined_value<float>(), wy = undefined_value<float>(), wz = undefined_value<float>();
		}
		if (SKIP != VKV_SKIP_NONE && probe)
			dist = load_u8_global(R.dmap, cell);
		auto footprint_of = [&](float px_, float py_, float pz_, float &ox, float &oy, float &oz) {
			return kFull ? packed_footprint_full(fullc, L, px_, py_, pz_, ox, oy, oz)
			       : kLut ? packed_footprint_lut<kScalar>(A, px_, py_, pz_, ox, oy, oz)
			              : packed_footprint(A.packed, W, H, D, A.pmx, A.pmy, px_, py_, pz_, ox, oy, oz);
		};
		if (kHoist && !probe)
		{
			if (kFmt)
			{
				const int bo = (int) packed_footprint_full_offset(fullc, L, posx, posy, posz, wx, wy, wz);
				h00 = vkv_buffer_load_format_h4(rows, bo, 0, 0);
				h10 = vkv_buffer_load_format_h4(rows, bo + 10, 0, 0);
				h01 = vkv_buffer_load_format_h4(rows, bo + 50, 0, 0);
				h11 = vkv_buffer_load_format_h4(rows, bo + 60, 0, 0);
			}
			else if (kPrefetch && pf_i == i)
				q00 = pq00, q10 = pq10, q01 = pq01, q11 = pq11, wx = pwx, wy = pwy, wz = pwz;        // requested an iteration ago
			else if (kProbeOnly)
			{
				(void) footprint_of(posx, posy, posz, wx, wy, wz);
				q00 = q10 = q01 = q11 = 0u;
			}
			else if (kHalfRows)
			{
				const uint8_t *ba = footprint_of(posx, posy, posz, wx, wy, wz);
				q00 = load_row<kNt>(ba);
				q01 = load_row<kNt>(ba + 50);
				q10 = q00, q11 = q01;
			}
			else
			{
				const uint8_t *ba = footprint_of(posx, posy, posz, wx, wy, wz);
				q00 = load_row<kNt>(ba);
				q10 = load_row<kNt>(ba + 10);
				q01 = load_row<kNt>(ba + 50);
				q11 = load_row<kNt>(ba + 60);
			}
		}
		if (kPrefetch)
		{
			if (!probe && occ && i + (idx_t) 1 < ln)
			{
				const float    fn = (float) (i + (idx_t) 1);
				const uint8_t *bn = footprint_of(__builtin_fmaf(fn, R.sx, R.ex), __builtin_fmaf(fn, R.sy, R.ey), __builtin_fmaf(fn, R.sz, R.ez), pwx, pwy, pwz);
				pq00 = load_row<kNt>(bn), pq10 = load_row<kNt>(bn + 10), pq01 = load_row<kNt>(bn + 50), pq11 = load_row<kNt>(bn + 60);
				pf_i = i + (idx_t) 1;
			}
			else
				pf_i = (idx_t) -1;
		}
		// a wave whose live lanes all probe (the empty space in front of the volume) or all sample skips the other kind's arithmetic
		const bool any_probe = !kUni || kNest || __builtin_amdgcn_ballot_w64(probe) != 0ull, any_sample = !kUni || kNest || SKIP == VKV_SKIP_NONE || __builtin_amdgcn_ballot_w64(!probe) != 0ull;

		// ---- probe outcome (frag:234-247); needs the probe byte only ---------------------------------------------------
		idx_t skip = 0;
		auto  probe_outcome = [&]() {
			const float rx = __builtin_amdgcn_fmed3f((kFloatCell ? fuix : (float) uix) - ux, -1.0f, 0.0f);
			const float ry = __builtin_amdgcn_fmed3f((kFloatCell ? fuiy : (float) uiy) - uy, -1.0f, 0.0f);
			const float rz = __builtin_amdgcn_fmed3f((kFloatCell ? fuiz : (float) uiz) - uz, -1.0f, 0.0f);
			float       ax, ay, az;
			if (SKIP == VKV_SKIP_BLOCK)
			{
				ax = (((R.six < 0.0f) ? 0.0f : 1.0f) + rx) * R.six;
				ay = (((R.siy < 0.0f) ? 0.0f : 1.0f) + ry) * R.siy;
				az = (((R.siz < 0.0f) ? 0.0f : 1.0f) + rz) * R.siz;
			}
			else if (kCvt)
			{        // step(0, -s) + sign(s) * dist = fd for s > 0, 1 - fd for s < 0: one fma with per-ray constants (exact: the product is exact)
				const float fd = (float) dist;
				ax = (__builtin_fmaf(sgx, fd, ofx) + rx) * R.six;
				ay = (__builtin_fmaf(sgy, fd, ofy) + ry) * R.siy;
				az = (__builtin_fmaf(sgz, fd, ofz) + rz) * R.siz;
			}
			else
			{
				const float fd = (float) dist;
				ax = (((R.six > 0.0f) ? fd : 1.0f - fd) + rx) * R.six;
				ay = (((R.siy > 0.0f) ? fd : 1.0f - fd) + ry) * R.siy;
				az = (((R.siz > 0.0f) ? fd : 1.0f - fd) + rz) * R.siz;
			}
			// a NaN component (0 * inf on an axis-parallel ray) counts as +inf: minNum ignores it; all three cannot be NaN, and if
			// they were the comparison caps the result exactly as the select chain of the oracle does
			float m = __builtin_fminf(__builtin_fminf(ax, ay), az);
			m       = (m < 1073741824.0f) ? m : 1073741824.0f;
			if (kFloatI)
				skip = (idx_t) __builtin_fmaxf(1.0f, __builtin_ceilf(m));        // a NaN m gives 1 here as (int) NaN = 0 does below
			else
				skip = (idx_t) max(1, (int) __builtin_ceilf(m));
		};
		if (SKIP != VKV_SKIP_NONE && !kNest && any_probe)
			probe_outcome();

		// ---- sample outcome (frag:272-284) ---------------------------------------------------------------------------
		float    intensity = 0.0f, gradient = 1.0f;
		uint32_t ab = 0, texel = 0;
		float    a = 0.0f, c = 0.0f;
		auto sample_outcome = [&]() {
		if (kTf)
		{        // separable transfer function, table addresses straight from the filtered values
			const char *ai_tab = reinterpret_cast<const char *>(L.s.ai), *ag_tab = reinterpret_cast<const char *>(L.s.ag);
			float       g_unused;
			if (kFmt)
			{
				if (GRAD == 1 && !kGradSkip)
					packed_filter_fmt<true, true, true>(h00, h10, h01, h11, wx, wy, wz, intensity, gradient);
				else
					packed_filter_fmt<true, false, true>(h00, h10, h01, h11, wx, wy, wz, intensity, g_unused);
			}
			else if (GRAD == 1 && !kGradSkip)
				packed_filter_cvt<true, true>(q00, q10, q01, q11, wx, wy, wz, intensity, gradient);
			else
				packed_filter_cvt<false, true>(q00, q10, q01, q11, wx, wy, wz, intensity, g_unused);
			const float ai = *reinterpret_cast<const float *>(ai_tab + ((uint32_t) (int) intensity & ~3u));        // intensity, gradient: * 1024 here
			float       ag = GRAD == 0 ? L.s.ag[255] : 0.0f;
			// most samples behind a probe are still empty voxels (intensity below the window): when that holds for every sampling lane of
			// the wave the gradient channel is not needed (ai == 0 makes the alpha byte 0 whatever ag is)
			if (GRAD == 1 && (!kGradSkip || __builtin_amdgcn_ballot_w64(ai > 0.0f) != 0ull))
			{
				if (kGradSkip && kFmt)
					packed_filter_fmt<false, true, true>(h00, h10, h01, h11, wx, wy, wz, g_unused, gradient);
				else if (kGradSkip)
					packed_filter_g<true>(q00, q10, q01, q11, wx, wy, wz, gradient);
				ag = *reinterpret_cast<const float *>(ag_tab + ((uint32_t) (int) gradient & ~3u));
			}
			ab              = (uint32_t) ((ai * ag) * 255.0f);        // <= 255: ai, ag <= 1 (k_tf_tables_init)
			const float2 pr = L.s.pair[ab];
			a = pr.x, c = pr.y;
			return;
		}
		if (kHoist && kCvt)
		{
			float unused;
			if (GRAD == 1)
				packed_filter_cvt<true>(q00, q10, q01, q11, wx, wy, wz, intensity, gradient);
			else
				packed_filter_cvt<false>(q00, q10, q01, q11, wx, wy, wz, intensity, unused);
		}
		else if (kHoist)
		{
			float unused;
			if (GRAD == 1)
				packed_filter<true>(q00, q10, q01, q11, wx, wy, wz, intensity, gradient);
			else
				packed_filter<false>(q00, q10, q01, q11, wx, wy, wz, intensity, unused);
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
		if (!kNest && any_sample)
			sample_outcome();

		if (kBranch)
		{        // ---- the frag's state update (frag:224-310) under EXEC: plain moves and adds instead of selects ------------------
			bool probe_now = probe;
			if (kNest && kKeep && kWb)
				__builtin_amdgcn_wave_barrier();        // emits nothing; keeps the load blocks above apart from the blocks below
			else if (kNest && kKeep)
			{        // the same predicate through a register the compiler cannot see through: otherwise it merges these blocks with the
				 // load blocks above and the footprint is only requested after the probe outcome has been worked out
				int pi = probe ? 1 : 0;
				asm volatile("" : "+v"(pi));
				probe_now = pi != 0;
			}
			if (probe_now)
			{
				if (kNest)
					probe_outcome();        // an empty side is skipped by the branch the compiler puts around it (s_cbranch_execz)
				if (kCounts)
					++R.n_dist;
				// frag:244-247 (dist > 0: skip) and frag:253-261 (occupied cell: step back).  The step-back side is three instructions: as
				// selects next to the skip side's EXEC-masked block, not as a block of its own (one region and one branch less per probe)
				const idx_t back_to = kFloatI ? (idx_t) max_f32_raw((float) (i - lback), (float) li_min) : (idx_t) max((int) (i - lback), (int) li_min);
				const bool  hit     = dist == 0u;
				li                  = back_to;
				if (!hit)
					li = i + skip;
				occ = hit ? true : occ;
				ul  = hit ? cell : ul;
				done = li >= ln;
			}
			else
			{
				if (kNest)
					sample_outcome();
				if (kCounts)
					++R.n_vol;
				occ        = ab > 0u;        // frag:276
				bool ended = false;
				if (occ)
				{
					if (SKIP != VKV_SKIP_NONE)
						ul = cell;
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
					if (ERT)
					{        // frag:293-299, as selects (the block form costs a save / restore of EXEC around two moves)
						ended = R.a > 0.99f;
						R.a   = ended ? 1.0f : R.a;
					}
				}
				else if (kCounts)
					++R.n_empty;
				if (!ended)
				{
					li     = i + (idx_t) 1;
					li_min = li;
				}
				done = ended || li >= ln;
			}
			if (__builtin_amdgcn_readfirstlane(++iter) == 48u)
				__builtin_amdgcn_s_setprio(3);
			continue;
		}
		// ---- the frag's state update (frag:224-310) as one block of selects ---------------------------------------------
		const bool smp    = !probe;
		const bool p_skip = probe && dist > 0u;        // frag:236-247
		const bool p_occ  = probe && dist == 0u;       // frag:248-262
		const bool occ_s  = ab > 0u;                   // frag:276
		const bool hit    = smp && occ_s;
		if (kCounts)
		{
			R.n_dist += probe ? 1u : 0u;
			R.n_vol += smp ? 1u : 0u;
			R.n_empty += (smp && !occ_s) ? 1u : 0u;
		}
		const float om = 1.0f - R.a;        // frag:287
		const float na = __builtin_fmaf(om, a, R.a);
		if (SEP)
			grey = hit ? __builtin_fmaf(om, c, grey) : grey;        // r = g = b: one channel is tracked
		else
		{
			const float r_ = L.g.unorm[texel & 255u] * a, g_ = L.g.unorm[(texel >> 8) & 255u] * a, b_ = L.g.unorm[(texel >> 16) & 255u] * a;
			const float nr = __builtin_fmaf(om, r_, R.r), ng = __builtin_fmaf(om, g_, R.g), nb = __builtin_fmaf(om, b_, R.b);
			R.r = hit ? nr : R.r, R.g = hit ? ng : R.g, R.b = hit ? nb : R.b;
		}
		const bool ended = ERT && hit && na > 0.99f;        // frag:293-299
		R.a              = hit ? (ended ? 1.0f : na) : R.a;
		lfirst           = (hit && a > 0.0f) ? i : lfirst;
		occ              = smp ? occ_s : (p_occ || occ);
		if (SKIP != VKV_SKIP_NONE)
			ul = (hit || p_occ) ? cell : ul;
		const int jb = max((int) i - A.back, (int) li_min);
		const int ni = p_skip ? (int) i + (int) skip : (p_occ ? jb : (int) i + 1);
		li_min       = smp ? (idx_t) ((int) i + 1) : li_min;
		li           = (idx_t) ni;
		done         = ended || ni >= R.n_steps;
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
	if (SEP)
		R.r = grey, R.g = grey, R.b = grey;
}

// the body of one workgroup: 16x16 pixels of the frame described by A; `bid` is the workgroup's id inside that frame's grid
template <int SKIP, bool ERT, int GRAD, bool PACKED, uint32_t LF, int WPB = 4>
__device__ __forceinline__ void lab_lean_block(const RayMarchArgs &A, uint32_t bid, RmLds &L)
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
	const uint32_t k = A.tile_order ? A.tile_order[rank] : rank;
	if (k >= A.tile_count)
		return;        // never with a well-formed order; keeps a damaged one (a target shared by two streams without an event) from becoming a wild address
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t       px, py, o;
	const uint32_t rb = (part * WPB + wave) * 64u + lane;
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
		if ((LF & kLabLut) != 0 && PACKED && GRAD != 2)
		{        // before the barrier of stage_tables_er
			if ((LF & kLabFull) != 0 && tf_is_separable(A))
				stage_full_lut(A, L);
			else
				stage_addr_lut(A);
		}
		const bool sep = stage_tables_er(A, L);
		if (marched)
		{
			if (sep)
				lab_lean_march<SKIP, ERT, GRAD, PACKED, true, LF>(A, R, L, iter, stamp);
			else
				lab_lean_march<SKIP, ERT, GRAD, PACKED, false, LF>(A, R, L, iter, stamp);
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
		if ((LF & kLabStamp) != 0 && iter == it && lane == (uint32_t) __builtin_ctzll(__ballot(iter == it)))
		{        // the stamps of the lane whose ray lived through every iteration of the wave (the others stop counting when their ray ends)
			unsigned long long *rec = A.trace + ((size_t) blockIdx.x * WPB + wave) * kTraceWords;
			rec[4] = ((unsigned long long) stamp.cnt[0] << 32) | stamp.sum[0], rec[5] = ((unsigned long long) stamp.cnt[1] << 32) | stamp.sum[1];
			rec[8] = ((unsigned long long) stamp.cnt[2] << 32) | stamp.sum[2];
		}
		else if ((LF & kLabStamp) == 0 && lane == (uint32_t) __builtin_ctzll(__ballot(1)))
		{
			unsigned long long *rec = A.trace + ((size_t) blockIdx.x * WPB + wave) * kTraceWords;
			rec[4] = 0, rec[5] = 0, rec[8] = 0;
		}
		if (lane == (uint32_t) __builtin_ctzll(__ballot(1)))
		{
			unsigned long long *rec = A.trace + ((size_t) blockIdx.x * WPB + wave) * kTraceWords;        // per launch (a batch: all its frames)
			rec[0] = t_start, rec[1] = t_end, rec[2] = it, rec[3] = ((unsigned long long) __builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | (k * A.blocks_per_tile + sb);
			rec[6] = c_start, rec[7] = __builtin_amdgcn_s_memtime(), rec[9] = 0;
		}
	}
}