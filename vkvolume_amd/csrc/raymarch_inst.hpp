// raymarch_inst.hpp — the kernel instantiations of the ray-march integrator and the launchers that choose among them.
// Included by raymarch_s<skip>e<ert>.hip, one translation unit per (skipping type, early ray termination) pair, each of which
// instantiates vkv::RayMarchLaunchers<SKIP, ERT> explicitly (eight files: the kernels of one pair compile in ~20 s, all pairs side by
// side), and by raymarch.hip, which only sees the declarations (extern template) and dispatches on the two run-time options.
#pragma once

#include "raymarch_core.hpp"

namespace vkv
{

enum Scheduler
{
	kSchedLean       = 0,        // k_raymarch_lean: one lane per ray, predicated loop body (default)
	kSchedPersistent = 1         // k_raymarch_persistent: resident waves, ballot + mbcnt lane re-fill (bit-identical, slower)
};

// The instantiations of the lean kernel a launch chooses from (raymarch_core.hpp explains the flags):
constexpr uint32_t kLfPlain  = 0;                                // footprint address worked out in registers: any volume, any map
constexpr uint32_t kLfLut    = kLeanLut;                         // two-level address tables in LDS (volumes up to ~2500 voxels per axis)
#ifdef VKV_LEAN_NO_ASYNC        // A/B builds only
constexpr uint32_t kLfFull   = kLeanLut | kLeanFull | kLeanSafe;
#else
constexpr uint32_t kLfFull   = kLeanLut | kLeanFull | kLeanSafe | kLeanAsync; // + one entry per voxel index with the separable transfer function, clamp-free loop, hand-set load waits
#endif
constexpr uint32_t kLfFullNc = kLfFull | kLeanNoCounts;          // the same without the per-pixel counters (what a renderer launches)
constexpr uint32_t kLfLutNc  = kLfLut | kLeanNoCounts;

struct LeanChoice
{
	int    kind;        // 0 plain, 1 two-level tables, 2 full tables
	size_t lds;         // dynamic LDS bytes (lean_lds_bytes: the kernels' whole LDS layout lives in the dynamic segment)
};

LeanChoice choose_lean(const RayMarchArgs &a, const VkvTuning &T);        // raymarch.hip

// A launch may run the loop without the three per-pixel counters when nothing reads them: no counter buffer and not the frag's
// sample-count test output (frag:324-334 turns n_vol + n_dist into the colour - Test::NumTextureSamples is independent of ERT and of
// the skipping type in the reference's GUI, src/volume_render.cpp:539)
inline bool wants_counts(const RayMarchArgs &a) { return a.out_counts != nullptr || a.test == VKV_TEST_NUM_TEXTURE_SAMPLES; }

template <int SKIP, bool ERT>
struct RayMarchLaunchers
{
	// one frame (vkv_render); grad: 0 unused, 1 precomputed map, 2 on the fly
	static int single(vkv_ctx *ctx, int sched, const VkvTuning &T, int grad, RayMarchArgs &a, hipStream_t s);
	// n frames interleaved in one grid (vkv_render_batch); c = {0, 0}: the frames disagree about the tables
	static void batch(int grad, const RayMarchArgs *d_frames, uint32_t n, uint32_t grid, uint32_t gpf, LeanChoice c, bool no_counts, hipStream_t s);
	// the same with resident workgroups whose waves pull their units; returns the grid size
	static uint32_t pull(vkv_ctx *ctx, int grad, const RayMarchArgs *d_frames, uint32_t n, uint32_t *d_heads, LeanChoice c, uint64_t units, hipStream_t s);
	// set-up: the runtime loads a translation unit's code object (device memory, milliseconds) at the first use of one of its kernels - make
	// that now, on the current device, instead of inside the first launch (vkv_prepare_render)
	static void load();
};

#ifdef VKV_RAYMARCH_INSTANTIATE

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
				no_counts = c.kind != 0 && !wants_counts(a);        // the common configuration only: every further instantiation costs build time
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

template <int SKIP, bool ERT>
int RayMarchLaunchers<SKIP, ERT>::single(vkv_ctx *ctx, int sched, const VkvTuning &T, int grad, RayMarchArgs &a, hipStream_t s)
{
	if (a.packed)
		return grad == 0 ? launch_one<SKIP, ERT, 0, true>(ctx, sched, T, a, s) : (grad == 1 ? launch_one<SKIP, ERT, 1, true>(ctx, sched, T, a, s) : launch_one<SKIP, ERT, 2, true>(ctx, sched, T, a, s));
	return grad == 0 ? launch_one<SKIP, ERT, 0, false>(ctx, sched, T, a, s) : (grad == 1 ? launch_one<SKIP, ERT, 1, false>(ctx, sched, T, a, s) : launch_one<SKIP, ERT, 2, false>(ctx, sched, T, a, s));
}

template <int SKIP, bool ERT>
void RayMarchLaunchers<SKIP, ERT>::load()
{
	hipFuncAttributes at;        // any kernel of this translation unit: the code object is loaded as a whole
	(void) hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_raymarch_lean<SKIP, ERT, 0, true, kLfPlain>));
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
void RayMarchLaunchers<SKIP, ERT>::batch(int grad, const RayMarchArgs *d_frames, uint32_t n, uint32_t grid, uint32_t gpf, LeanChoice c, bool no_counts, hipStream_t s)
{
	// c.kind > 0: every frame of the batch has address tables of that size (same volume extent) and they fit the LDS budget
	if (grad == 0)
		launch_batch_kind<SKIP, ERT, 0>(c, no_counts, d_frames, n, grid, gpf, s);
	else if (grad == 1)
		launch_batch_kind<SKIP, ERT, 1>(c, no_counts, d_frames, n, grid, gpf, s);
	else
		launch_batch_kind<SKIP, ERT, 2>(c, no_counts, d_frames, n, grid, gpf, s);
}

// ---- the same, with resident workgroups whose waves pull their units ------------------------------------------------
// grid = the workgroups the device holds at once (occupancy of this instantiation x CUs), never more than there are units / 4
template <int SKIP, bool ERT, int GRAD, uint32_t LF>
static uint32_t launch_pull_one(vkv_ctx *ctx, const RayMarchArgs *d_frames, uint32_t n, uint32_t *d_heads, size_t lds, uint64_t units, hipStream_t s)
{
	// eight workgroups of four waves fill a CU's 32 wave slots (this instantiation is held to 64 VGPRs); should fewer fit, the surplus
	// workgroups start when others have finished and take what tickets are left
	const uint64_t resident = (uint64_t) 8 * (uint64_t) (ctx->cu_count > 1 ? ctx->cu_count : 1);
	const uint64_t want     = resident < (units + 3) / 4 ? resident : (units + 3) / 4;
	const uint32_t grid     = (uint32_t) (want < 8 ? 8 : want);
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

template <int SKIP, bool ERT>
uint32_t RayMarchLaunchers<SKIP, ERT>::pull(vkv_ctx *ctx, int grad, const RayMarchArgs *d_frames, uint32_t n, uint32_t *d_heads, LeanChoice c, uint64_t units, hipStream_t s)
{
	return grad == 0 ? launch_pull_kind<SKIP, ERT, 0>(ctx, c, d_frames, n, d_heads, units, s)
	                 : (grad == 1 ? launch_pull_kind<SKIP, ERT, 1>(ctx, c, d_frames, n, d_heads, units, s) : launch_pull_kind<SKIP, ERT, 2>(ctx, c, d_frames, n, d_heads, units, s));
}

#else        // declarations only (raymarch.hip): the definitions are instantiated by the eight raymarch_s*e*.hip files

extern template struct RayMarchLaunchers<VKV_SKIP_NONE, false>;
extern template struct RayMarchLaunchers<VKV_SKIP_NONE, true>;
extern template struct RayMarchLaunchers<VKV_SKIP_BLOCK, false>;
extern template struct RayMarchLaunchers<VKV_SKIP_BLOCK, true>;
extern template struct RayMarchLaunchers<VKV_SKIP_DISTANCE, false>;
extern template struct RayMarchLaunchers<VKV_SKIP_DISTANCE, true>;
extern template struct RayMarchLaunchers<VKV_SKIP_ANISOTROPIC_DISTANCE, false>;
extern template struct RayMarchLaunchers<VKV_SKIP_ANISOTROPIC_DISTANCE, true>;

#endif

}        // namespace vkv
