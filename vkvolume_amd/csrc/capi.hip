// capi.hip — the extern "C" boundary of include/vkvolume_amd.h: context, argument checking, host-side
// uniform / transfer-function helpers, and dispatch to the kernel launchers.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <new>

#include <algorithm>
#include <exception>
#include <string>
#include <vector>

#include "../host/load_volume.h"
#include "../host/vkv_math.hpp"
#include "vkv_device.hpp"
#include "../../include/vkvolume_amd_debug.h"

namespace vkv
{
int launch_gradient_map(vkv_ctx *, const uint8_t *, uint8_t *, VkvExtent3D, const VkvTransferFunctionUniform *, hipStream_t);
int launch_occupancy_map(vkv_ctx *, const uint8_t *, const uint8_t *, const uint8_t *, const VkvTransferFunctionUniform *, VkvExtent3D, uint8_t *,
                         VkvExtent3D, hipStream_t);
int launch_distance_map(vkv_ctx *, uint8_t *, uint8_t *, VkvExtent3D, hipStream_t);
int launch_distance_map_anisotropic(vkv_ctx *, uint8_t *const[8], uint8_t *, VkvExtent3D, hipStream_t);
int launch_synth_volume(vkv_ctx *, uint8_t *, VkvExtent3D, uint32_t, uint32_t, hipStream_t);
int launch_scatter_tiles_frames(vkv_ctx *, uint32_t, void *const *, const void *const *, const VkvTileRect *, const uint32_t *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                                uint32_t, hipStream_t);
void screen_tile_rect(const VkvRayCastUniform *, const VkvRayGen *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, VkvTileRect *);
int prepare_render(vkv_ctx *, const VkvRenderParams *, uint32_t, hipStream_t);
void load_feedback_code();
int launch_render(vkv_ctx *, const VkvRenderParams *, const float *, hipStream_t);
int launch_render_batch(vkv_ctx *, const VkvRenderParams *, uint32_t, const float *, hipStream_t);
int launch_pack_volume(vkv_ctx *, const uint8_t *, const uint8_t *, VkvExtent3D, void *, hipStream_t);
int launch_check_numerics(vkv_ctx *, int, uint32_t, uint64_t, unsigned long long *, hipStream_t);
int launch_tf_tables(vkv_ctx *, const uint8_t *, const VkvTransferFunctionUniform *, uint32_t *, hipStream_t);
int launch_convert_volume(vkv_ctx *, const void *, int, bool, float, float, uint64_t, uint8_t *, hipStream_t);
int launch_occupied_voxel_count(vkv_ctx *, const uint8_t *, const uint8_t *, const VkvTransferFunctionUniform *, VkvExtent3D, uint64_t *, hipStream_t);

int set_error(vkv_ctx *ctx, int code, const char *fmt, ...)
{
	if (ctx)
	{
		va_list ap;
		va_start(ap, fmt);
		vsnprintf(ctx->error, sizeof(ctx->error), fmt, ap);
		va_end(ap);
	}
	return code;
}

VkvTuning tuning_of(vkv_ctx *ctx)
{
	std::lock_guard<std::mutex> lock(ctx->mutex);
	return ctx->tuning;
}

// `bytes` of device memory that stay valid until vkv_destroy (caller holds ctx->mutex).  Launch paths only take from the arena; set-up
// calls may fall back to hipMalloc.
// The arena has two regions: the first kScratchReserve scratch blocks (stream_scratch) and, behind them, the tables.  A renderer that
// keeps meeting new window sizes fills the TABLE region (a 1920x1080 schedule of 16x16 tiles costs 32 KiB per distinct size) - it can
// never take the room a new stream's scratch block needs, and vkv_trim gives the table region back.
static void *arena_take_table(vkv_ctx *ctx, size_t bytes, bool setup, bool *from_arena)
{
	const size_t need = (bytes + 255u) & ~(size_t) 255u;
	*from_arena       = false;
	if (ctx->arena && ctx->table_used + need <= ctx->arena_bytes - ctx->table_base)
	{
		void *p = ctx->arena + ctx->table_base + ctx->table_used;
		ctx->table_used += need;
		*from_arena = true;
		return p;
	}
	if (!setup)
		return nullptr;
	void *p = nullptr;
	if (hipMalloc(&p, need) != hipSuccess)
		return nullptr;
	ctx->overflow.push_back(p);
	return p;
}

uint8_t *stream_scratch(vkv_ctx *ctx, hipStream_t stream, bool setup)
{
	std::lock_guard<std::mutex> lock(ctx->mutex);
	auto                        it = ctx->scratch.find(stream);
	if (it != ctx->scratch.end())
		return it->second;
	uint8_t *p = nullptr;
	if (!ctx->free_scratch.empty())
	{
		p = ctx->free_scratch.back();
		ctx->free_scratch.pop_back();
	}
	else if (ctx->arena && ctx->scratch_used + kScratchBytes <= ctx->table_base)
	{
		p = ctx->arena + ctx->scratch_used;
		ctx->scratch_used += kScratchBytes;
	}
	else if (setup)
	{        // more streams than the arena reserves blocks for: a set-up call may allocate (freed by vkv_destroy)
		void *q = nullptr;
		if (hipMalloc(&q, kScratchBytes) == hipSuccess)
		{
			ctx->overflow_scratch.push_back(q);
			p = static_cast<uint8_t *>(q);
		}
	}
	if (!p)
	{
		set_error(ctx, VKV_E_UNSUPPORTED, "no scratch block left for a new stream: the arena reserves %zu (call vkv_prepare_render for the stream at set-up "
		                                  "time, give finished streams back with vkv_release_stream, or raise VKV_ARENA_BYTES)", ctx->table_base / kScratchBytes);
		return nullptr;
	}
	ctx->scratch.emplace(stream, p);
	return p;
}

// Device copy of a new table: memory out of the arena, asynchronous upload from the entry's own host copy on the launch's stream, an
// event behind it for launches on other streams.  Caller holds ctx->mutex.  False (and nothing allocated that matters) when there is no room.
static bool table_upload(vkv_ctx *ctx, vkv_ctx::Table &t, hipStream_t s, bool setup)
{
	const size_t bytes = t.host.size() * sizeof(uint32_t);
	bool         from_arena = false;
	t.d                = static_cast<uint32_t *>(arena_take_table(ctx, bytes, setup, &from_arena));
	if (!t.d)
		return false;
	if (hipEventCreateWithFlags(&t.uploaded, hipEventDisableTiming) != hipSuccess)
		return false;        // (the arena bytes stay taken until the next vkv_trim: harmless)
	// The source of the asynchronous copy is the table's twin in the PINNED mirror of the table region (same offset): a copy from pageable
	// memory may block the enqueueing thread behind earlier work of the stream, which a launch must not do; the mirror lives as long as
	// the arena, so the source outlives the copy whatever happens to the entry.  A table a set-up call put outside the arena (hipMalloc
	// fallback) is copied from the entry's own vector and waited for right here.
	const void *src = t.host.data();
	if (from_arena && ctx->table_mirror)
	{
		uint8_t *m = ctx->table_mirror + (reinterpret_cast<uint8_t *>(t.d) - (ctx->arena + ctx->table_base));
		std::memcpy(m, t.host.data(), bytes);
		src = m;
	}
	const bool queued = hipMemcpyAsync(t.d, src, bytes, hipMemcpyHostToDevice, s) == hipSuccess;
	if (!queued || hipEventRecord(t.uploaded, s) != hipSuccess)
	{
		if (queued)
			(void) hipStreamSynchronize(s);        // the copy may still be reading its source: not while the caller deletes the entry
		(void) hipEventDestroy(t.uploaded);
		t.uploaded = nullptr;
		return false;
	}
	t.upload_stream = s;
	if (setup || src == t.host.data())
	{        // a set-up call hands out finished tables (and a pageable source must not be left to an asynchronous copy)
		(void) hipEventSynchronize(t.uploaded);
		t.ready = true;
	}
	return true;
}

// the table for a launch on stream s: behind its upload (caller holds ctx->mutex)
static const uint32_t *table_on_stream(vkv_ctx::Table &t, hipStream_t s, bool setup)
{
	if (t.ready)
		return t.d;
	if (setup ? hipEventSynchronize(t.uploaded) == hipSuccess : hipEventQuery(t.uploaded) == hipSuccess)
	{
		t.ready = true;
		return t.d;
	}
	if (s != t.upload_stream && hipStreamWaitEvent(s, t.uploaded, 0) != hipSuccess)
		return nullptr;
	return t.d;
}

constexpr size_t kMaxCachedTables = 1024;        // per kind; beyond that a launch runs without (never evicts: a launch may still read any of them)

const uint32_t *packed_addr_lut(vkv_ctx *ctx, int W, int H, int D, uint32_t *lut_y, uint32_t *lut_z, uint32_t *words, hipStream_t stream, bool setup)
{
	std::lock_guard<std::mutex> lock(ctx->mutex);
	for (auto *t : ctx->addr_luts)
		if (t->W == W && t->H == H && t->D == D)
		{
			*lut_y = t->lut_y, *lut_z = t->lut_z, *words = t->words;
			return table_on_stream(t->table, stream, setup);
		}
	if (ctx->addr_luts.size() >= kMaxCachedTables)
		return nullptr;
	// two levels per axis: position inside a macro-brick (32 entries: padded index b & 31) and the macro-brick term (b >> 5)
	const PackedDims pd  = packed_dims(W, H, D);
	const uint32_t   nmx = (uint32_t) (W + 1) / 32 + 1, nmy = (uint32_t) (H + 1) / 32 + 1, nmz = (uint32_t) (D + 1) / 32 + 1;
	const uint32_t   ny = 96 + nmx, nz = (ny + nmy + 1) & ~1u, total = nz + 2 * nmz;
	auto *           e  = new (std::nothrow) vkv_ctx::AddrLut{W, H, D, ny, nz, total, {}};
	if (!e)
		return nullptr;
	std::vector<uint32_t> &h = e->table.host;
	h.assign(total, 0u);
	for (uint32_t j = 0; j < 32; ++j)
	{
		h[j]      = (((j >> 2) & 7u) << 8) + (j & 3u) * 2u;
		h[32 + j] = (((j >> 2) & 7u) << 11) + (j & 3u) * 10u;
		h[64 + j] = (((j >> 2) & 7u) << 14) + (j & 3u) * 50u;
	}
	for (uint32_t m = 0; m < nmx; ++m)
		h[96 + m] = m << 17;
	for (uint32_t m = 0; m < nmy; ++m)
		h[ny + m] = (m * (uint32_t) pd.mx) << 17;
	for (uint32_t m = 0; m < nmz; ++m)
	{
		const uint64_t z = ((uint64_t) m * (uint64_t) pd.my * (uint64_t) pd.mx) << 17;
		h[nz + 2 * m] = (uint32_t) z, h[nz + 2 * m + 1] = (uint32_t) (z >> 32);
	}
	if (!table_upload(ctx, e->table, stream, setup))
	{
		delete e;
		return nullptr;
	}
	ctx->addr_luts.push_back(e);
	*lut_y = ny, *lut_z = nz, *words = total;
	return e->table.d;
}

const uint32_t *tile_start_order(vkv_ctx *ctx, uint32_t img_w, uint32_t img_h, uint32_t tile_w, uint32_t tile_h, uint32_t first, uint32_t stride, uint32_t count,
                                 hipStream_t stream, bool setup)
{
	if (count < 2)
		return nullptr;
	const uint32_t tiles_x = (img_w + tile_w - 1) / tile_w, tiles_y = (img_h + tile_h - 1) / tile_h;
	std::lock_guard<std::mutex> lock(ctx->mutex);
	const float mix_heavy_f = ctx->tuning.tile_mix_heavy, mix_spread_f = ctx->tuning.tile_mix_spread;
	for (auto *t : ctx->tile_orders)
		if (t->tiles_x == tiles_x && t->tiles_y == tiles_y && t->tile_w == tile_w && t->tile_h == tile_h && t->img_w == img_w && t->img_h == img_h && t->first == first &&
		    t->stride == stride && t->count == count && t->mix_heavy == mix_heavy_f && t->mix_spread == mix_spread_f)
			return table_on_stream(t->table, stream, setup);
	if (ctx->tile_orders.size() >= kMaxCachedTables)
		return nullptr;
	std::vector<std::pair<double, uint32_t>> key(count);
	for (uint32_t k = 0; k < count; ++k)
	{
		const uint64_t t  = (uint64_t) first + (uint64_t) k * stride;
		const double   cx = ((double) (t % tiles_x) + 0.5) * tile_w - 0.5 * img_w, cy = ((double) (t / tiles_x) + 0.5) * tile_h - 0.5 * img_h;
		key[k]            = {cx * cx + cy * cy, k};
	}
	std::stable_sort(key.begin(), key.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
	std::vector<uint32_t> order(count);
	for (uint32_t r = 0; r < count; ++r)
		order[r] = key[r].second;
	// experiment (VkvTuning.tile_mix_heavy / tile_mix_spread): the central <heavy share> of the tiles is spread evenly over the first
	// <spread> of the start order, the remaining (border) tiles fill the gaps and the end
	const double mix_heavy = mix_heavy_f, mix_spread = mix_spread_f;
	if (mix_heavy > 0.0 && mix_heavy < 1.0 && mix_spread >= mix_heavy && mix_spread <= 1.0)
	{
		// in groups of eight ranks: rank r runs on XCD r & 7, so a group gives every XCD one tile of the same kind
		const uint32_t groups = count / 8, nh = (uint32_t) (mix_heavy * groups), span = (uint32_t) (mix_spread * groups);
		std::vector<uint32_t> mixed;
		mixed.reserve(count);
		uint32_t h = 0, l = nh;
		for (uint32_t r = 0; r < groups; ++r)
		{
			// heavy group number h is due at position h * span / nh
			const bool     take_heavy = h < nh && ((uint64_t) h * span <= (uint64_t) r * nh || l >= groups);
			const uint32_t g          = take_heavy ? h++ : l++;
			for (uint32_t j = 0; j < 8; ++j)
				mixed.push_back(order[g * 8 + j]);
		}
		for (uint32_t r = groups * 8; r < count; ++r)
			mixed.push_back(order[r]);
		order.swap(mixed);
	}
	auto *e = new (std::nothrow) vkv_ctx::TileOrder{tiles_x, tiles_y, tile_w, tile_h, img_w, img_h, first, stride, count, mix_heavy_f, mix_spread_f, {}};
	if (!e)
		return nullptr;
	e->table.host.swap(order);
	if (!table_upload(ctx, e->table, stream, setup))
	{
		delete e;
		return nullptr;
	}
	ctx->tile_orders.push_back(e);
	return e->table.d;
}

int check_launch(vkv_ctx *ctx, const char *what)
{
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess)
		return set_error(ctx, (int) e, "%s: %s", what, hipGetErrorString(e));
	return VKV_OK;
}

static bool extent_ok(VkvExtent3D e) { return e.width > 0 && e.height > 0 && e.depth > 0; }

// ceil(volume / map) must reproduce a valid block size (src/compute_distance_map.cpp:110-113)
static bool map_extent_ok(VkvExtent3D e, VkvExtent3D me) { return extent_ok(me) && me.width <= e.width && me.height <= e.height && me.depth <= e.depth; }

}        // namespace vkv

using namespace vkv;

extern "C" {

const char *vkv_version(void) { return "vkvolume_amd 0.1.0 (gfx950)"; }

// Range checks of a tuning block, shared by vkv_set_tuning (which rejects a bad block) and default_tuning (which falls back to the
// built-in value of a field the environment set out of range).  Returns null when the block is fine, else what is wrong with it.
static const char *tuning_problem(const VkvTuning &t)
{
	if (t.scheduler < 0 || t.scheduler > 1 || t.batch_mode < 0 || t.batch_mode > 1 || t.address_tables < 0 || t.address_tables > 2 || t.feedback_period == 0 ||
	    t.gradient_segment > 255u || (t.pack_tile != 0 && t.pack_tile != 2 && t.pack_tile != 4) || t.clamp_always < 0 || t.clamp_always > 1 ||
	    t.occupancy_kernel < 0 || t.occupancy_kernel > 1 || (t.wave_shape != 0 && t.wave_shape != 4 && t.wave_shape != 8 && t.wave_shape != 16))
		return "field out of range";
	// a tile mix that is not a number never compares equal to a cached schedule's: every launch would build a new table
	if (!std::isfinite(t.tile_mix_heavy) || !std::isfinite(t.tile_mix_spread) || t.tile_mix_heavy < 0.0f || t.tile_mix_heavy > 1.0f || t.tile_mix_spread < 0.0f ||
	    t.tile_mix_spread > 1.0f)
		return "tile_mix_heavy / tile_mix_spread must be numbers in [0, 1]";
	return nullptr;
}

// the kernels request their whole LDS layout as dynamic LDS without raising the 64 KiB default limit: a larger figure would make every
// launch fail instead of choosing the smaller tables
static void clamp_tuning(VkvTuning &t) { t.full_table_lds_limit = std::min<uint32_t>(t.full_table_lds_limit, kMaxDynamicLds); }

// defaults of the tuning block, then the environment (read HERE, once per context, and nowhere else)
static void default_tuning(VkvTuning &t)
{
	std::memset(&t, 0, sizeof(t));
	t.struct_size          = (uint32_t) sizeof(VkvTuning);
	t.address_tables       = 2;
	t.full_table_lds_limit = 17920;        // = kFullLdsLimit (raymarch_core.hpp): 9 workgroups per CU
	t.screen_cull          = 1;
	t.feedback             = 1;
	t.feedback_period      = 8;
	t.arena_bytes          = 8u << 20;
	auto env = [](const char *name) -> const char * { const char *e = std::getenv(name); return (e && e[0]) ? e : nullptr; };
	if (const char *e = env("VKV_RAYMARCH_SCHEDULER"))
		t.scheduler = e[0] == 'p';
	if (const char *e = env("VKV_RAYMARCH_BATCH"))
		t.batch_mode = e[0] == 'p';
	if (const char *e = env("VKV_RAYMARCH_BATCH_ORDER"))
		t.batch_sequential = e[0] == 's';
	if (const char *e = env("VKV_RAYMARCH_TILE_ORDER"))
		t.tile_order_linear = e[0] == 'l';
	if (const char *e = env("VKV_RAYMARCH_LUT"))
		t.address_tables = e[0] == '0' ? 0 : (e[0] == '2' ? 1 : 2);
	if (const char *e = env("VKV_RAYMARCH_FULL_LIMIT"))
		t.full_table_lds_limit = (uint32_t) std::max(0l, std::atol(e));
	if (const char *e = env("VKV_RAYMARCH_CULL"))
		t.screen_cull = e[0] != '0';
	if (const char *e = env("VKV_RAYMARCH_FEEDBACK"))
		t.feedback = e[0] != '0';
	if (const char *e = env("VKV_RAYMARCH_FEEDBACK_PERIOD"))
		t.feedback_period = (uint32_t) std::max(1l, std::atol(e));
	if (const char *e = env("VKV_RAYMARCH_TILE_MIX"))
	{
		double h = 0.0, sp = 0.0;
		if (std::sscanf(e, "%lf,%lf", &h, &sp) == 2)
			t.tile_mix_heavy = (float) h, t.tile_mix_spread = (float) sp;
	}
	if (const char *e = env("VKV_GRADIENT_SEGMENT"))
		t.gradient_segment = (uint32_t) std::min(std::max(std::atol(e), 0l), 255l);
	if (const char *e = env("VKV_PACK_TILE"))
		t.pack_tile = std::atoi(e);
	if (const char *e = env("VKV_ARENA_BYTES"))
		t.arena_bytes = (uint32_t) std::min(std::max(std::atol(e), 1l << 20), 1l << 30);
	if (const char *e = env("VKV_RAYMARCH_CLAMP"))
		t.clamp_always = e[0] == 'a';
	if (const char *e = env("VKV_RAYMARCH_WAVE_SHAPE"))
	{
		const int v = std::atoi(e);
		t.wave_shape = (v == 4 || v == 8 || v == 16) ? v : 0;
	}
	if (const char *e = env("VKV_OCCUPANCY_KERNEL"))
		t.occupancy_kernel = e[0] == 'r';
	// the environment gets the checks vkv_set_tuning applies: an out-of-range value falls back to the built-in default of its group
	clamp_tuning(t);
	if (tuning_problem(t))
	{
		if (!std::isfinite(t.tile_mix_heavy) || !std::isfinite(t.tile_mix_spread) || t.tile_mix_heavy < 0.0f || t.tile_mix_heavy > 1.0f || t.tile_mix_spread < 0.0f ||
		    t.tile_mix_spread > 1.0f)
			t.tile_mix_heavy = t.tile_mix_spread = 0.0f;
		if (t.pack_tile != 0 && t.pack_tile != 2 && t.pack_tile != 4)
			t.pack_tile = 0;
	}
}

// every cached table gone, the table region of the arena empty again (caller: the device is idle, ctx->mutex held or nobody else around)
static void drop_tables(vkv_ctx *ctx)
{
	for (auto *t : ctx->tile_orders)
	{
		if (t->table.uploaded)
			(void) hipEventDestroy(t->table.uploaded);
		delete t;
	}
	for (auto *t : ctx->addr_luts)
	{
		if (t->table.uploaded)
			(void) hipEventDestroy(t->table.uploaded);
		delete t;
	}
	ctx->tile_orders.clear();
	ctx->addr_luts.clear();
	for (void *p : ctx->overflow)
		(void) hipFree(p);
	ctx->overflow.clear();
	ctx->table_used = 0;
}

// the argument blocks of captured vkv_render_batch launches: the graphs that read them are the caller's, who promised not to replay them
static void drop_capture_blocks(vkv_ctx *ctx)
{
	std::vector<vkv_ctx::CaptureSlot> kept;
	for (auto &c : ctx->capture_slots)
	{
		if (c.pooled)
		{
			c.in_use = false, c.owner = nullptr;
			kept.push_back(c);
		}
		else
		{
			(void) hipHostFree(c.pinned);
			(void) hipFree(c.device);
		}
	}
	ctx->capture_slots.swap(kept);
}

int vkv_release_captured(vkv_ctx *ctx, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	std::lock_guard<std::mutex> lock(ctx->mutex);
	for (auto &c : ctx->capture_slots)
		if (c.in_use && c.owner == (hipStream_t) stream)
			c.in_use = false, c.owner = nullptr;
	return VKV_OK;
}

int vkv_trim(vkv_ctx *ctx)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard      guard(ctx->device);
	// the lock first: the launch paths take it to look their tables up, so no launch can slip in between the wait and the drop
	std::lock_guard<std::mutex> lock(ctx->mutex);
	const hipError_t e = hipDeviceSynchronize();        // launches that still read a table
	if (e != hipSuccess)
		return set_error(ctx, (int) e, "trim: %s", hipGetErrorString(e));
	drop_tables(ctx);
	drop_capture_blocks(ctx);
	return VKV_OK;
}

int vkv_create(int device_ordinal, vkv_ctx **out_ctx)
{
	if (!out_ctx)
		return VKV_E_INVALID_ARGUMENT;
	*out_ctx  = nullptr;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_ordinal < 0 || device_ordinal >= count)
		return VKV_E_NO_DEVICE;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess)
		return VKV_E_NO_DEVICE;
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !std::getenv("VKV_ALLOW_ANY_ARCH"))
		return VKV_E_NO_DEVICE;        // the code object is built for gfx950 only
	vkv_ctx *ctx = new (std::nothrow) vkv_ctx();
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	ctx->device   = device_ordinal;
	ctx->cu_count = prop.multiProcessorCount;
	ctx->error[0] = 0;
	ctx->d_trace  = nullptr;
	ctx->d_debug_orders = nullptr, ctx->debug_order_frames = ctx->debug_order_count = 0;
	default_tuning(ctx->tuning);
	{        // the arena every later launch carves its tables and scratch blocks out of (the one allocation of the render path)
		DeviceGuard guard(device_ordinal);
		void *      p = nullptr;
		if (hipMalloc(&p, ctx->tuning.arena_bytes) != hipSuccess)
		{
			delete ctx;
			return VKV_E_NO_DEVICE;
		}
		ctx->arena = static_cast<uint8_t *>(p), ctx->arena_bytes = ctx->tuning.arena_bytes;
		// scratch region: kScratchReserve blocks, at most half of a small arena; the tables get the rest and a pinned host mirror of it
		const size_t blocks = std::min<size_t>(kScratchReserve, ctx->arena_bytes / 2 / kScratchBytes);
		ctx->table_base = blocks * kScratchBytes, ctx->scratch_used = 0, ctx->table_used = 0;
		void *m = nullptr;
		if (hipHostMalloc(&m, ctx->arena_bytes - ctx->table_base, hipHostMallocDefault) == hipSuccess)
			ctx->table_mirror = static_cast<uint8_t *>(m);        // (without it uploads fall back to the entry's own vector + a wait)
		void *cp = nullptr, *cd = nullptr;
		if (hipHostMalloc(&cp, (size_t) kCaptureSlots * kCaptureSlotBytes, hipHostMallocDefault) == hipSuccess &&
		    hipMalloc(&cd, (size_t) kCaptureSlots * kCaptureSlotBytes) == hipSuccess)
		{        // (without them a captured vkv_render_batch allocates its blocks during the capture)
			ctx->capture_pool = static_cast<uint8_t *>(cp), ctx->capture_pool_device = static_cast<uint8_t *>(cd);
			for (uint32_t i = 0; i < kCaptureSlots; ++i)
			{
				vkv_ctx::CaptureSlot c;
				c.pinned = ctx->capture_pool + (size_t) i * kCaptureSlotBytes, c.device = ctx->capture_pool_device + (size_t) i * kCaptureSlotBytes, c.pooled = true;
				ctx->capture_slots.push_back(c);
			}
		}
		else if (cp)
			(void) hipHostFree(cp);
	}
	*out_ctx      = ctx;        // the caller's current device is left as it is: every entry point switches to ctx->device itself
	return VKV_OK;
}

void vkv_destroy(vkv_ctx *ctx)
{
	if (!ctx)
		return;
	{
		DeviceGuard guard(ctx->device);
		(void) hipDeviceSynchronize();        // launches that still read the context's tables, scratch or feedback buffers
		drop_tables(ctx);
		for (auto *f : ctx->feedback)
		{
			(void) hipFree(f->d_cost);
			(void) hipFree(f->d_order);
					delete f;
		}
		for (void *p : ctx->overflow_scratch)
			(void) hipFree(p);
		drop_capture_blocks(ctx);
		if (ctx->capture_pool)
			(void) hipHostFree(ctx->capture_pool);
		if (ctx->capture_pool_device)
			(void) hipFree(ctx->capture_pool_device);
		if (ctx->table_mirror)
			(void) hipHostFree(ctx->table_mirror);
		(void) hipFree(ctx->arena);
	}
	delete ctx;
}

int vkv_get_tuning(const vkv_ctx *ctx, VkvTuning *out)
{
	if (!ctx || !out)
		return VKV_E_INVALID_ARGUMENT;
	std::lock_guard<std::mutex> lock(const_cast<vkv_ctx *>(ctx)->mutex);
	*out = ctx->tuning;
	return VKV_OK;
}

int vkv_set_tuning(vkv_ctx *ctx, const VkvTuning *tuning)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	if (!tuning || tuning->struct_size != sizeof(VkvTuning))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "set_tuning: struct_size must be sizeof(VkvTuning) = %zu (start from vkv_get_tuning)", sizeof(VkvTuning));
	if (const char *why = tuning_problem(*tuning))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "set_tuning: %s", why);
	std::lock_guard<std::mutex> lock(ctx->mutex);
	const uint32_t              arena = ctx->tuning.arena_bytes;
	ctx->tuning                       = *tuning;
	ctx->tuning.arena_bytes           = arena;        // read-only
	clamp_tuning(ctx->tuning);
	return VKV_OK;
}

int vkv_release_stream(vkv_ctx *ctx, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	std::lock_guard<std::mutex> lock(ctx->mutex);
	auto                        it = ctx->scratch.find((hipStream_t) stream);
	if (it != ctx->scratch.end())
	{
		ctx->free_scratch.push_back(it->second);
		ctx->scratch.erase(it);
	}
	return VKV_OK;
}

int vkv_register_target(vkv_ctx *ctx, const void *d_target, uint32_t image_width, uint32_t image_height, const VkvTileSchedule *tiles)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_target || !tiles || image_width == 0 || image_height == 0 || tiles->tile_width == 0 || tiles->tile_height == 0 || tiles->tile_stride == 0)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "register_target: null pointer or zero size");
	(void) vkv_forget_target(ctx, d_target);        // a target has one state: registering it again replaces it
	if (tiles->tile_count == 0)
		return VKV_OK;
	uint32_t *cost = nullptr, *order = nullptr;
	const size_t bytes = (size_t) tiles->tile_count * sizeof(uint32_t);
	hipError_t   e      = hipMalloc((void **) &cost, bytes);
	if (e == hipSuccess)
		e = hipMalloc((void **) &order, bytes);
	std::vector<uint32_t> identity(tiles->tile_count);
	for (uint32_t i = 0; i < tiles->tile_count; ++i)
		identity[i] = i;
	// the order starts out as a valid permutation, the costs as zero; both are complete when this call returns
	if (e == hipSuccess)
		e = hipMemset(cost, 0, bytes);
	if (e == hipSuccess)
		e = hipMemcpy(order, identity.data(), bytes, hipMemcpyHostToDevice);
	const bool     whole   = tiles->rect.w == 0 || tiles->rect.h == 0;
	const uint32_t org_x = whole ? 0u : tiles->rect.x0 * tiles->tile_width, org_y = whole ? 0u : tiles->rect.y0 * tiles->tile_height;
	const uint32_t tiles_x = whole ? (image_width + tiles->tile_width - 1) / tiles->tile_width : tiles->rect.w;
	auto *f = e == hipSuccess ? new (std::nothrow) vkv_ctx::TileFeedback{d_target, image_width, image_height, tiles->tile_width, tiles->tile_height, tiles->tile_first,
	                                                                      tiles->tile_stride, tiles->tile_count, org_x, org_y, tiles_x, cost, order, false, 0u, 0u, 8u, 0u}
	                          : nullptr;
	if (!f)
	{
		(void) hipFree(cost);
		(void) hipFree(order);
		return set_error(ctx, e != hipSuccess ? (int) e : VKV_E_UNSUPPORTED, "register_target: %s", e != hipSuccess ? hipGetErrorString(e) : "out of memory");
	}
	load_feedback_code();        // the sort kernels' code object on this device now, not inside the first launch into the target
	std::lock_guard<std::mutex> lock(ctx->mutex);
	f->period = ctx->tuning.feedback_period;
	ctx->feedback.push_back(f);
	return VKV_OK;
}

int vkv_forget_target(vkv_ctx *ctx, const void *d_target)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard            guard(ctx->device);
	vkv_ctx::TileFeedback *f = nullptr;
	{
		std::lock_guard<std::mutex> lock(ctx->mutex);
		for (size_t i = 0; i < ctx->feedback.size(); ++i)
			if (ctx->feedback[i]->target == d_target)
			{
				f = ctx->feedback[i];
				ctx->feedback.erase(ctx->feedback.begin() + (long) i);
				break;
			}
	}
	if (!f)
		return VKV_OK;
	(void) hipDeviceSynchronize();        // launches that still write costs or read the order (the entry is out of the list: no new ones)
	(void) hipFree(f->d_cost);
	(void) hipFree(f->d_order);
	delete f;
	return VKV_OK;
}

// ---- diagnostic entry points: include/vkvolume_amd_debug.h (not part of the drop-in boundary) ----
int vkv_debug_trace(vkv_ctx *ctx, void *d_buffer)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	ctx->d_trace = d_buffer;
	return VKV_OK;
}

int vkv_debug_tile_orders(vkv_ctx *ctx, const uint32_t *d_orders, uint32_t frames, uint32_t count)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	ctx->d_debug_orders = d_orders, ctx->debug_order_frames = d_orders ? frames : 0u, ctx->debug_order_count = count;
	return VKV_OK;
}

int vkv_debug_check(vkv_ctx *ctx, int32_t what, uint32_t first_bits, uint64_t count, uint64_t *d_mismatches, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_mismatches)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "check: null pointer");
	return launch_check_numerics(ctx, what, first_bits, count, reinterpret_cast<unsigned long long *>(d_mismatches), (hipStream_t) stream);
}

const char *vkv_last_error(const vkv_ctx *ctx) { return ctx ? ctx->error : "null context"; }

// ---- host helpers ------------------------------------------------------------------------------

// src/volume_component.cpp:226-240
int vkv_transfer_function_uniform(const VkvVolumeOptions *o, VkvTransferFunctionUniform *u)
{
	if (!o || !u)
		return VKV_E_INVALID_ARGUMENT;
	u->sampling_factor         = o->sampling_factor;
	u->voxel_alpha_factor      = o->voxel_alpha_factor;
	u->grad_magnitude_modifier = 1.0f;
	u->use_gradient            = o->gradient_max != o->gradient_min;
	u->intensity_min           = o->intensity_min;
	u->intensity_range_inv     = 1.0f / (o->intensity_max - o->intensity_min);
	u->gradient_min            = o->gradient_min;
	u->gradient_range_inv      = 1.0f / (o->gradient_max - o->gradient_min);
	return VKV_OK;
}

// src/volume_component.cpp:242-261
int vkv_transfer_function_texture(const VkvVolumeOptions *o, uint8_t *tex)
{
	if (!o || !tex)
		return VKV_E_INVALID_ARGUMENT;
	auto        clampf       = [](float x, float lo, float hi) { return std::min(std::max(x, lo), hi); };
	const float i_inv        = 1.0f / (o->intensity_max - o->intensity_min);
	const float g_inv        = 1.0f / (o->gradient_max - o->gradient_min);
	const bool  use_gradient = o->gradient_max != o->gradient_min;
	size_t      idx          = 0;
	for (int gi = 0; gi < 256; ++gi)
		for (int ii = 0; ii < 256; ++ii, ++idx)
		{
			const float   g = (float) gi, i = (float) ii;
			const float   alpha_i = clampf(((i / 255.0f) - o->intensity_min) * i_inv, 0.0f, 1.0f);
			const float   alpha_g = use_gradient ? clampf(((g / 255.0f) - o->gradient_min) * g_inv, 0.0f, 1.0f) : 1.0f;
			const uint8_t alpha   = static_cast<uint8_t>(clampf(alpha_i * alpha_g * 255, 0, 255));
			tex[idx * 4 + 0] = tex[idx * 4 + 1] = tex[idx * 4 + 2] = tex[idx * 4 + 3] = alpha;
		}
	return VKV_OK;
}

// src/volume_render_subpass.cpp:221-249
int vkv_build_uniforms(const float *view, const float *proj, const float *node_transform, const float *image_transform, float clip_distance,
                       uint32_t image_width, uint32_t image_height, VkvExtent3D ve, VkvExtent3D me, VkvCameraUniform *cam, VkvRayCastUniform *rc,
                       VkvRayGen *rg)
{
	if (!view || !proj || !node_transform || !image_transform || !cam || !rc || !rg || !extent_ok(ve) || !extent_ok(me) || image_width == 0 ||
	    image_height == 0)
		return VKV_E_INVALID_ARGUMENT;
	const mat4 V(view), P(proj), N(node_transform), I(image_transform);
	const mat4 model         = N * I;
	const mat4 model_inv     = inverse(model);
	const mat4 view_proj_inv = inverse(P * V);
	std::memcpy(cam->camera_view, V.m, 64);
	std::memcpy(cam->camera_proj, P.m, 64);
	std::memcpy(cam->camera_view_proj_inv, view_proj_inv.m, 64);
	std::memcpy(cam->model, model.m, 64);
	std::memcpy(cam->model_inv, model_inv.m, 64);

	const mat4 model_to_tex  = translate(vec3{0.5f, 0.5f, 0.5f});
	const mat4 global_to_tex = model_to_tex * model_inv;
	const mat4 view_inv      = inverse(V);
	const vec3 cam_pos_global{view_inv.at(0, 3), view_inv.at(1, 3), view_inv.at(2, 3)};
	const vec4 cam_pos_model = model_inv * vec4{cam_pos_global.x, cam_pos_global.y, cam_pos_global.z, 1.0f};
	const vec4 cam_pos_tex   = model_to_tex * vec4{cam_pos_model.x, cam_pos_model.y, cam_pos_model.z, 1.0f};
	const vec4 cam_dir4      = view_inv * vec4{0, 0, -1, 0};
	const vec3 cam_dir{cam_dir4.x, cam_dir4.y, cam_dir4.z};
	const vec4 plane{cam_dir.x, cam_dir.y, cam_dir.z,
	                 -clip_distance - (cam_pos_global.x * cam_dir.x + cam_pos_global.y * cam_dir.y + cam_pos_global.z * cam_dir.z)};
	const vec4 plane_tex = inverse_transpose(global_to_tex) * plane;
	rc->plane[0] = plane.x, rc->plane[1] = plane.y, rc->plane[2] = plane.z, rc->plane[3] = plane.w;
	rc->plane_tex[0] = plane_tex.x, rc->plane_tex[1] = plane_tex.y, rc->plane_tex[2] = plane_tex.z, rc->plane_tex[3] = plane_tex.w;
	rc->camera_pos_tex[0] = cam_pos_tex.x, rc->camera_pos_tex[1] = cam_pos_tex.y, rc->camera_pos_tex[2] = cam_pos_tex.z, rc->camera_pos_tex[3] = cam_pos_tex.w;
	rc->front_index   = (plane_tex.x < 0 ? 1 : 0) + (plane_tex.y < 0 ? 2 : 0) + (plane_tex.z < 0 ? 4 : 0);
	rc->block_size[0] = (float) ((ve.width + me.width - 1) / me.width);
	rc->block_size[1] = (float) ((ve.height + me.height - 1) / me.height);
	rc->block_size[2] = (float) ((ve.depth + me.depth - 1) / me.depth);
	rc->block_size[3] = 0.0f;

	// Ray generator (replaces the rasteriser): in double precision from the same float matrices.  Unproject pixel-space
	// points (0,0), (1,0), (0,1) at two depths, express the direction in texture space and scale it to unit distance along
	// the view direction (plane_tex.xyz is that covector), which makes the direction affine in pixel coordinates.
	double PV[16], PVinv[16], Md[16], Minv[16];
	{
		const mat4 pv = P * V;        // product in float like the uniform above, inverted in double
		for (int i = 0; i < 16; ++i)
			PV[i] = pv.m[i], Md[i] = model.m[i];
		if (!invert4x4<double>(PV, PVinv) || !invert4x4<double>(Md, Minv))
			return VKV_E_INVALID_ARGUMENT;
	}
	auto mulv = [](const double *m, const double *v, double *r) {
		for (int i = 0; i < 4; ++i)
			r[i] = m[i] * v[0] + m[4 + i] * v[1] + m[8 + i] * v[2] + m[12 + i] * v[3];
	};
	double       dirs[3][3];
	const double pts[3][2] = {{0, 0}, {1, 0}, {0, 1}};
	for (int p = 0; p < 3; ++p)
	{
		const double nx = 2.0 * pts[p][0] / (double) image_width - 1.0, ny = 2.0 * pts[p][1] / (double) image_height - 1.0;
		const double c1[4] = {nx, ny, 1.0, 1.0}, c2[4] = {nx, ny, 0.25, 1.0};
		double       w1[4], w2[4], t1[4], t2[4];
		mulv(PVinv, c1, w1);
		mulv(PVinv, c2, w2);
		for (int i = 0; i < 3; ++i)
			w1[i] /= w1[3], w2[i] /= w2[3];
		w1[3] = w2[3] = 1.0;
		mulv(Minv, w1, t1);        // model space; the +0.5 translation cancels in the difference
		mulv(Minv, w2, t2);
		const double dx = t2[0] - t1[0], dy = t2[1] - t1[1], dz = t2[2] - t1[2];
		const double along = (double) plane_tex.x * dx + (double) plane_tex.y * dy + (double) plane_tex.z * dz;
		dirs[p][0] = dx / along, dirs[p][1] = dy / along, dirs[p][2] = dz / along;
	}
	for (int i = 0; i < 3; ++i)
	{
		rg->dir00[i] = (float) dirs[0][i];
		rg->ddx[i]   = (float) (dirs[1][i] - dirs[0][i]);
		rg->ddy[i]   = (float) (dirs[2][i] - dirs[0][i]);
	}
	rg->dir00[3] = rg->ddx[3] = rg->ddy[3] = 0.0f;
	return VKV_OK;
}

// ---- device entry points -----------------------------------------------------------------------

int vkv_gradient_map(vkv_ctx *ctx, const uint8_t *d_volume, uint8_t *d_gradient, VkvExtent3D extent, const VkvTransferFunctionUniform *tf, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_volume || !d_gradient || !tf || !extent_ok(extent))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "gradient_map: null pointer or zero extent");
	return launch_gradient_map(ctx, d_volume, d_gradient, extent, tf, (hipStream_t) stream);
}

int vkv_occupancy_map(vkv_ctx *ctx, const uint8_t *d_volume, const uint8_t *d_gradient, const uint8_t *d_tf, const VkvTransferFunctionUniform *tf,
                      VkvExtent3D extent, uint8_t *d_map, VkvExtent3D map_extent, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_volume || !d_tf || !tf || !d_map || !extent_ok(extent) || !map_extent_ok(extent, map_extent))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "occupancy_map: null pointer or bad extent");
	return launch_occupancy_map(ctx, d_volume, d_gradient, d_tf, tf, extent, d_map, map_extent, (hipStream_t) stream);
}

int vkv_distance_map(vkv_ctx *ctx, uint8_t *d_map, uint8_t *d_swap, VkvExtent3D map_extent, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_map || !d_swap || d_map == d_swap)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "distance_map: null or aliased buffers");
	return launch_distance_map(ctx, d_map, d_swap, map_extent, (hipStream_t) stream);
}

int vkv_distance_map_anisotropic(vkv_ctx *ctx, uint8_t *const d_maps[8], uint8_t *d_swap, VkvExtent3D map_extent, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_maps || !d_swap)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "distance_map_anisotropic: null pointer");
	for (int i = 0; i < 8; ++i)
		if (!d_maps[i] || d_maps[i] == d_swap)
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "distance_map_anisotropic: map %d null or aliasing swap", i);
	return launch_distance_map_anisotropic(ctx, d_maps, d_swap, map_extent, (hipStream_t) stream);
}

// src/compute_distance_map.cpp:65-101
int vkv_compute_distance_map(vkv_ctx *ctx, const uint8_t *d_volume, const uint8_t *d_gradient, const uint8_t *d_tf, const VkvTransferFunctionUniform *tf,
                             VkvExtent3D extent, uint8_t *const d_maps[8], uint8_t *d_swap, VkvExtent3D map_extent, int32_t skipping_type, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (skipping_type < VKV_SKIP_NONE || skipping_type > VKV_SKIP_ANISOTROPIC_DISTANCE || !d_maps)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "compute_distance_map: bad skipping_type or null maps");
	const bool aniso = skipping_type == VKV_SKIP_ANISOTROPIC_DISTANCE;
	const int  n     = aniso ? 8 : 1;
	int        rc    = vkv_occupancy_map(ctx, d_volume, d_gradient, d_tf, tf, extent, d_maps[n - 1], map_extent, stream);
	if (rc)
		return rc;
	if (aniso)
		return vkv_distance_map_anisotropic(ctx, d_maps, d_swap, map_extent, stream);
	if (skipping_type == VKV_SKIP_DISTANCE)
		return vkv_distance_map(ctx, d_maps[0], d_swap, map_extent, stream);
	return VKV_OK;        // None / Block use the raw 0/255 occupancy map (:96-99)
}

int vkv_occupied_voxel_count(vkv_ctx *ctx, const uint8_t *d_volume, const uint8_t *d_gradient, const VkvTransferFunctionUniform *tf, VkvExtent3D extent,
                             uint64_t *d_count, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_volume || !tf || !d_count || !extent_ok(extent))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "occupied_voxel_count: null pointer or zero extent");
	return launch_occupied_voxel_count(ctx, d_volume, d_gradient, tf, extent, d_count, (hipStream_t) stream);
}

// ---- loader (host side; the C++ class throws, the C ABI returns codes) ------------------------------------------
int vkv_load_header(const char *filename_header, VkvVolumeHeader *out)
{
	if (!filename_header || !out)
		return VKV_E_INVALID_ARGUMENT;
	try
	{
		const LoadVolume::Header h = LoadVolume::load_header(filename_header);
		std::memset(out, 0, sizeof(*out));
		out->extent = h.extent;
		out->voxel_size[0] = h.voxel_size.x, out->voxel_size[1] = h.voxel_size.y, out->voxel_size[2] = h.voxel_size.z;
		out->normalisation_range[0] = h.normalisation_range[0], out->normalisation_range[1] = h.normalisation_range[1];
		std::strncpy(out->type, h.type.c_str(), sizeof(out->type) - 1);
		std::strncpy(out->endianness, h.endianness.c_str(), sizeof(out->endianness) - 1);
		std::memcpy(out->image_transform, h.image_transform.m, sizeof(out->image_transform));
		return VKV_OK;
	}
	catch (const std::exception &)
	{
		return VKV_E_IO;
	}
}

int vkv_load_data(const char *filename_data, const VkvVolumeHeader *header, uint8_t *out_voxels, size_t out_bytes)
{
	if (!filename_data || !header || !out_voxels)
		return VKV_E_INVALID_ARGUMENT;
	try
	{
		LoadVolume::Header h;
		h.extent                 = header->extent;
		h.normalisation_range[0] = header->normalisation_range[0], h.normalisation_range[1] = header->normalisation_range[1];
		h.type       = std::string(header->type, strnlen(header->type, sizeof(header->type)));
		h.endianness = std::string(header->endianness, strnlen(header->endianness, sizeof(header->endianness)));
		const std::vector<uint8_t> v = LoadVolume::load_data(filename_data, h);
		if (v.size() != out_bytes)
			return VKV_E_INVALID_ARGUMENT;
		std::memcpy(out_voxels, v.data(), v.size());
		return VKV_OK;
	}
	catch (const std::runtime_error &e)
	{
		return std::string(e.what()) == "unsupported image data type" ? VKV_E_INVALID_ARGUMENT : VKV_E_IO;
	}
	catch (const std::exception &)
	{
		return VKV_E_IO;
	}
}

int vkv_convert_volume(vkv_ctx *ctx, const void *d_raw, int32_t type, int32_t big_endian, float range_min, float range_max, uint64_t n_voxels,
                       uint8_t *d_out, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_raw || !d_out)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "convert_volume: null pointer");
	if ((type == VKV_VOXEL_UINT16 || type == VKV_VOXEL_INT16) && (((uintptr_t) d_raw) & 1u))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "convert_volume: 16-bit input must be 2-byte aligned");
	return launch_convert_volume(ctx, d_raw, type, big_endian != 0, range_min, range_max, n_voxels, d_out, (hipStream_t) stream);
}

size_t vkv_packed_volume_bytes(VkvExtent3D e)
{
	if (!extent_ok(e))
		return 0;
	return packed_bytes(packed_dims((int) e.width, (int) e.height, (int) e.depth));
}

int vkv_pack_volume(vkv_ctx *ctx, const uint8_t *d_volume, const uint8_t *d_gradient, VkvExtent3D extent, void *d_packed, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_volume || !d_packed || !extent_ok(extent))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "pack_volume: null pointer or zero extent");
	if (((uintptr_t) d_packed & 255u) != 0)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "pack_volume: d_packed must be 256-byte aligned");
	return launch_pack_volume(ctx, d_volume, d_gradient, extent, d_packed, (hipStream_t) stream);
}

int vkv_transfer_function_tables(vkv_ctx *ctx, const uint8_t *d_tf, const VkvTransferFunctionUniform *tf, uint32_t *d_tables, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_tf || !d_tables)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "transfer_function_tables: null pointer");
	if (((uintptr_t) d_tf & 3u) != 0 || ((uintptr_t) d_tables & 3u) != 0)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "transfer_function_tables: pointers must be 4-byte aligned");
	return launch_tf_tables(ctx, d_tf, tf, d_tables, (hipStream_t) stream);
}

int vkv_transfer_function_bits(vkv_ctx *ctx, const uint8_t *d_tf, uint32_t *d_tables, void *stream)
{
	return vkv_transfer_function_tables(ctx, d_tf, nullptr, d_tables, stream);
}

// argument checks shared by vkv_render and vkv_render_batch
static int check_render_params(vkv_ctx *ctx, const VkvRenderParams *P)
{
	if (!P)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: null params");
	const VkvRenderOptions &o = P->options;
	if (o.depth_attachment && !P->d_in_depth)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: options.depth_attachment needs d_in_depth");
	if (o.test < VKV_TEST_NONE || o.test > VKV_TEST_NUM_TEXTURE_SAMPLES)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: bad test mode %d", o.test);
	if (o.skipping_type < VKV_SKIP_NONE || o.skipping_type > VKV_SKIP_ANISOTROPIC_DISTANCE)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: bad skipping_type %d", o.skipping_type);
	if (!extent_ok(P->volume_extent) || P->image_width == 0 || P->image_height == 0)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: zero extent");
	if (!P->d_volume || !P->d_transfer_function)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: null volume or transfer function");
	if (P->transfer_function.use_gradient && P->use_precomputed_gradient && !P->d_gradient)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: precomputed gradient requested but d_gradient is null");
	if (!(P->transfer_function.sampling_factor > 0.0f))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: sampling_factor must be positive");
	if (o.skipping_type != VKV_SKIP_NONE)
	{
		if (!map_extent_ok(P->volume_extent, P->map_extent))
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: bad map extent");
		if ((uint64_t) P->map_extent.width * P->map_extent.height * P->map_extent.depth > 0xffffffffull)
			return set_error(ctx, VKV_E_UNSUPPORTED, "render: distance maps with more than 2^32 cells are not supported");
		if (P->map_extent.width >= (1u << 24) || P->map_extent.height >= (1u << 24) || P->map_extent.depth >= (1u << 24))
			return set_error(ctx, VKV_E_UNSUPPORTED, "render: distance map axes of 2^24 cells or more are not supported");
		const int n = o.skipping_type == VKV_SKIP_ANISOTROPIC_DISTANCE ? 8 : 1;
		for (int i = 0; i < n; ++i)
			if (!P->d_distance_maps[i])
				return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: distance map %d is null", i);
	}
	const VkvTileSchedule &t = P->tiles;
	if (t.tile_width == 0 || t.tile_height == 0 || (t.tile_width % 16) || (t.tile_height % 16) || t.tile_stride == 0)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: tile size must be a positive multiple of 16 and tile_stride > 0");
	{
		const uint64_t tiles_x = (P->image_width + t.tile_width - 1) / t.tile_width, tiles_y = (P->image_height + t.tile_height - 1) / t.tile_height;
		const bool     whole   = t.rect.w == 0 || t.rect.h == 0;
		if (!whole && ((uint64_t) t.rect.x0 + t.rect.w > tiles_x || (uint64_t) t.rect.y0 + t.rect.h > tiles_y))
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: the schedule's tile rectangle runs past the image");
		if (t.fill_outside && !whole && (t.compact || t.tile_first != 0 || t.tile_stride != 1 || (uint64_t) t.tile_count != (uint64_t) t.rect.w * t.rect.h))
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: fill_outside needs image-indexed outputs and the whole rectangle in one launch (tile_first 0, tile_stride 1)");
		const uint64_t scheduled = whole ? tiles_x * tiles_y : (uint64_t) t.rect.w * t.rect.h;
		if (t.tile_count && (uint64_t) t.tile_first + (uint64_t) (t.tile_count - 1) * t.tile_stride >= scheduled)
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: tile schedule runs past the %s", whole ? "image" : "tile rectangle");
	}
	if (P->d_packed_volume && ((uintptr_t) P->d_packed_volume & 255u) != 0)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: d_packed_volume must be 256-byte aligned");
	if (!P->d_out_color && !P->d_out_rgba8 && !P->d_out_counts && !P->d_out_depth)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render: no output buffer");
	return VKV_OK;
}

// opacity-correction table keyed by the TF alpha byte (frag:283):
// lut[a] = clamp(voxel_alpha_factor * (1 - pow(1 - a/255, 1/sampling_factor)), 0, 1)
static void build_alpha_lut(const VkvTransferFunctionUniform &tf, float *lut)
{
	const float sf_inv = 1.0f / tf.sampling_factor;
	for (int a = 0; a < 256; ++a)
	{
		const float v = tf.voxel_alpha_factor * (1.0f - std::pow(1.0f - (float) a / 255.0f, sf_inv));
		lut[a]        = std::min(std::max(v, 0.0f), 1.0f);
	}
}

int vkv_render(vkv_ctx *ctx, const VkvRenderParams *P, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	const int   rc = check_render_params(ctx, P);
	if (rc != VKV_OK)
		return rc;
	float lut[256];
	build_alpha_lut(P->transfer_function, lut);
	return launch_render(ctx, P, lut, (hipStream_t) stream);
}

int vkv_render_batch(vkv_ctx *ctx, const VkvRenderParams *P, uint32_t count, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!P || count == 0 || count > VKV_MAX_BATCH)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render_batch: count must be 1 .. %d", VKV_MAX_BATCH);
	std::vector<float> luts((size_t) count * 256);
	for (uint32_t i = 0; i < count; ++i)
	{
		const int rc = check_render_params(ctx, &P[i]);
		if (rc != VKV_OK)
			return rc;
		const VkvRenderParams &a = P[i], &b = P[0];
		if (a.options.skipping_type != b.options.skipping_type || (a.options.early_ray_termination != 0) != (b.options.early_ray_termination != 0) ||
		    (a.transfer_function.use_gradient != 0) != (b.transfer_function.use_gradient != 0) ||
		    (a.use_precomputed_gradient != 0) != (b.use_precomputed_gradient != 0))
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "render_batch: frame %u needs a different kernel variant than frame 0", i);
		build_alpha_lut(a.transfer_function, luts.data() + (size_t) i * 256);
	}
	return launch_render_batch(ctx, P, count, luts.data(), (hipStream_t) stream);
}

// the tile rectangle a caller passed (NULL / empty: every tile of the image), checked against the image; false = it runs past the image
static bool resolve_rect(const VkvTileRect *rect, uint32_t image_width, uint32_t image_height, uint32_t tile_width, uint32_t tile_height, VkvTileRect &out)
{
	const uint32_t tiles_x = (image_width + tile_width - 1) / tile_width, tiles_y = (image_height + tile_height - 1) / tile_height;
	if (!rect || rect->w == 0 || rect->h == 0)
	{
		out = VkvTileRect{0u, 0u, tiles_x, tiles_y};
		return true;
	}
	out = *rect;
	return (uint64_t) rect->x0 + rect->w <= tiles_x && (uint64_t) rect->y0 + rect->h <= tiles_y;
}

int vkv_screen_tile_rect(const VkvRayCastUniform *ray_cast, const VkvRayGen *ray_gen, uint32_t image_width, uint32_t image_height, uint32_t tile_width,
                         uint32_t tile_height, uint32_t align_tiles, VkvTileRect *out_rect)
{
	if (!ray_cast || !ray_gen || !out_rect || !image_width || !image_height || !tile_width || !tile_height)
		return VKV_E_INVALID_ARGUMENT;
	screen_tile_rect(ray_cast, ray_gen, image_width, image_height, tile_width, tile_height, align_tiles, out_rect);
	return VKV_OK;
}

int vkv_scatter_tiles(vkv_ctx *ctx, const void *d_gathered, void *d_image, uint32_t image_width, uint32_t image_height, uint32_t tile_width,
                      uint32_t tile_height, const VkvTileRect *rect, uint32_t n_ranks, uint32_t rank_stride_tiles, uint32_t bytes_per_pixel, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_gathered || !d_image || !image_width || !image_height || !tile_width || !tile_height || !n_ranks)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "scatter_tiles: null pointer or zero size");
	VkvTileRect r;
	if (!resolve_rect(rect, image_width, image_height, tile_width, tile_height, r))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "scatter_tiles: the tile rectangle runs past the image");
	if ((uint64_t) rank_stride_tiles * n_ranks < (uint64_t) r.w * r.h)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "scatter_tiles: gathered buffer holds fewer tiles than the rectangle");
	return launch_scatter_tiles_frames(ctx, 1u, &d_image, &d_gathered, &r, &rank_stride_tiles, image_width, image_height, tile_width, tile_height, n_ranks,
	                                   bytes_per_pixel, (hipStream_t) stream);
}

// ---- RCCL, resolved at run time (the library is not a link-time dependency of the product) ---------------------------------
extern "C++" {
namespace
{
typedef int (*nccl_gather_fn)(const void *, void *, size_t, int /* ncclDataType_t */, int, void * /* ncclComm_t */, hipStream_t);
typedef const char *(*nccl_error_fn)(int);
typedef int (*nccl_group_fn)(void);
struct Rccl
{
	void *         handle = nullptr;
	nccl_gather_fn gather = nullptr;
	nccl_error_fn  error  = nullptr;
	nccl_group_fn  group_start = nullptr, group_end = nullptr;
	bool           tried  = false;
	std::string    why;        // the loader's message when no library could be opened (dlerror() clears itself: captured once)
};
Rccl       g_rccl;
std::mutex g_rccl_mutex;

const Rccl &rccl()
{
	std::lock_guard<std::mutex> lock(g_rccl_mutex);
	if (g_rccl.tried)
		return g_rccl;
	g_rccl.tried = true;
	const char *override_path = std::getenv("VKV_RCCL_LIBRARY");
	void *      h             = nullptr;
	if (override_path && override_path[0])
		h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
	// the copy the process already uses (the application's, or the one inside PyTorch): communicator and call must come from the same library
	for (const char *name : {"librccl.so.1", "librccl.so"})
		if (!h)
			h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
	for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
		if (!h)
		{
			h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (!h)
			{
				const char *msg = dlerror();
				g_rccl.why += (g_rccl.why.empty() ? "" : "; ");
				g_rccl.why += msg ? msg : name;
			}
		}
	if (h && !dlsym(h, "ncclGather"))
		g_rccl.why = "the RCCL library that was found does not export ncclGather";
	if (h)
	{
		g_rccl.handle = h;
		g_rccl.gather = reinterpret_cast<nccl_gather_fn>(dlsym(h, "ncclGather"));
		g_rccl.error  = reinterpret_cast<nccl_error_fn>(dlsym(h, "ncclGetErrorString"));
		g_rccl.group_start = reinterpret_cast<nccl_group_fn>(dlsym(h, "ncclGroupStart"));
		g_rccl.group_end   = reinterpret_cast<nccl_group_fn>(dlsym(h, "ncclGroupEnd"));
	}
	return g_rccl;
}
}        // namespace
}        // extern "C++"

int vkv_gather_tiles(vkv_ctx *ctx, const void *d_tiles, void *d_gathered, size_t bytes_per_rank, int32_t root, void *nccl_comm, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_tiles || !nccl_comm || root < 0)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "gather_tiles: null buffer / communicator or negative root");
	if (bytes_per_rank == 0)
		return VKV_OK;
	const Rccl &r = rccl();
	if (!r.gather)
		return set_error(ctx, VKV_E_UNSUPPORTED, "gather_tiles: no RCCL library with ncclGather could be loaded (%s)", r.why.empty() ? "librccl.so.1" : r.why.c_str());
	const int rc = r.gather(d_tiles, d_gathered, bytes_per_rank, 0 /* ncclInt8 / ncclChar */, root, nccl_comm, (hipStream_t) stream);
	if (rc != 0)
		return set_error(ctx, 1000 + rc, "gather_tiles: ncclGather: %s", r.error ? r.error(rc) : "error");
	return VKV_OK;
}

int vkv_assemble_frame(vkv_ctx *ctx, const void *d_tiles, void *d_gathered, void *d_image, uint32_t image_width, uint32_t image_height, uint32_t tile_width,
                       uint32_t tile_height, const VkvTileRect *rect, uint32_t n_ranks, uint32_t rank, uint32_t bytes_per_pixel, int32_t root, void *nccl_comm,
                       void *stream)
{
	return vkv_assemble_frames(ctx, d_tiles, d_gathered, &d_image, 1u, image_width, image_height, tile_width, tile_height, rect, n_ranks, rank, bytes_per_pixel, root, nullptr,
	                           nccl_comm, stream);
}

int vkv_assemble_frames(vkv_ctx *ctx, const void *d_tiles, void *d_gathered, void *const *d_images, uint32_t frames, uint32_t image_width, uint32_t image_height,
                        uint32_t tile_width, uint32_t tile_height, const VkvTileRect *rects, uint32_t n_ranks, uint32_t rank, uint32_t bytes_per_pixel, int32_t root,
                        const int32_t *roots, void *nccl_comm, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	if (n_ranks == 0 || rank >= n_ranks || frames == 0 || frames > VKV_MAX_BATCH)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "assemble_frames: bad rank / n_ranks, or frames not in 1 .. %d", VKV_MAX_BATCH);
	if (!image_width || !image_height || !tile_width || !tile_height || (bytes_per_pixel != 4 && bytes_per_pixel != 16))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "assemble_frames: zero size, or bytes_per_pixel not 4 or 16");
	// per frame: rectangle, owner, tiles per rank, where the frame starts in a rank's block
	VkvTileRect rect[VKV_MAX_BATCH];
	int32_t     owner[VKV_MAX_BATCH];
	uint32_t    tpr[VKV_MAX_BATCH];
	uint64_t    off[VKV_MAX_BATCH + 1];
	const bool  one_owner = roots == nullptr;        // the caller's choice: one gather of the whole block to `root`, or a group of gathers, one per frame
	bool        mine      = false;
	off[0] = 0;
	for (uint32_t f = 0; f < frames; ++f)
	{
		if (!resolve_rect(rects ? &rects[f] : nullptr, image_width, image_height, tile_width, tile_height, rect[f]))
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "assemble_frames: the tile rectangle of frame %u runs past the image", f);
		owner[f] = roots ? roots[f] : root;
		if (owner[f] < 0 || (uint32_t) owner[f] >= n_ranks)
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "assemble_frames: bad root %d of frame %u", owner[f], f);
		const uint64_t tiles = (uint64_t) rect[f].w * rect[f].h;
		tpr[f]     = (uint32_t) ((tiles + n_ranks - 1) / n_ranks);
		off[f + 1] = off[f] + tpr[f];
		if ((uint32_t) owner[f] == rank)
		{
			mine = true;
			if (!d_images || !d_images[f])
				return set_error(ctx, VKV_E_INVALID_ARGUMENT, "assemble_frames: d_images[%u] is null on the frame's owner", f);
		}
	}
	if (off[frames] * n_ranks > 0xffffffffull)
		return set_error(ctx, VKV_E_UNSUPPORTED, "assemble_frames: too many tiles for one exchange");
	if (mine && !d_gathered)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "assemble_frames: a rank that owns a frame needs d_gathered");
	const size_t   tile_bytes = (size_t) tile_width * tile_height * bytes_per_pixel;
	const uint8_t *tiles_b    = static_cast<const uint8_t *>(d_tiles);
	uint8_t *      gath_b     = static_cast<uint8_t *>(d_gathered);
	const void *   src[VKV_MAX_BATCH];
	uint32_t       stride[VKV_MAX_BATCH];
	if (one_owner)
	{
		// ONE collective for the whole launch: [frame][tiles] of every rank -> [rank][frame][tiles] on the owner
		const int rc = vkv_gather_tiles(ctx, d_tiles, d_gathered, (size_t) off[frames] * tile_bytes, owner[0], nccl_comm, stream);
		if (rc != VKV_OK)
			return rc;
		for (uint32_t f = 0; f < frames; ++f)
			src[f] = gath_b + (size_t) off[f] * tile_bytes, stride[f] = (uint32_t) off[frames];
	}
	else
	{
		// owners spread over the ranks: one gather per frame, all of them in ONE group (RCCL fuses the group's point-to-point transfers: every
		// owner receives at the same time over its own inbound links); frame f arrives as [rank][tpr(f) tiles] at tile n_ranks * off[f]
		if (!d_tiles || !nccl_comm)
			return set_error(ctx, VKV_E_INVALID_ARGUMENT, "assemble_frames: null buffer / communicator");
		const Rccl &r = rccl();
		if (!r.gather)
			return set_error(ctx, VKV_E_UNSUPPORTED, "assemble_frames: no RCCL library with ncclGather could be loaded (%s)", r.why.empty() ? "librccl.so.1" : r.why.c_str());
		DeviceGuard guard(ctx->device);
		const bool  grouped = r.group_start && r.group_end;
		int         rc      = grouped ? r.group_start() : 0;
		for (uint32_t f = 0; f < frames && rc == 0; ++f)
			if (tpr[f])
				rc = r.gather(tiles_b + (size_t) off[f] * tile_bytes, (uint32_t) owner[f] == rank ? gath_b + (size_t) n_ranks * off[f] * tile_bytes : nullptr,
				              (size_t) tpr[f] * tile_bytes, 0 /* ncclInt8 / ncclChar */, owner[f], nccl_comm, (hipStream_t) stream);
		if (grouped)
		{
			const int rc2 = r.group_end();        // (always closed: an open group would swallow the caller's next collective)
			rc = rc ? rc : rc2;
		}
		if (rc != 0)
			return set_error(ctx, 1000 + rc, "assemble_frames: ncclGather group: %s", r.error ? r.error(rc) : "error");
		for (uint32_t f = 0; f < frames; ++f)
			src[f] = gath_b + (size_t) n_ranks * off[f] * tile_bytes, stride[f] = tpr[f];
	}
	if (!mine)
		return VKV_OK;
	// ONE de-interleave kernel for the frames this rank owns
	void *      img[VKV_MAX_BATCH];
	uint32_t    n = 0;
	for (uint32_t f = 0; f < frames; ++f)
		if ((uint32_t) owner[f] == rank)
			img[n] = d_images[f], src[n] = src[f], rect[n] = rect[f], stride[n] = stride[f], ++n;
	DeviceGuard guard(ctx->device);
	return launch_scatter_tiles_frames(ctx, n, img, src, rect, stride, image_width, image_height, tile_width, tile_height, n_ranks, bytes_per_pixel, (hipStream_t) stream);
}

int vkv_prepare_render(vkv_ctx *ctx, const VkvRenderParams *P, uint32_t count, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!P || count == 0 || count > VKV_MAX_BATCH)
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "prepare_render: count must be 1 .. %d", VKV_MAX_BATCH);
	for (uint32_t i = 0; i < count; ++i)
	{
		const int rc = check_render_params(ctx, &P[i]);
		if (rc != VKV_OK)
			return rc;
	}
	return prepare_render(ctx, P, count, (hipStream_t) stream);
}

int vkv_synth_volume(vkv_ctx *ctx, uint8_t *d_volume, VkvExtent3D extent, uint32_t kind, uint32_t seed, void *stream)
{
	if (!ctx)
		return VKV_E_INVALID_ARGUMENT;
	DeviceGuard guard(ctx->device);
	if (!d_volume || !extent_ok(extent))
		return set_error(ctx, VKV_E_INVALID_ARGUMENT, "synth_volume: null pointer or zero extent");
	return launch_synth_volume(ctx, d_volume, extent, kind, seed, (hipStream_t) stream);
}

}        // extern "C"
