// vkv_offscreen.cpp — offscreen driver for the hot path: the call order of the reference application
// (VolumeRender::prepare / update_transfer_function, src/volume_render.cpp:163-238, :392-445) without the window,
// swap-chain, Sponza scene and GUI.  Accepts the reference's command-line flags (src/volume_render.h:46-56,
// scripts/benchmark.py:42-52) and prints the three log lines its harness parses (scripts/benchmark.py:55-59).
//
//   vkv_offscreen [--width=W --height=H] [--imin= --imax= --gmin= --gmax=] [--blocksize=B] [--skipmode=0..3]
//                 [--gradient_test] [--benchmark=FRAMES] [--synthetic=WxHxD[:kind[:seed]] | <volume file>]
//                 [--azimuth=DEG --elevation=DEG] [--dump-rgba8=file] [--dump-counts=file] [--dump-params=file]
//                 [--second-synthetic=WxHxD[:kind[:seed]] --second-offset=X,Y,Z --dump-params2=file]   a second volume in the same
//                                  subpass (VolumeRenderSubpass::draw loops over its volumes, src/volume_render_subpass.cpp:219)
//                 [--reload]       load the volume twice into the same Volume object
//                 [--ert[=0|1]]    early ray termination on / off whatever the mode says (the benchmark mode of the reference switches it off)
//                 [--virtual-ranks=N --dump-assembled=file]   the multi-GPU decomposition on one GPU: rank r of N renders
//                                  VolumeRenderSubpass::rank_schedule(r, N) - its share of the tiles of the frame's screen rectangle - into a compact
//                                  buffer, vkv_scatter_tiles assembles the N buffers (what the frame's owner does behind ncclGather)
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "compute_distance_map.h"
#include "compute_gradient_map.h"
#include "compute_occupied_voxel_count.h"
#include "volume_render_subpass.h"

namespace
{

struct Args
{
	uint32_t    width = 1280, height = 720;
	float       imin = 0.1f, imax = 1.0f, gmin = 0.0f, gmax = 0.2f;        // src/volume_render.cpp:67-70
	uint32_t    blocksize = 4, skipmode = 2;                             // :71-80
	bool        gradient_test = false;
	int         benchmark     = 0;
	int         frames_in_flight = 8;        // benchmark mode: frames rendered concurrently (one vkv_render_batch launch), like the reference's swap-chain images
	std::string dataset;
	std::string synthetic;
	float       azimuth = 30.0f, elevation = 20.0f;
	std::string dump_rgba8, dump_counts, dump_params, dump_params2;
	std::string second_synthetic;
	float       second_offset[3] = {0.0f, 0.0f, 0.0f};
	bool        reload           = false;
	int         ert              = -1;        // --ert[=0|1]: -1 = the mode's own setting (benchmark mode: off)
	int         virtual_ranks    = 0;         // --virtual-ranks=N: this GPU plays N ranks in turn, --dump-assembled gets their assembled frame
	std::string dump_assembled;
};

bool flag(const char *arg, const char *name, std::string &value)
{
	const size_t n = std::strlen(name);
	if (std::strncmp(arg, name, n) != 0)
		return false;
	if (arg[n] == '=')
	{
		value = arg + n + 1;
		return true;
	}
	if (arg[n] == 0)
	{
		value.clear();
		return true;
	}
	return false;
}

Args parse(int argc, char **argv)
{
	Args        a;
	std::string v;
	for (int i = 1; i < argc; ++i)
	{
		const char *s = argv[i];
		if (flag(s, "--width", v)) a.width = (uint32_t) std::stoul(v);
		else if (flag(s, "--height", v)) a.height = (uint32_t) std::stoul(v);
		else if (flag(s, "--imin", v)) a.imin = std::stof(v);
		else if (flag(s, "--imax", v)) a.imax = std::stof(v);
		else if (flag(s, "--gmin", v)) a.gmin = std::stof(v);
		else if (flag(s, "--gmax", v)) a.gmax = std::stof(v);
		else if (flag(s, "--blocksize", v)) a.blocksize = (uint32_t) std::stoul(v);
		else if (flag(s, "--skipmode", v)) { const uint32_t m = (uint32_t) std::stoul(v); if (m <= 3) a.skipmode = m; }
		else if (flag(s, "--gradient_test", v)) a.gradient_test = true;
		else if (flag(s, "--benchmark", v)) a.benchmark = std::stoi(v);
		else if (flag(s, "--frames-in-flight", v)) a.frames_in_flight = std::max(1, std::stoi(v));
		else if (flag(s, "--synthetic", v)) a.synthetic = v;
		else if (flag(s, "--azimuth", v)) a.azimuth = std::stof(v);
		else if (flag(s, "--elevation", v)) a.elevation = std::stof(v);
		else if (flag(s, "--dump-rgba8", v)) a.dump_rgba8 = v;
		else if (flag(s, "--dump-counts", v)) a.dump_counts = v;
		else if (flag(s, "--dump-params2", v)) a.dump_params2 = v;
		else if (flag(s, "--dump-params", v)) a.dump_params = v;
		else if (flag(s, "--second-synthetic", v)) a.second_synthetic = v;
		else if (flag(s, "--second-offset", v))
		{
			if (std::sscanf(v.c_str(), "%f,%f,%f", &a.second_offset[0], &a.second_offset[1], &a.second_offset[2]) != 3)
				throw std::runtime_error("--second-offset=X,Y,Z");
		}
		else if (flag(s, "--reload", v)) a.reload = true;
		else if (flag(s, "--ert", v)) a.ert = v.empty() ? 1 : std::stoi(v);
		else if (flag(s, "--virtual-ranks", v)) a.virtual_ranks = std::stoi(v);
		else if (flag(s, "--dump-assembled", v)) a.dump_assembled = v;
		else if (s[0] != '-') a.dataset = s;
		else throw std::runtime_error(std::string("unknown flag ") + s);
	}
	return a;
}

template <typename T>
T *device_alloc(size_t n)
{
	void *p = nullptr;
	if (hipMalloc(&p, n * sizeof(T)) != hipSuccess)
		throw std::runtime_error("hipMalloc failed");
	return static_cast<T *>(p);
}

template <typename T>
void dump(const std::string &path, const T *device, size_t n, hipStream_t stream)
{
	std::vector<T> host(n);
	if (hipMemcpyAsync(host.data(), device, n * sizeof(T), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
		throw std::runtime_error("read-back failed");
	std::ofstream f(path, std::ios::binary);
	f.write(reinterpret_cast<const char *>(host.data()), (std::streamsize) (n * sizeof(T)));
}

double ms_since(std::chrono::steady_clock::time_point t0)
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}        // namespace

int main(int argc, char **argv)
{
	try
	{
		const Args args = parse(argc, argv);
		vkv_ctx *  ctx  = nullptr;
		if (vkv_create(0, &ctx) != VKV_OK)
			throw std::runtime_error("no gfx950 device");
		hipStream_t stream = nullptr;
		if (hipStreamCreate(&stream) != hipSuccess)
			throw std::runtime_error("hipStreamCreate failed");
		DeviceContext dc{ctx, stream};

		ComputeDistanceMap        compute_distance_map(dc);
		ComputeGradientMap        compute_gradient_map(dc);
		ComputeOccupiedVoxelCount compute_occupied_voxel_count(dc);

		const bool benchmark = args.benchmark > 0;
		VolumeRenderSubpass::Options render_options;
		render_options.skipping_type = static_cast<VolumeRenderSubpass::SkippingType>(args.skipmode);
		if (benchmark)
		{        // src/volume_render.cpp:177-183
			render_options.clip_distance         = 1.0f;
			render_options.early_ray_termination = false;
			render_options.test                  = VolumeRenderSubpass::Test::NumTextureSamples;
		}
		if (args.ert >= 0)        // --ert[=0|1]: early ray termination as asked for (the reference's benchmark mode switches it off)
			render_options.early_ray_termination = args.ert != 0;

		Volume volume(args.dataset.empty() ? "synthetic" : args.dataset);
		volume.options.intensity_min            = args.imin;
		volume.options.intensity_max            = args.imax;
		volume.options.gradient_min             = args.gmin;
		volume.options.gradient_max             = args.gmax;
		volume.options.use_precomputed_gradient = !args.gradient_test;
		if (!args.synthetic.empty())
		{
			unsigned w = 0, h = 0, d = 0, kind = 1, seed = 0xC0FFEE03u;
			if (std::sscanf(args.synthetic.c_str(), "%ux%ux%u:%u:%u", &w, &h, &d, &kind, &seed) < 3)
				throw std::runtime_error("--synthetic=WxHxD[:kind[:seed]]");
			volume.load_synthetic(dc, VkvExtent3D{w, h, d}, kind, seed, args.blocksize);
			if (args.reload)        // a Volume object is reusable: the second load replaces every resource of the first
				volume.load_synthetic(dc, VkvExtent3D{w, h, d}, kind, seed, args.blocksize);
			volume.set_image_transform(vkv::scale(vkv::vec3{(float) w, (float) h, (float) d}));
		}
		else if (!args.dataset.empty())
			volume.load_from_file(dc, args.dataset, args.blocksize);
		else
			throw std::runtime_error("give a volume file or --synthetic=WxHxD");

		// gradient map (src/volume_render.cpp:203-216)
		if (volume.options.use_precomputed_gradient)
		{
			const auto tf = volume.get_transfer_function_uniform();
			if (benchmark)
			{        // untimed first run: the first launch out of a code object pays for loading it (~14 ms), the reference's pipelines
				 // are built before its timers start
				compute_gradient_map.compute(volume, tf);
				(void) hipStreamSynchronize(stream);
			}
			const auto t0 = std::chrono::steady_clock::now();
			compute_gradient_map.compute(volume, tf);
			(void) hipStreamSynchronize(stream);
			std::printf("Updated gradient map in %gms\n", ms_since(t0));
		}

		// update_transfer_function (src/volume_render.cpp:392-445)
		{
			const auto tf = volume.get_transfer_function_uniform();
			if (benchmark)
			{
				uint64_t * buffer = compute_occupied_voxel_count.initialise_buffer(volume);
				const auto t0     = std::chrono::steady_clock::now();
				volume.update_transfer_function_texture(dc);
				compute_occupied_voxel_count.compute(volume, buffer, tf);
				const uint64_t n_occupied = compute_occupied_voxel_count.get_result(buffer);
				const auto &   e          = volume.get_volume().extent;
				const size_t   n_voxels   = (size_t) e.width * e.height * e.depth;
				std::printf("Occupied voxels: %g%% in %gms\n", 100.0f * (float) n_occupied / (float) n_voxels, ms_since(t0));
				compute_distance_map.compute(volume, tf, render_options.skipping_type);        // untimed first run (see above)
				(void) hipStreamSynchronize(stream);
				const int  runs = 5;
				const auto t1   = std::chrono::steady_clock::now();
				for (int i = 0; i < runs; ++i)
				{
					compute_distance_map.compute(volume, tf, render_options.skipping_type);
					(void) hipStreamSynchronize(stream);        // compute_submit() waits on a fence after every update
				}
				std::printf("Updated occupancy/distance map in %gms\n", ms_since(t1) / runs);
			}
			else
			{
				volume.update_transfer_function_texture(dc);
				compute_distance_map.compute(volume, tf, render_options.skipping_type);
			}
		}

		// node scale: longest physical edge -> 100 units (src/volume_render.cpp:224-238)
		{
			const vkv::mat4 &m = volume.get_image_transform();
			float            longest = 0.0f;
			for (int c = 0; c < 3; ++c)
				longest = std::max(longest, std::sqrt(m.at(0, c) * m.at(0, c) + m.at(1, c) * m.at(1, c) + m.at(2, c) * m.at(2, c)));
			const float s         = 100.0f / longest;
			volume.node_transform = vkv::scale(vkv::vec3{s, s, s});
		}

		// camera on an orbit around the volume centre (the reference's free camera pose comes from a glTF scene that is
		// not part of this path)
		Camera camera;
		{
			const vkv::mat4 model = volume.node_transform * volume.get_image_transform();
			float           diag2 = 0.0f;
			for (int c = 0; c < 3; ++c)
				diag2 += model.at(0, c) * model.at(0, c) + model.at(1, c) * model.at(1, c) + model.at(2, c) * model.at(2, c);
			const float radius = 1.5f * 0.5f * std::sqrt(diag2);
			const float az = vkv::radians(args.azimuth), el = vkv::radians(args.elevation);
			const vkv::vec3 eye{radius * std::cos(el) * std::sin(az), radius * std::sin(el), radius * std::cos(el) * std::cos(az)};
			camera.view       = vkv::look_at(eye, vkv::vec3{0, 0, 0}, vkv::vec3{0, 1, 0});
			camera.projection = vkv::perspective_vulkan(vkv::radians(60.0f), (float) args.width / (float) args.height, 0.1f, 1000.0f);
		}

		// optional second volume: same call order, its node translated; it is blended onto the first one's result
		Volume volume2("second");
		std::vector<Volume *> volumes{&volume};
		if (!args.second_synthetic.empty())
		{
			unsigned w = 0, h = 0, d = 0, kind = 1, seed = 0xC0FFEE03u;
			if (std::sscanf(args.second_synthetic.c_str(), "%ux%ux%u:%u:%u", &w, &h, &d, &kind, &seed) < 3)
				throw std::runtime_error("--second-synthetic=WxHxD[:kind[:seed]]");
			volume2.options = volume.options;
			volume2.load_synthetic(dc, VkvExtent3D{w, h, d}, kind, seed, args.blocksize);
			volume2.set_image_transform(vkv::scale(vkv::vec3{(float) w, (float) h, (float) d}));
			const auto tf2 = volume2.get_transfer_function_uniform();
			if (volume2.options.use_precomputed_gradient)
				compute_gradient_map.compute(volume2, tf2);
			volume2.update_transfer_function_texture(dc);
			compute_distance_map.compute(volume2, tf2, render_options.skipping_type);
			float longest = 0.0f;
			const vkv::mat4 &m = volume2.get_image_transform();
			for (int c = 0; c < 3; ++c)
				longest = std::max(longest, std::sqrt(m.at(0, c) * m.at(0, c) + m.at(1, c) * m.at(1, c) + m.at(2, c) * m.at(2, c)));
			const float s2         = 100.0f / longest;
			volume2.node_transform = vkv::translate(vkv::vec3{args.second_offset[0], args.second_offset[1], args.second_offset[2]}) * vkv::scale(vkv::vec3{s2, s2, s2});
			volumes.push_back(&volume2);
		}

		VolumeRenderSubpass subpass(dc, volumes, camera, render_options);
		subpass.prepare();
		RenderTarget target;
		target.width = args.width, target.height = args.height;
		const size_t n_pixels = (size_t) args.width * args.height;
		target.rgba8          = device_alloc<uint8_t>(n_pixels * 4);
		target.counts         = device_alloc<uint32_t>(n_pixels * 3);

		const int frames = benchmark ? args.benchmark : 1;
		subpass.prepare_targets({target});
		subpass.draw(target);        // warm-up (and the frame the dumps below read)
		(void) hipStreamSynchronize(stream);
		{
			// Benchmark frames go round-robin over a few HIP streams, each with its own colour target - what the reference gets from
			// its per-swap-chain-image command buffers: the long tail of one frame overlaps the start of the next.
			const int                fif = benchmark ? args.frames_in_flight : 1;
			std::vector<hipStream_t> streams(fif, stream);
			std::vector<RenderTarget> targets(fif, target);
			for (int i = 0; i < fif; ++i)
			{
				targets[i].counts = nullptr;        // the frag's counters are test-mode outputs, not part of a frame
				if (i > 0)
				{
					if (hipStreamCreate(&streams[i]) != hipSuccess)
						throw std::runtime_error("hipStreamCreate failed");
					targets[i].rgba8 = device_alloc<uint8_t>(n_pixels * 4);
				}
			}
			// the frames in flight share one launch; consecutive launches alternate over three streams (each with its own set of targets),
			// so the tail of one launch is covered by the next (bench.py --batch-streams)
			const bool                             batched = benchmark && volumes.size() == 1 && fif > 1 && fif <= VKV_MAX_BATCH;
			constexpr int                          kLaunchStreams = 3;
			std::vector<hipStream_t>               lstreams(kLaunchStreams, stream);
			std::vector<std::vector<RenderTarget>> ltargets(kLaunchStreams, targets);
			for (int s = 1; batched && s < kLaunchStreams; ++s)
			{
				if (hipStreamCreate(&lstreams[s]) != hipSuccess)
					throw std::runtime_error("hipStreamCreate failed");
				for (int i = 0; i < fif; ++i)
					ltargets[s][i].rgba8 = device_alloc<uint8_t>(n_pixels * 4);
			}
			// set-up: tables, scratch and feedback state of every (stream, target) pair, so that the timed loop only enqueues
			if (batched)
				for (int s = 0; s < kLaunchStreams; ++s)
				{
					dc.stream = lstreams[s];
					subpass.prepare_targets(ltargets[s], nullptr, true);
				}
			else
				for (int i = 0; i < fif; ++i)
				{
					dc.stream = streams[i];
					subpass.prepare_targets({targets[i]});
				}
			dc.stream     = stream;
			const auto t0 = std::chrono::steady_clock::now();
			if (batched)
			{
				int launch = 0;
				for (int f = 0; f < frames; f += fif, ++launch)
				{
					const int s = launch % kLaunchStreams;
					dc.stream   = lstreams[s];
					subpass.draw_batch(std::vector<RenderTarget>(ltargets[s].begin(), ltargets[s].begin() + std::min(fif, frames - f)));
				}
				for (int s = 1; s < kLaunchStreams; ++s)
					(void) hipStreamSynchronize(lstreams[s]);
			}
			else
				for (int f = 0; f < frames; ++f)
				{
					dc.stream = streams[f % fif];
					subpass.draw(benchmark ? targets[f % fif] : target);
				}
			for (int i = 0; i < fif; ++i)
				(void) hipStreamSynchronize(streams[i]);
			const double ms = ms_since(t0);
			dc.stream       = stream;
			std::printf("ran %d frames, averaged %g fps\n", frames, 1000.0 * frames / ms);
			for (int i = 1; i < fif; ++i)
			{
				subpass.forget_targets({targets[i]});
				(void) hipFree(targets[i].rgba8);
				(void) vkv_release_stream(ctx, streams[i]);        // all its work has completed (synchronised above)
				(void) hipStreamDestroy(streams[i]);
			}
			for (int s = 1; batched && s < kLaunchStreams; ++s)
			{
				subpass.forget_targets(ltargets[s]);
				for (int i = 0; i < fif; ++i)
					(void) hipFree(ltargets[s][i].rgba8);
				(void) vkv_release_stream(ctx, lstreams[s]);
				(void) hipStreamDestroy(lstreams[s]);
			}
		}

		if (!args.dump_params.empty())
		{        // the exact parameter block draw() handed to vkv_render (device pointers included), for the parity tests
			const VkvRenderParams p = subpass.make_params(volume, target, nullptr);
			std::ofstream         f(args.dump_params, std::ios::binary);
			f.write(reinterpret_cast<const char *>(&p), sizeof(p));
		}
		if (!args.dump_params2.empty() && volumes.size() > 1)
		{
			const VkvRenderParams p = subpass.make_params(volume2, target, nullptr, true);
			std::ofstream         f(args.dump_params2, std::ios::binary);
			f.write(reinterpret_cast<const char *>(&p), sizeof(p));
		}
		if (args.virtual_ranks > 0 && !args.dump_assembled.empty() && volumes.size() == 1)
		{
			const uint32_t N = (uint32_t) args.virtual_ranks, tile = 16;
			const VkvTileSchedule s0 = subpass.rank_schedule(volume, target, 0, N);        // every rank derives the same rectangle
			const uint32_t tiles_per_rank = (s0.rect.w * s0.rect.h + N - 1) / N;
			const size_t   rank_bytes     = (size_t) tiles_per_rank * tile * tile * 4;
			uint8_t *      d_gathered     = device_alloc<uint8_t>(rank_bytes * N);        // [rank][tiles]: what ncclGather delivers to the owner
			uint8_t *      d_image        = device_alloc<uint8_t>(n_pixels * 4);
			(void) hipMemsetAsync(d_image, 0x5a, n_pixels * 4, stream);
			for (uint32_t r = 0; r < N; ++r)
			{
				const VkvTileSchedule sr = subpass.rank_schedule(volume, target, r, N);
				RenderTarget          tr = target;
				tr.rgba8 = d_gathered + r * rank_bytes, tr.counts = nullptr;
				subpass.draw(tr, &sr);
			}
			if (vkv_scatter_tiles(ctx, d_gathered, d_image, args.width, args.height, tile, tile, &s0.rect, N, tiles_per_rank, 4, stream) != VKV_OK)
				throw std::runtime_error(std::string("vkv_scatter_tiles: ") + vkv_last_error(ctx));
			dump(args.dump_assembled, d_image, n_pixels * 4, stream);
			std::printf("assembled %u virtual ranks: rectangle %ux%u tiles at (%u, %u) of %ux%u, %u tiles per rank\n", N, s0.rect.w, s0.rect.h, s0.rect.x0, s0.rect.y0,
			            (args.width + tile - 1) / tile, (args.height + tile - 1) / tile, tiles_per_rank);
			(void) hipFree(d_gathered);
			(void) hipFree(d_image);
		}
		if (!args.dump_rgba8.empty())
			dump(args.dump_rgba8, target.rgba8, n_pixels * 4, stream);
		if (!args.dump_counts.empty())
			dump(args.dump_counts, target.counts, n_pixels * 3, stream);
		subpass.forget_targets({target});
		(void) hipFree(target.rgba8);
		(void) hipFree(target.counts);
		(void) hipStreamDestroy(stream);
		vkv_destroy(ctx);
		return 0;
	}
	catch (const std::exception &e)
	{
		std::fprintf(stderr, "vkv_offscreen: %s\n", e.what());
		return 1;
	}
}
