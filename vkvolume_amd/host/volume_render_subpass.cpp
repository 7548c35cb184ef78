#include "volume_render_subpass.h"

#include <cstring>
#include <stdexcept>

VolumeRenderSubpass::VolumeRenderSubpass(DeviceContext &device_context, std::vector<Volume *> volumes_, Camera &cam, Options options_) :
    dc(device_context), camera(cam), volumes(std::move(volumes_)), options(options_)
{}

void VolumeRenderSubpass::prepare() {}

VkvRenderParams VolumeRenderSubpass::make_params(Volume &volume, const RenderTarget &target, const VkvTileSchedule *tiles, bool blend) const
{
	VkvRenderParams p;
	std::memset(&p, 0, sizeof(p));
	const VkvExtent3D volume_extent = volume.get_volume().extent;
	const VkvExtent3D map_extent    = volume.get_distance_map_swap().extent;
	if (vkv_build_uniforms(camera.view.data(), camera.projection.data(), volume.node_transform.data(), volume.get_image_transform().data(),
	                       options.clip_distance, target.width, target.height, volume_extent, map_extent, &p.camera, &p.ray_cast, &p.ray_gen) != VKV_OK)
		throw std::runtime_error("VolumeRenderSubpass: bad uniforms");
	p.transfer_function             = volume.get_transfer_function_uniform();
	p.options.skipping_type         = static_cast<int32_t>(options.skipping_type);
	p.options.clip_distance         = options.clip_distance;
	p.options.early_ray_termination = options.early_ray_termination;
	p.options.depth_attachment      = options.depth_attachment;
	p.options.test                  = static_cast<int32_t>(options.test);
	p.use_precomputed_gradient      = volume.options.use_precomputed_gradient;
	p.image_width = target.width, p.image_height = target.height;
	if (tiles)
		p.tiles = *tiles;
	else
	{
		const uint32_t t = 16;
		p.tiles          = VkvTileSchedule{t, t, 0, 1, ((target.width + t - 1) / t) * ((target.height + t - 1) / t), 0};
	}
	p.volume_extent = volume_extent, p.map_extent = map_extent;
	p.d_volume            = volume.get_volume().data;
	p.d_gradient          = volume.options.use_precomputed_gradient ? volume.get_gradient().data : nullptr;
	p.d_transfer_function = volume.get_transfer_function().data;
	for (size_t i = 0; i < volume.get_number_of_distance_maps() && i < 8; ++i)
		p.d_distance_maps[i] = volume.get_distance_map(i).data;
	p.d_packed_volume          = volume.get_packed_volume();
	p.d_transfer_function_bits = volume.get_transfer_function_bits();
	p.d_out_color = target.color, p.d_out_rgba8 = target.rgba8, p.d_out_counts = target.counts, p.d_out_depth = target.depth;
	p.d_in_depth        = target.in_depth;
	p.blend_over_target = blend ? 1u : 0u;
	return p;
}

VkvTileSchedule VolumeRenderSubpass::rank_schedule(Volume &volume, const RenderTarget &target, uint32_t rank, uint32_t n_ranks, uint32_t align_tiles) const
{
	const VkvRenderParams p = make_params(volume, target, nullptr);
	const uint32_t        t = 16;
	VkvTileRect           rect{};
	if (n_ranks == 0 || vkv_screen_tile_rect(&p.ray_cast, &p.ray_gen, target.width, target.height, t, t, align_tiles, &rect) != VKV_OK)
		throw std::runtime_error("VolumeRenderSubpass: bad rank schedule");
	const uint32_t total = rect.w * rect.h;
	return VkvTileSchedule{t, t, rank, n_ranks, total > rank ? (total - rank + n_ranks - 1) / n_ranks : 0u, 1u, rect};
}

VkvTileSchedule VolumeRenderSubpass::frame_schedule(Volume &volume, const RenderTarget &target) const
{
	VkvTileSchedule s = rank_schedule(volume, target, 0, 1);
	s.compact = 0, s.fill_outside = 1;
	return s;
}

void VolumeRenderSubpass::prepare_targets(const std::vector<RenderTarget> &targets, const VkvTileSchedule *tiles, bool for_batch)
{
	const bool own_schedule = !tiles && for_batch && volumes.size() == 1;        // what draw_batch(targets) will pass
	for (const RenderTarget &t : targets)
	{
		for (Volume *volume : volumes)
		{
			if (!volume->get_packed_volume())
				volume->pack(dc);
			const VkvTileSchedule fs = own_schedule ? frame_schedule(*volume, t) : VkvTileSchedule{};
			const VkvRenderParams p  = make_params(*volume, t, own_schedule ? &fs : tiles, false);
			if (vkv_prepare_render(dc.ctx, &p, 1, dc.stream) != VKV_OK)
				throw std::runtime_error(std::string("VolumeRenderSubpass::prepare_targets: ") + vkv_last_error(dc.ctx));
			const void *id = p.d_out_rgba8 ? (const void *) p.d_out_rgba8 : (const void *) p.d_out_color;
			if (id && vkv_register_target(dc.ctx, id, p.image_width, p.image_height, &p.tiles) != VKV_OK)
				throw std::runtime_error(std::string("VolumeRenderSubpass::prepare_targets: ") + vkv_last_error(dc.ctx));
		}
	}
}

void VolumeRenderSubpass::forget_targets(const std::vector<RenderTarget> &targets)
{
	for (const RenderTarget &t : targets)
		if (const void *id = t.rgba8 ? (const void *) t.rgba8 : (const void *) t.color)
			(void) vkv_forget_target(dc.ctx, id);
}

void VolumeRenderSubpass::draw(const RenderTarget &target, const VkvTileSchedule *tiles)
{
	bool blend = target.blend;
	for (Volume *volume : volumes)
	{
		if (!volume->get_packed_volume())
			volume->pack(dc);        // no gradient pass ran (gradient_test / no-gradient TF)
		const VkvRenderParams p = make_params(*volume, target, tiles, blend);
		blend                   = true;        // every further volume is blended onto the result (volume_render_subpass.cpp:176-181, :219)
		if (vkv_render(dc.ctx, &p, dc.stream) != VKV_OK)
			throw std::runtime_error(std::string("VolumeRenderSubpass::draw: ") + vkv_last_error(dc.ctx));
	}
}

void VolumeRenderSubpass::draw_batch(const std::vector<RenderTarget> &targets, const VkvTileSchedule *tiles)
{
	if (volumes.size() != 1 || targets.empty() || targets.size() > VKV_MAX_BATCH)
	{
		for (const RenderTarget &t : targets)
			draw(t, tiles);
		return;
	}
	Volume *volume = volumes.front();
	if (!volume->get_packed_volume())
		volume->pack(dc);
	std::vector<VkvRenderParams> params;
	params.reserve(targets.size());
	for (const RenderTarget &t : targets)
	{
		const VkvTileSchedule fs = tiles ? *tiles : frame_schedule(*volume, t);
		params.push_back(make_params(*volume, t, &fs, t.blend));
	}
	if (vkv_render_batch(dc.ctx, params.data(), (uint32_t) params.size(), dc.stream) != VKV_OK)
		throw std::runtime_error(std::string("VolumeRenderSubpass::draw_batch: ") + vkv_last_error(dc.ctx));
}
