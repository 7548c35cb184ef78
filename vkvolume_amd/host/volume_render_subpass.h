// volume_render_subpass.h — VolumeRenderSubpass (reference: src/volume_render_subpass.h:55-101) as an offscreen pass:
// same Options / SkippingType / Test, same uniform structs, draw() renders every volume into caller-owned buffers.
#pragma once

#include <vector>

#include "volume_component.h"

using CameraUniform  = VkvCameraUniform;         // src/volume_render_subpass.h:32-39
using RayCastUniform = VkvRayCastUniform;        // src/volume_render_subpass.h:46-53

// vkb::sg::Camera stand-in: view matrix + (vulkan-style) projection
struct Camera
{
	vkv::mat4 view;
	vkv::mat4 projection;        // as vkb::vulkan_style_projection(camera.get_projection()) returns it
};

// Output images of one draw (device pointers; any may be null)
struct RenderTarget
{
	uint32_t  width = 0, height = 0;
	float *   color  = nullptr;        // RGBA32F, premultiplied
	uint8_t * rgba8  = nullptr;        // RGBA8
	uint32_t *counts = nullptr;        // 3 x u32 per pixel
	float *   depth  = nullptr;
	const float *in_depth = nullptr;        // scene depth for Options::depth_attachment (input attachment 0 of the reference subpass)
	bool         blend    = false;          // true: blend onto the existing contents of color / rgba8 (the subpass's blend state);
	                                        // false: the first volume overwrites the (cleared) target, further volumes blend onto it
};

class VolumeRenderSubpass
{
  public:
	enum class SkippingType : int
	{
		None                = 0,
		Block               = 1,
		Distance            = 2,
		AnisotropicDistance = 3
	};

	enum class Test : int
	{
		None              = 0,
		RayEntry          = 1,
		RayExit           = 2,
		NumTextureSamples = 3
	};

	struct Options
	{
		SkippingType skipping_type         = SkippingType::Distance;
		float        clip_distance         = 50.0f;
		bool         early_ray_termination = true;
		bool         depth_attachment      = false;
		Test         test                  = Test::None;
	};

	VolumeRenderSubpass(DeviceContext &device_context, std::vector<Volume *> volumes, Camera &camera, Options options);
	virtual ~VolumeRenderSubpass() = default;

	void prepare();        // nothing to pre-build: kernel variants are compiled ahead of time

	// Set-up for the targets this subpass will draw into on the context's current stream (the counterpart of the reference building its
	// per-swap-chain-image resources, src/volume_render_subpass.cpp:95-157): vkv_prepare_render + vkv_register_target, so that draw() /
	// draw_batch() afterwards only enqueue (no allocation, no wait).  forget_targets() before the buffers are freed.
	// for_batch: the targets will be drawn with draw_batch(targets) and no schedule of the caller's: registered for draw_batch's own schedule
	void prepare_targets(const std::vector<RenderTarget> &targets, const VkvTileSchedule *tiles = nullptr, bool for_batch = false);
	void forget_targets(const std::vector<RenderTarget> &targets);

	// src/volume_render_subpass.cpp:159-294: per volume, build the uniforms and march.  `tiles` selects the screen tiles of
	// this launch (null = the whole frame).
	void draw(const RenderTarget &target, const VkvTileSchedule *tiles = nullptr);

	// Several frames of the same subpass in ONE launch (vkv_render_batch): the reference keeps a few swap-chain images in flight
	// (per-image command buffers); here their ray-marches share a grid, so the long tail of one frame is covered by the others.
	// One volume, up to VKV_MAX_BATCH targets with distinct output buffers; falls back to draw() per target otherwise.
	// tiles == nullptr: every frame through frame_schedule() - the tiles of its screen rectangle, the rest of the frame filled by the same launch.
	void draw_batch(const std::vector<RenderTarget> &targets, const VkvTileSchedule *tiles = nullptr);

	// The schedule of a whole frame on one GPU (round 6): the 16x16 tiles of the rectangle the clipped box projects into (vkv_screen_tile_rect: the
	// rasteriser of the reference only shades the box's faces, :262-293) with VkvTileSchedule.fill_outside - the launch's workgroups write the
	// no-fragment result everywhere else, so the frame is complete without a workgroup per empty tile (C3: half of the frame's tiles, 2.7 % of its time).
	VkvTileSchedule frame_schedule(Volume &volume, const RenderTarget &target) const;

	// the parameter block of one volume (what draw() binds), exposed for tests / the multi-GPU driver
	VkvRenderParams make_params(Volume &volume, const RenderTarget &target, const VkvTileSchedule *tiles, bool blend = false) const;

	// Multi-GPU: the schedule of rank `rank` of `n_ranks` for this camera - the 16x16 tiles of the rectangle the clipped box
	// of `volume` projects into (vkv_screen_tile_rect: the counterpart of the rasteriser only shading the box's faces, :262-293), dealt round-robin,
	// compact outputs.  Every rank derives the same rectangle from the same uniforms; pass it to draw() / draw_batch() and, with `rect`, to
	// vkv_assemble_frame(s).  align_tiles > 1 rounds the rectangle outwards (fewer distinct schedules for a camera that moves).
	VkvTileSchedule rank_schedule(Volume &volume, const RenderTarget &target, uint32_t rank, uint32_t n_ranks, uint32_t align_tiles = 1) const;

  private:
	DeviceContext &       dc;
	Camera &              camera;
	std::vector<Volume *> volumes;
	Options               options;
};
