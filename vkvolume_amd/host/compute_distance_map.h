// compute_distance_map.h — ComputeDistanceMap (reference: src/compute_distance_map.h:32-51): occupancy map from the
// transfer function, then the isotropic or 8-octant Chebyshev transform.
#pragma once

#include "volume_component.h"
#include "volume_render_subpass.h"

class ComputeDistanceMap
{
  public:
	explicit ComputeDistanceMap(DeviceContext &device_context) : dc(device_context) {}
	virtual ~ComputeDistanceMap() = default;

	// src/compute_distance_map.cpp:65-101
	void compute(Volume &volume, const TransferFunctionUniform &transfer_function_uniform, VolumeRenderSubpass::SkippingType skipping_type);

  private:
	DeviceContext &dc;
};
