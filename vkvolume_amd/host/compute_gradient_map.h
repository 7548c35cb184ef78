// compute_gradient_map.h — ComputeGradientMap (reference: src/compute_gradient_map.h:32-47): one pass that writes the
// gradient-magnitude volume.  The Vulkan dispatch becomes vkv_gradient_map().
#pragma once

#include "volume_component.h"

class ComputeGradientMap
{
  public:
	explicit ComputeGradientMap(DeviceContext &device_context) : dc(device_context) {}
	virtual ~ComputeGradientMap() = default;

	// src/compute_gradient_map.cpp:57-81.  Also rebuilds the packed sampling image, which interleaves the gradient bytes.
	void compute(Volume &volume, const TransferFunctionUniform &transfer_function_uniform);

  private:
	DeviceContext &dc;
};
