#include "compute_gradient_map.h"

#include <stdexcept>

void ComputeGradientMap::compute(Volume &volume, const TransferFunctionUniform &tf)
{
	const auto &vol = volume.get_volume();
	const auto &grd = volume.get_gradient();
	if (vkv_gradient_map(dc.ctx, vol.data, grd.data, vol.extent, &tf, dc.stream) != VKV_OK)
		throw std::runtime_error(std::string("ComputeGradientMap: ") + vkv_last_error(dc.ctx));
	volume.pack(dc);
}
