// raymarch_s0e0.hip — the ray-march kernels and launchers of one (skipping type, early ray termination) pair: see raymarch_inst.hpp.
#define VKV_RAYMARCH_INSTANTIATE
#include "raymarch_inst.hpp"

template struct vkv::RayMarchLaunchers<VKV_SKIP_NONE, false>;
