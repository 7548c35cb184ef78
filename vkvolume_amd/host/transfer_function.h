// transfer_function.h — TransferFunctionUniform (reference: src/transfer_function.h:20-32), kept as the same plain POD.
#pragma once

#include "../../include/vkvolume_amd.h"

// Field-for-field the reference struct (VkBool32 use_gradient is a uint32_t); 32 bytes.
using TransferFunctionUniform = VkvTransferFunctionUniform;
static_assert(sizeof(TransferFunctionUniform) == 32, "TransferFunctionUniform must keep the reference's 32-byte layout");
