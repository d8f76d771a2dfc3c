// TEST INFRASTRUCTURE ONLY: shadows kernels/quantization/gptq/compat.cuh (atomicCAS-based half atomics for old GPUs)
// when the reference GPTQ kernels are compiled for the host; cuda_host_shim.h already provides atomicAdd.
#pragma once
#include "cuda_host_shim.h"
