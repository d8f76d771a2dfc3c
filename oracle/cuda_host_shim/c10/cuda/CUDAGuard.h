// TEST INFRASTRUCTURE ONLY: stands in for <c10/cuda/CUDAGuard.h> when the reference GPTQ kernels are compiled for the host.
#pragma once
#include "../../cuda_host_shim.h"
