// ORACLE — test infrastructure only.  Force-included (-include) ahead of every reference .cu compiled by
// oracle/ref_build.py: the real NvInfer.h drags the CUDA runtime in, and nvcc's headers leak <cfloat>/<cmath>;
// the reference relies on both.  thrust::cuda::par is spelled thrust::hip::par in rocThrust.
#pragma once
#include "cuda_runtime_api.h"

#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>

#include <thrust/execution_policy.h>
#include <thrust/system/hip/execution_policy.h>
namespace thrust {
namespace cuda = ::thrust::hip;
}
