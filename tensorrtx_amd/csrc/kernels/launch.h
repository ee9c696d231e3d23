// Kernel launch through the thread's LaunchProbe when one is set (kernels.h), plain hipLaunchKernelGGL otherwise.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include "kernels.h"

#define TRTX_LAUNCH(kernel, grid, block, lds, stream, ...)                                                              \
    do {                                                                                                                \
        ::trtx::LaunchProbe* pr__ = ::trtx::conv_launch_probe();                                                        \
        if (pr__ && pr__->launches == 0 && pr__->start && pr__->stop)                                                   \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, pr__->start, pr__->stop, 0, __VA_ARGS__);           \
        else                                                                                                            \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                          \
        if (pr__) ++pr__->launches;                                                                                     \
    } while (0)
