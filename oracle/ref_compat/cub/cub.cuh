// ORACLE — test infrastructure only: cub:: -> hipcub:: for the unmodified reference sources (see ../cuda_runtime_api.h)
#pragma once
#include <hipcub/hipcub.hpp>
namespace cub = hipcub;
