// ORACLE — test infrastructure only: see ../opencv.hpp
#pragma once
#include "../opencv.hpp"
