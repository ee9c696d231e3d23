/* Forwarding header: TensorRT splits its API over several headers (the reference includes this one from
 * lenet/logging.h:29, resnet/logging.h:20, yolov8/include/logging.h); the MI355X shim keeps everything in NvInfer.h. */
#pragma once
#include "NvInfer.h"
