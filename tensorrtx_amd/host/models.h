// Model builders written against the nvinfer1-shaped API (include/NvInfer.h), mirroring the reference's
// per-model builder functions.  Compile-time constants of the reference (kInputH/W, kBatchSize, kNumClass,
// USE_FP16 ... yolov8/include/config.h:1-31) are run-time arguments here so one binary serves batch 32,
// 1280x1280, etc. (SURVEY.md §5 "Config / flags").
#pragma once
#include <string>

#include "NvInfer.h"

namespace trtx_host {

// lenet/lenet.cpp:36-155
nvinfer1::IHostMemory* buildLenet(nvinfer1::IBuilder* builder, nvinfer1::IBuilderConfig* config, const std::string& wts,
                                  int32_t N);
// resnet/resnet50.cpp:155-229
nvinfer1::IHostMemory* buildResnet50(nvinfer1::IBuilder* builder, nvinfer1::IBuilderConfig* config, const std::string& wts,
                                     int maxBatch, bool fp16, int H = 224, int W = 224);

// retinaface/retina_r50.cpp:100-242
nvinfer1::IHostMemory* buildRetinaFaceR50(nvinfer1::IBuilder* builder, nvinfer1::IBuilderConfig* config, const std::string& wts,
                                          int maxBatch, bool fp16, int H = 480, int W = 640);

struct Yolov8Config {
    int input_h = 640, input_w = 640;   // kInputH / kInputW
    int num_class = 80;                 // kNumClass
    int max_batch = 1;                  // kBatchSize
    int max_out_bbox = 1000;            // kMaxNumOutputBbox
    bool fp16 = true;                   // USE_FP16
    float gd = 0.33f, gw = 0.25f;       // 'n' scale (yolov8_det.cpp:130-133)
    int max_channels = 1024;
    bool mark_heads = false;            // debugging: also expose the three plugin inputs as outputs "head0..2"
};
// yolov8/src/model.cpp:98-336
nvinfer1::IHostMemory* buildEngineYolov8Det(nvinfer1::IBuilder* builder, nvinfer1::IBuilderConfig* config,
                                            const std::string& wts, const Yolov8Config& cfg);

}  // namespace trtx_host
