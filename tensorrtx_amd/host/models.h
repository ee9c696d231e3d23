// Model builders written against the nvinfer1-shaped API (include/NvInfer.h), mirroring the reference's
// per-model builder functions.  Compile-time constants of the reference (kInputH/W, kBatchSize, kNumClass,
// USE_FP16 ... yolov8/include/config.h:1-31) are run-time arguments here so one binary serves batch 32,
// 1280x1280, etc. (SURVEY.md §5 "Config / flags").
#pragma once
#include <string>
#include <vector>

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
    // 0 det (buildEngineYolov8Det), 1 seg (..Seg, model.cpp:1057-1308: + 32 mask coefficients per cell and the "proto" output),
    // 2 pose (..Pose, :1310-1563: + kNumberOfPoints * 3 keypoint values), 3 obb (..Obb, :2499-2740: + 1 angle logit)
    int task = 0;
    int num_points = 17;                // kNumberOfPoints (include/config.h:28)
    float kpt_conf = 0.5f;              // kConfThreshKeypoints (include/config.h:14); the plugin field carries (int)kpt_conf like the reference
};
// yolov8/src/model.cpp:98-336 (det), 1057-1308 (seg), 1310-1563 (pose), 2499-2740 (obb): one graph, the task adds the cv4 branch
nvinfer1::IHostMemory* buildEngineYolov8Det(nvinfer1::IBuilder* builder, nvinfer1::IBuilderConfig* config,
                                            const std::string& wts, const Yolov8Config& cfg);

// The reference's file-scope constants (rcnn/rcnn.cpp:16-60) as run-time configuration.
struct RcnnConfig {
    int input_h = 800, input_w = 1067;      // INPUT_H / INPUT_W: 480x640 resized by calculateSize() (rcnn.cpp:349-366)
    int max_batch = 1;                      // BATCH_SIZE
    bool fp16 = true;
    float pixel_mean[3] = {103.53f, 116.28f, 123.675f};
    float pixel_std[3] = {1.0f, 1.0f, 1.0f};
    int num_classes = 80;
    int res2_out_channels = 256;            // R50
    std::vector<float> anchor_sizes = {32, 64, 128, 256, 512};
    std::vector<float> aspect_ratios = {0.5f, 1.0f, 2.0f};
    int pre_nms_topk = 6000;                // PRE_NMS_TOP_K_TEST
    float rpn_nms_thresh = 0.7f;
    int post_nms_topk = 1000;
    int stride = 16;                        // STRIDES
    int sampling_ratio = 0;
    int pooler_resolution = 14;
    float nms_thresh_test = 0.5f;
    int detections_per_image = 100;
    float bbox_reg_weights[4] = {10.0f, 10.0f, 5.0f, 5.0f};
    int nms_method = 1;                     // 0 hard, 1 soft-NMS linear (reference default), 2 soft-NMS gaussian
    bool mask_on = false;                   // MASK_ON: Mask R-CNN head (rcnn.cpp:202-232), extra output "masks"
    bool mark_stages = false;               // debugging: also expose "features" and "proposals"
};
// rcnn/rcnn.cpp:79-278 (box head, and the mask head when mask_on)
nvinfer1::IHostMemory* buildRcnnR50C4(nvinfer1::IBuilder* builder, nvinfer1::IBuilderConfig* config, const std::string& wts,
                                      const RcnnConfig& cfg);

}  // namespace trtx_host
