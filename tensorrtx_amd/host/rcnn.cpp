// Faster R-CNN R50-C4 (BASELINE config 5).  Mirrors the reference builder (rcnn/rcnn.cpp:79-278 with the backbone of
// rcnn/backbone.hpp:26-229): HWC image -> CHW shuffle -> (x - mean) / std with ConstantLayers -> BN-fused ResNet stem +
// res2..res4 (detectron2 weights after fuse-bn, so every conv carries a bias; stride lives in the first 1x1,
// STRIDE_IN_1X1) -> RPN head (3x3 + objectness/anchor-delta 1x1) -> RpnDecode -> RpnNms -> RoiAlign(14) -> res5 run on
// the (proposals, C, 14, 14) tensor -> HW mean -> two FullyConnected -> softmax / slice -> PredictorDecode -> BatchedNms.
// The Mask head (MASK_ON, rcnn.cpp:202-232; off by default in the reference) is built when RcnnConfig::mask_on is set.
// The reference's file-scope constants (rcnn.cpp:16-60) are the fields of RcnnConfig.
#include "common.h"
#include "models.h"
#include "plugins/rcnn_plugins.h"

using namespace nvinfer1;

namespace nvinfer1 {
REGISTER_TENSORRT_PLUGIN(RpnDecodePluginCreator);
REGISTER_TENSORRT_PLUGIN(RpnNmsPluginCreator);
REGISTER_TENSORRT_PLUGIN(RoiAlignPluginCreator);
REGISTER_TENSORRT_PLUGIN(PredictorDecodePluginCreator);
REGISTER_TENSORRT_PLUGIN(BatchedNmsPluginCreator);
REGISTER_TENSORRT_PLUGIN(MaskRcnnInferencePluginCreator);
}  // namespace nvinfer1

namespace trtx_host {
namespace {

struct Ctx {
    INetworkDefinition* net;
    WeightMap& wm;
};

// conv with bias (+ReLU): every convolution of the fused-BN detectron2 export has the same shape of code
ITensor* conv(Ctx& c, ITensor& in, int out, int k, int stride, int pad, const std::string& name, bool withRelu) {
    auto* l = c.net->addConvolutionNd(in, out, DimsHW{k, k}, need(c.wm, name + ".weight"), need(c.wm, name + ".bias"));
    assert(l);
    l->setStrideNd(DimsHW{stride, stride});
    l->setPaddingNd(DimsHW{pad, pad});
    ITensor* t = l->getOutput(0);
    return withRelu ? c.net->addActivation(*t, ActivationType::kRELU)->getOutput(0) : t;
}

// backbone.hpp:104-169; the channel axis is the third from the end so the same code serves (C,H,W) and (P,C,H,W)
ITensor* bottleneckBlock(Ctx& c, ITensor& in, int inCh, int midCh, int outCh, int stride, const std::string& l) {
    ITensor* a = conv(c, in, midCh, 1, stride, 0, l + ".conv1", true);  // STRIDE_IN_1X1
    ITensor* b = conv(c, *a, midCh, 3, 1, 1, l + ".conv2", true);
    ITensor* d = conv(c, *b, outCh, 1, 1, 0, l + ".conv3", false);
    ITensor* sc = inCh != outCh ? conv(c, in, outCh, 1, stride, 0, l + ".shortcut", false) : &in;
    auto* sum = c.net->addElementWise(*d, *sc, ElementWiseOperation::kSUM);
    assert(sum);
    return c.net->addActivation(*sum->getOutput(0), ActivationType::kRELU)->getOutput(0);
}

ITensor* makeStage(Ctx& c, ITensor& in, int blocks, int inCh, int midCh, int outCh, int firstStride, const std::string& l) {
    ITensor* t = &in;
    for (int i = 0; i < blocks; ++i) {
        t = bottleneckBlock(c, *t, inCh, midCh, outCh, i == 0 ? firstStride : 1, l + "." + std::to_string(i));
        inCh = outCh;
    }
    return t;
}

// rcnn/common.hpp GenerateAnchors: (-w/2, -h/2, w/2, h/2) per (size, ratio), w = sqrt(size^2 / ratio), h = ratio * w
std::vector<float> generateAnchors(const std::vector<float>& sizes, const std::vector<float>& ratios) {
    std::vector<float> a;
    for (float s : sizes)
        for (float r : ratios) {
            const float w = std::sqrt(s * s / r), h = r * w;
            a.insert(a.end(), {-w / 2.0f, -h / 2.0f, w / 2.0f, h / 2.0f});
        }
    return a;
}

}  // namespace

IHostMemory* buildRcnnR50C4(IBuilder* builder, IBuilderConfig* config, const std::string& wts, const RcnnConfig& cfg) {
    INetworkDefinition* network = builder->createNetworkV2(0U);
    WeightMap wm = loadWeights(wts);
    Ctx c{network, wm};

    // preprocess (rcnn.cpp:79-99)
    ITensor* data = network->addInput("images", DataType::kFLOAT, Dims3{cfg.input_h, cfg.input_w, 3});
    assert(data);
    auto* chw = network->addShuffle(*data);
    chw->setFirstTranspose(Permutation{2, 0, 1});
    auto* mean = network->addConstant(Dims3{3, 1, 1}, Weights{DataType::kFLOAT, cfg.pixel_mean, 3});
    auto* sub = network->addElementWise(*chw->getOutput(0), *mean->getOutput(0), ElementWiseOperation::kSUB);
    auto* std_ = network->addConstant(Dims3{3, 1, 1}, Weights{DataType::kFLOAT, cfg.pixel_std, 3});
    auto* div = network->addElementWise(*sub->getOutput(0), *std_->getOutput(0), ElementWiseOperation::kDIV);

    // backbone: stem + res2..res4 (backbone.hpp:26-49, 199-229)
    ITensor* t = conv(c, *div->getOutput(0), 64, 7, 2, 3, "backbone.stem.conv1", true);
    auto* pool = network->addPoolingNd(*t, PoolingType::kMAX, DimsHW{3, 3});
    pool->setStrideNd(DimsHW{2, 2});
    pool->setPaddingNd(DimsHW{1, 1});
    t = pool->getOutput(0);
    const int blocks[3] = {3, 4, 6};
    int inCh = 64, midCh = 64, outCh = cfg.res2_out_channels;
    for (int s = 0; s < 3; ++s) {
        t = makeStage(c, *t, blocks[s], inCh, midCh, outCh, s == 0 ? 1 : 2, "backbone.res" + std::to_string(s + 2));
        inCh = outCh;
        midCh *= 2;
        outCh *= 2;
    }
    ITensor* features = t;  // {1024, H/16, W/16}
    const int featCh = features->getDimensions().d[0];

    // RPN (rcnn.cpp:101-146)
    const int numAnchors = (int)(cfg.anchor_sizes.size() * cfg.aspect_ratios.size());
    ITensor* rpnHidden = conv(c, *features, featCh, 3, 1, 1, "proposal_generator.rpn_head.conv", true);
    ITensor* logits = conv(c, *rpnHidden, numAnchors, 1, 1, 0, "proposal_generator.rpn_head.objectness_logits", false);
    ITensor* deltas = conv(c, *rpnHidden, numAnchors * 4, 1, 1, 0, "proposal_generator.rpn_head.anchor_deltas", false);
    RpnDecodePlugin rpnDecode(cfg.pre_nms_topk, generateAnchors(cfg.anchor_sizes, cfg.aspect_ratios), (float)cfg.stride,
                              cfg.input_h, cfg.input_w);
    ITensor* decodeIn[] = {logits, deltas};
    auto* decoded = network->addPluginV2(decodeIn, 2, rpnDecode);
    assert(decoded);
    RpnNmsPlugin rpnNms(cfg.rpn_nms_thresh, cfg.post_nms_topk);
    ITensor* nmsIn[] = {decoded->getOutput(0), decoded->getOutput(1)};
    auto* nms = network->addPluginV2(nmsIn, 2, rpnNms);
    assert(nms);
    ITensor* proposals = nms->getOutput(0);  // {post_nms_topk, 4}

    // box head (rcnn.cpp:148-200)
    RoiAlignPlugin roiAlign(cfg.pooler_resolution, 1.0f / (float)cfg.stride, cfg.sampling_ratio, cfg.post_nms_topk, featCh);
    ITensor* roiIn[] = {proposals, features};
    auto* rois = network->addPluginV2(roiIn, 2, roiAlign);
    assert(rois);
    ITensor* boxFeatures = makeStage(c, *rois->getOutput(0), 3, featCh, 512, cfg.res2_out_channels * 8, 2, "roi_heads.res5");
    auto* pooled = network->addReduce(*boxFeatures, ReduceOperation::kAVG, 12, true);  // axes H,W of {P,C,H,W}
    assert(pooled);
    auto* scores = network->addFullyConnected(*pooled->getOutput(0), cfg.num_classes + 1,
                                              need(wm, "roi_heads.box_predictor.cls_score.weight"),
                                              need(wm, "roi_heads.box_predictor.cls_score.bias"));
    auto* probs = network->addSoftMax(*scores->getOutput(0));
    const Dims pd = probs->getOutput(0)->getDimensions();
    auto* fg = network->addSlice(*probs->getOutput(0), Dims4{0, 0, 0, 0}, Dims4{pd.d[0], pd.d[1] - 1, 1, 1}, Dims4{1, 1, 1, 1});
    auto* boxDeltas = network->addFullyConnected(*pooled->getOutput(0), cfg.num_classes * 4,
                                                 need(wm, "roi_heads.box_predictor.bbox_pred.weight"),
                                                 need(wm, "roi_heads.box_predictor.bbox_pred.bias"));
    PredictorDecodePlugin predictorDecode((int)pd.d[0], cfg.input_h, cfg.input_w,
                                          std::vector<float>(cfg.bbox_reg_weights, cfg.bbox_reg_weights + 4));
    ITensor* predIn[] = {fg->getOutput(0), boxDeltas->getOutput(0), proposals};
    auto* pred = network->addPluginV2(predIn, 3, predictorDecode);
    assert(pred);
    BatchedNmsPlugin batchedNms(cfg.nms_method, cfg.nms_thresh_test, cfg.detections_per_image);
    ITensor* finalIn[] = {pred->getOutput(0), pred->getOutput(1), pred->getOutput(2)};
    auto* dets = network->addPluginV2(finalIn, 3, batchedNms);
    assert(dets);

    const char* names[3] = {"scores", "boxes", "labels"};  // OUTPUT_NAMES, rcnn.cpp:51-52
    for (int i = 0; i < 3; ++i) {
        dets->getOutput(i)->setName(names[i]);
        network->markOutput(*dets->getOutput(i));
    }
    if (cfg.mask_on) {  // MaskHead, rcnn.cpp:202-232: the final boxes go through RoIAlign + res5 (shared weights) once more
        RoiAlignPlugin maskRoiAlign(cfg.pooler_resolution, 1.0f / (float)cfg.stride, cfg.sampling_ratio, cfg.detections_per_image,
                                    featCh);
        ITensor* maskRoiIn[] = {dets->getOutput(1), features};
        auto* maskRois = network->addPluginV2(maskRoiIn, 2, maskRoiAlign);
        assert(maskRois);
        ITensor* maskFeatures =
                makeStage(c, *maskRois->getOutput(0), 3, featCh, 512, cfg.res2_out_channels * 8, 2, "roi_heads.res5");
        auto* deconv = network->addDeconvolutionNd(*maskFeatures, 256, DimsHW{2, 2}, need(wm, "roi_heads.mask_head.deconv.weight"),
                                                   need(wm, "roi_heads.mask_head.deconv.bias"));
        assert(deconv);
        deconv->setStrideNd(DimsHW{2, 2});
        auto* deconvRelu = network->addActivation(*deconv->getOutput(0), ActivationType::kRELU);
        ITensor* maskLogits = conv(c, *deconvRelu->getOutput(0), cfg.num_classes, 1, 1, 0, "roi_heads.mask_head.predictor", false);
        ITensor* masks;
        if (cfg.num_classes == 1) {
            masks = network->addActivation(*maskLogits, ActivationType::kSIGMOID)->getOutput(0);
        } else {
            MaskRcnnInferencePlugin maskSelect(cfg.detections_per_image, cfg.pooler_resolution);
            ITensor* selIn[] = {dets->getOutput(2), maskLogits};
            auto* sel = network->addPluginV2(selIn, 2, maskSelect);
            assert(sel);
            masks = sel->getOutput(0);
        }
        masks->setName("masks");
        network->markOutput(*masks);
    }
    if (cfg.mark_stages) {  // debugging taps for the parity tests
        features->setName("features");
        network->markOutput(*features);
        proposals->setName("proposals");
        network->markOutput(*proposals);
    }

    builder->setMaxBatchSize(cfg.max_batch);
    config->setMaxWorkspaceSize(1ULL << 30);
    if (cfg.fp16) config->setFlag(BuilderFlag::kFP16);
    IHostMemory* plan = builder->buildSerializedNetwork(*network, *config);
    delete network;
    freeWeights(wm);
    return plan;
}

}  // namespace trtx_host
