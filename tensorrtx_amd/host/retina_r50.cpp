// RetinaFace with a ResNet-50 body (BASELINE config 4).  Mirrors the reference builder
// (retinaface/retina_r50.cpp:27-242): R50 body (BN eps 1e-5 as IScaleLayer) -> FPN whose top-down path
// upsamples with a 2x2 stride-2 grouped deconvolution of ones (:157-176) -> three SSH context modules (:87-98)
// -> bbox(8) / class(4) / landmark(20) 1x1 heads per level -> concat -> "Decode_TRT" plugin.
// Input size is a run-time argument (the reference compiles 480x640 into decode.h:16-17).
#include "common.h"
#include "models.h"

using namespace nvinfer1;

namespace trtx_host {
namespace {

struct Ctx {
    INetworkDefinition* net;
    WeightMap& wm;
};

ITensor* convBn(Ctx& c, ITensor& in, int out, int k, int s, int p, const std::string& conv, const std::string& bn) {
    auto* l = c.net->addConvolutionNd(in, out, DimsHW{k, k}, need(c.wm, conv + ".weight"), noWeights());
    assert(l);
    l->setStrideNd(DimsHW{s, s});
    l->setPaddingNd(DimsHW{p, p});
    return addBatchNorm2d(c.net, c.wm, *l->getOutput(0), bn, 1e-5f, /*sqrt_in_double=*/true)->getOutput(0);
}
ITensor* relu(Ctx& c, ITensor* t) { return c.net->addActivation(*t, ActivationType::kRELU)->getOutput(0); }

ITensor* bottleneck(Ctx& c, ITensor& in, int inch, int outch, int stride, const std::string& l) {  // retina_r50.cpp:27-68
    ITensor* a = relu(c, convBn(c, in, outch, 1, 1, 0, l + "conv1", l + "bn1"));
    ITensor* b = relu(c, convBn(c, *a, outch, 3, stride, 1, l + "conv2", l + "bn2"));
    ITensor* d = convBn(c, *b, outch * 4, 1, 1, 0, l + "conv3", l + "bn3");
    ITensor* sc = &in;
    if (stride != 1 || inch != outch * 4) sc = convBn(c, in, outch * 4, 1, stride, 0, l + "downsample.0", l + "downsample.1");
    return relu(c, c.net->addElementWise(*sc, *d, ElementWiseOperation::kSUM)->getOutput(0));
}

// conv_bn_relu, retina_r50.cpp:70-85 (weights "<lname>.0", BN "<lname>.1")
ITensor* convBnRelu(Ctx& c, ITensor& in, int out, int k, int s, int p, bool useRelu, const std::string& lname) {
    ITensor* t = convBn(c, in, out, k, s, p, lname + ".0", lname + ".1");
    return useRelu ? relu(c, t) : t;
}

ITensor* ssh(Ctx& c, ITensor& in, const std::string& l) {  // retina_r50.cpp:87-98
    ITensor* c3 = convBnRelu(c, in, 128, 3, 1, 1, false, l + ".conv3X3");
    ITensor* c5a = convBnRelu(c, in, 64, 3, 1, 1, true, l + ".conv5X5_1");
    ITensor* c5 = convBnRelu(c, *c5a, 64, 3, 1, 1, false, l + ".conv5X5_2");
    ITensor* c7 = convBnRelu(c, *c5a, 64, 3, 1, 1, true, l + ".conv7X7_2");
    c7 = convBnRelu(c, *c7, 64, 3, 1, 1, false, l + ".conv7x7_3");
    ITensor* parts[] = {c3, c5, c7};
    return relu(c, c.net->addConcatenation(parts, 3)->getOutput(0));
}

ITensor* upsampleOnes(Ctx& c, ITensor& in, const Weights& ones) {  // :157-166 nearest x2 spelled as a grouped deconvolution
    auto* d = c.net->addDeconvolutionNd(in, 256, DimsHW{2, 2}, ones, noWeights());
    assert(d);
    d->setStrideNd(DimsHW{2, 2});
    d->setNbGroups(256);
    return d->getOutput(0);
}

}  // namespace

IHostMemory* buildRetinaFaceR50(IBuilder* builder, IBuilderConfig* config, const std::string& wts, int maxBatch, bool fp16, int H, int W) {
    WeightMap wm = loadWeights(wts);
    INetworkDefinition* net = builder->createNetworkV2(0U);
    Ctx c{net, wm};
    ITensor* x = net->addInput("data", DataType::kFLOAT, Dims3{3, H, W});
    assert(x);
    x = relu(c, convBn(c, *x, 64, 7, 2, 3, "body.conv1", "body.bn1"));
    auto* pool = net->addPoolingNd(*x, PoolingType::kMAX, DimsHW{3, 3});
    pool->setStrideNd(DimsHW{2, 2});
    pool->setPaddingNd(DimsHW{1, 1});
    x = pool->getOutput(0);
    const int blocks[4] = {3, 4, 6, 3};
    ITensor* stage[4];
    int inch = 64;
    for (int s = 0; s < 4; ++s) {
        const int width = 64 << s;
        for (int b = 0; b < blocks[s]; ++b) {
            x = bottleneck(c, *x, inch, width, (b == 0 && s > 0) ? 2 : 1, "body.layer" + std::to_string(s + 1) + "." + std::to_string(b) + ".");
            inch = width * 4;
        }
        stage[s] = x;
    }
    // FPN (:152-176)
    ITensor* o1 = convBnRelu(c, *stage[1], 256, 1, 1, 0, true, "fpn.output1");
    ITensor* o2 = convBnRelu(c, *stage[2], 256, 1, 1, 0, true, "fpn.output2");
    ITensor* o3 = convBnRelu(c, *stage[3], 256, 1, 1, 0, true, "fpn.output3");
    float* ones = static_cast<float*>(std::malloc(sizeof(float) * 256 * 2 * 2));
    for (int i = 0; i < 256 * 2 * 2; ++i) ones[i] = 1.0f;
    wm["up3"] = Weights{DataType::kFLOAT, ones, 256 * 2 * 2};
    o2 = net->addElementWise(*o2, *upsampleOnes(c, *o3, wm["up3"]), ElementWiseOperation::kSUM)->getOutput(0);
    o2 = convBnRelu(c, *o2, 256, 3, 1, 1, true, "fpn.merge2");
    o1 = net->addElementWise(*o1, *upsampleOnes(c, *o2, wm["up3"]), ElementWiseOperation::kSUM)->getOutput(0);
    o1 = convBnRelu(c, *o1, 256, 3, 1, 1, true, "fpn.merge1");
    // SSH + heads (:178-202)
    ITensor* feats[3] = {ssh(c, *o1, "ssh1"), ssh(c, *o2, "ssh2"), ssh(c, *o3, "ssh3")};
    // layer creation order = the reference's (:183-202): the three bbox heads, the three class heads, the three landmark heads, then
    // one concatenation per level -- the plan of the reference's own createEngine is byte-identical (tests/test_ref_builders.py)
    const char* kinds[3] = {"BboxHead", "ClassHead", "LandmarkHead"};
    const int kind_ch[3] = {2 * 4, 2 * 2, 2 * 10};
    ITensor* heads[3][3];
    for (int k = 0; k < 3; ++k)
        for (int l = 0; l < 3; ++l) {
            const std::string base = std::string(kinds[k]) + "." + std::to_string(l) + ".conv1x1.";
            heads[k][l] = net->addConvolutionNd(*feats[l], kind_ch[k], DimsHW{1, 1}, need(wm, base + "weight"), need(wm, base + "bias"))->getOutput(0);
        }
    std::vector<ITensor*> cats;
    for (int l = 0; l < 3; ++l) {
        ITensor* parts[] = {heads[0][l], heads[1][l], heads[2][l]};
        cats.push_back(net->addConcatenation(parts, 3)->getOutput(0));
    }
    auto* creator = getPluginRegistry()->getPluginCreator("Decode_TRT", "1");
    assert(creator && "Decode_TRT creator not registered");
    PluginFieldCollection pfc{};
    IPluginV2* plugin = creator->createPlugin("decode", &pfc);
    auto* dec = net->addPluginV2(cats.data(), 3, *plugin);
    assert(dec);
    plugin->destroy();
    dec->getOutput(0)->setName("prob");
    net->markOutput(*dec->getOutput(0));

    builder->setMaxBatchSize(maxBatch);
    config->setMaxWorkspaceSize(1 << 20);
    if (fp16) config->setFlag(BuilderFlag::kFP16);
    IHostMemory* plan = builder->buildSerializedNetwork(*net, *config);
    delete net;
    freeWeights(wm);
    return plan;
}

}  // namespace trtx_host
