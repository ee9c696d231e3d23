// ResNet-50 classifier (BASELINE config 2).  Mirrors the reference builder (resnet/resnet50.cpp:111-229):
// 7x7/2 stem + BN + ReLU + maxpool3/2, bottleneck stages [3,4,6,3] with the stride on the 3x3 conv,
// BN folded into IScaleLayers with eps 1e-5, average pool 7x7 and FC-1000.  Implicit batch.
#include "common.h"
#include "models.h"

using namespace nvinfer1;

namespace trtx_host {
namespace {

ITensor* convBn(INetworkDefinition* net, WeightMap& wm, ITensor& in, int out, int k, int s, int p, const std::string& conv,
                const std::string& bn) {
    auto* c = net->addConvolutionNd(in, out, DimsHW{k, k}, need(wm, conv + ".weight"), noWeights());
    assert(c);
    c->setStrideNd(DimsHW{s, s});
    c->setPaddingNd(DimsHW{p, p});
    return addBatchNorm2d(net, wm, *c->getOutput(0), bn, 1e-5f, /*sqrt_in_double=*/true)->getOutput(0);
}

// resnet50.cpp:111-151
ITensor* bottleneck(INetworkDefinition* net, WeightMap& wm, ITensor& in, int inch, int outch, int stride, const std::string& l) {
    ITensor* a = convBn(net, wm, in, outch, 1, 1, 0, l + "conv1", l + "bn1");
    a = net->addActivation(*a, ActivationType::kRELU)->getOutput(0);
    ITensor* b = convBn(net, wm, *a, outch, 3, stride, 1, l + "conv2", l + "bn2");
    b = net->addActivation(*b, ActivationType::kRELU)->getOutput(0);
    ITensor* c = convBn(net, wm, *b, outch * 4, 1, 1, 0, l + "conv3", l + "bn3");
    ITensor* shortcut = &in;
    if (stride != 1 || inch != outch * 4) shortcut = convBn(net, wm, in, outch * 4, 1, stride, 0, l + "downsample.0", l + "downsample.1");
    auto* sum = net->addElementWise(*shortcut, *c, ElementWiseOperation::kSUM);
    return net->addActivation(*sum->getOutput(0), ActivationType::kRELU)->getOutput(0);
}

}  // namespace

IHostMemory* buildResnet50(IBuilder* builder, IBuilderConfig* config, const std::string& wts, int maxBatch, bool fp16, int H, int W) {
    WeightMap wm = loadWeights(wts);
    INetworkDefinition* net = builder->createNetworkV2(0U);
    ITensor* x = net->addInput("data", DataType::kFLOAT, Dims3{3, H, W});
    assert(x);
    x = convBn(net, wm, *x, 64, 7, 2, 3, "conv1", "bn1");
    x = net->addActivation(*x, ActivationType::kRELU)->getOutput(0);
    auto* pool = net->addPoolingNd(*x, PoolingType::kMAX, DimsHW{3, 3});
    pool->setStrideNd(DimsHW{2, 2});
    pool->setPaddingNd(DimsHW{1, 1});
    x = pool->getOutput(0);

    const int blocks[4] = {3, 4, 6, 3};
    int inch = 64;
    for (int stage = 0; stage < 4; ++stage) {
        const int width = 64 << stage;
        for (int b = 0; b < blocks[stage]; ++b) {
            const int stride = (b == 0 && stage > 0) ? 2 : 1;
            x = bottleneck(net, wm, *x, inch, width, stride, "layer" + std::to_string(stage + 1) + "." + std::to_string(b) + ".");
            inch = width * 4;
        }
    }
    const Dims d = x->getDimensions();
    auto* avg = net->addPoolingNd(*x, PoolingType::kAVERAGE, DimsHW{(int)d.d[1], (int)d.d[2]});  // 7x7 at 224 input
    avg->setStrideNd(DimsHW{1, 1});
    auto* fc = net->addFullyConnected(*avg->getOutput(0), 1000, need(wm, "fc.weight"), need(wm, "fc.bias"));
    assert(fc);
    fc->getOutput(0)->setName("prob");
    net->markOutput(*fc->getOutput(0));

    builder->setMaxBatchSize(maxBatch);
    config->setMaxWorkspaceSize(1 << 20);
    if (fp16) config->setFlag(BuilderFlag::kFP16);
    IHostMemory* plan = builder->buildSerializedNetwork(*net, *config);
    delete net;
    freeWeights(wm);
    return plan;
}

}  // namespace trtx_host
