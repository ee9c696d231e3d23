// LeNet-5 through the network-definition API — the "hello world" config (BASELINE config 1).
// Mirrors createLenetEngine of the reference (lenet/lenet.cpp:36-155): explicit batch, conv5x5 -> relu ->
// maxpool2 twice, then three fully connected stages spelled as Constant + MatrixMultiply(kTRANSPOSE) +
// ElementWise(SUM), and a softmax.  PyTorch twin: lenet/gen_wts.py:10-45.
#include <cmath>

#include "common.h"
#include "models.h"

using namespace nvinfer1;

namespace trtx_host {

IHostMemory* buildLenet(IBuilder* builder, IBuilderConfig* config, const std::string& wts, int32_t N) {
    WeightMap wm = loadWeights(wts);
    const uint32_t flag = 1U << static_cast<int>(NetworkDefinitionCreationFlag::kEXPLICIT_BATCH);
    INetworkDefinition* net = builder->createNetworkV2(flag);
    ITensor* x = net->addInput("data", DataType::kFLOAT, Dims4{N, 1, 32, 32});
    assert(x);

    struct ConvSpec { const char* name; int out; };
    for (const ConvSpec& c : {ConvSpec{"conv1", 6}, ConvSpec{"conv2", 16}}) {
        auto* conv = net->addConvolutionNd(*x, c.out, DimsHW{5, 5}, need(wm, std::string(c.name) + ".weight"),
                                           need(wm, std::string(c.name) + ".bias"));
        assert(conv);
        conv->setStrideNd(DimsHW{1, 1});
        conv->setName(c.name);
        auto* relu = net->addActivation(*conv->getOutput(0), ActivationType::kRELU);
        auto* pool = net->addPoolingNd(*relu->getOutput(0), PoolingType::kMAX, DimsHW{2, 2});
        pool->setStrideNd(DimsHW{2, 2});
        x = pool->getOutput(0);
    }

    struct FcSpec { const char* name; int out, in; bool relu; };
    for (const FcSpec& f : {FcSpec{"fc1", 120, 400, true}, FcSpec{"fc2", 84, 120, true}, FcSpec{"fc3", 10, 84, false}}) {
        auto* flat = net->addShuffle(*x);
        flat->setReshapeDimensions(Dims2{-1, f.in});
        ITensor* w = net->addConstant(Dims2{f.out, f.in}, need(wm, std::string(f.name) + ".weight"))->getOutput(0);
        ITensor* b = net->addConstant(Dims2{f.out, 1}, need(wm, std::string(f.name) + ".bias"))->getOutput(0);
        auto* mm = net->addMatrixMultiply(*w, MatrixOperation::kNONE, *flat->getOutput(0), MatrixOperation::kTRANSPOSE);
        assert(mm);
        mm->setName(f.name);
        auto* sum = net->addElementWise(*mm->getOutput(0), *b, ElementWiseOperation::kSUM);
        x = sum->getOutput(0);
        if (f.relu) x = net->addActivation(*x, ActivationType::kRELU)->getOutput(0);
    }
    ISoftMaxLayer* prob = net->addSoftMax(*x);
    assert(prob);
    prob->getOutput(0)->setName("prob");
    net->markOutput(*prob->getOutput(0));

    config->setMemoryPoolLimit(MemoryPoolType::kWORKSPACE, 1 << 20);
    IHostMemory* plan = builder->buildSerializedNetwork(*net, *config);
    delete net;
    freeWeights(wm);
    return plan;
}

}  // namespace trtx_host
