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

    // Layer creation order and layer names are the reference's, statement for statement (lenet/lenet.cpp:54-124): its own builder,
    // compiled against the shim, serializes the same bytes (tests/test_ref_builders.py).
    auto* conv1 = net->addConvolutionNd(*x, 6, DimsHW{5, 5}, need(wm, "conv1.weight"), need(wm, "conv1.bias"));
    assert(conv1);
    conv1->setStrideNd(DimsHW{1, 1});
    conv1->setName("conv1");
    auto* relu1 = net->addActivation(*conv1->getOutput(0), ActivationType::kRELU);
    relu1->setName("relu1");
    auto* pool1 = net->addPoolingNd(*relu1->getOutput(0), PoolingType::kMAX, DimsHW{2, 2});
    pool1->setStrideNd(DimsHW{2, 2});
    pool1->setName("pool1");
    auto* conv2 = net->addConvolutionNd(*pool1->getOutput(0), 16, DimsHW{5, 5}, need(wm, "conv2.weight"), need(wm, "conv2.bias"));
    assert(conv2);
    conv2->setStrideNd(DimsHW{1, 1});
    conv2->setName("conv2");
    auto* relu2 = net->addActivation(*conv2->getOutput(0), ActivationType::kRELU);  // the one activation the reference leaves unnamed (:78)
    auto* pool2 = net->addPoolingNd(*relu2->getOutput(0), PoolingType::kMAX, DimsHW{2, 2});
    pool2->setStrideNd(DimsHW{2, 2});
    pool2->setName("pool2");

    using M = MatrixOperation;
    using E = ElementWiseOperation;
    // fc1: constants are created right where they are used (:90-96)
    auto* flatten = net->addShuffle(*pool2->getOutput(0));
    flatten->setReshapeDimensions(Dims2{-1, 400});
    ITensor* fc1w = net->addConstant(Dims2{120, 400}, need(wm, "fc1.weight"))->getOutput(0);
    auto* fc1 = net->addMatrixMultiply(*fc1w, M::kNONE, *flatten->getOutput(0), M::kTRANSPOSE);
    ITensor* fc1bias = net->addConstant(Dims2{120, 1}, need(wm, "fc1.bias"))->getOutput(0);
    auto* fc1b = net->addElementWise(*fc1->getOutput(0), *fc1bias, E::kSUM);
    fc1b->setName("fc1b");
    auto* relu3 = net->addActivation(*fc1b->getOutput(0), ActivationType::kRELU);
    auto* flatten3 = net->addShuffle(*relu3->getOutput(0));
    flatten3->setReshapeDimensions(Dims2{-1, 120});
    // fc2 / fc3: all four constants first (:104-107), then the products
    ITensor* fc2w = net->addConstant(Dims2{84, 120}, need(wm, "fc2.weight"))->getOutput(0);
    ITensor* fc2b = net->addConstant(Dims2{84, 1}, need(wm, "fc2.bias"))->getOutput(0);
    ITensor* fc3w = net->addConstant(Dims2{10, 84}, need(wm, "fc3.weight"))->getOutput(0);
    ITensor* fc3b = net->addConstant(Dims2{10, 1}, need(wm, "fc3.bias"))->getOutput(0);
    auto* fc2 = net->addMatrixMultiply(*fc2w, M::kNONE, *flatten3->getOutput(0), M::kTRANSPOSE);
    fc2->setName("fc2");
    auto* fc2sum = net->addElementWise(*fc2->getOutput(0), *fc2b, E::kSUM);
    auto* relu4 = net->addActivation(*fc2sum->getOutput(0), ActivationType::kRELU);
    auto* flatten4 = net->addShuffle(*relu4->getOutput(0));
    flatten4->setReshapeDimensions(Dims2{-1, 84});
    auto* fc3 = net->addMatrixMultiply(*fc3w, M::kNONE, *flatten4->getOutput(0), M::kTRANSPOSE);
    auto* fc3sum = net->addElementWise(*fc3->getOutput(0), *fc3b, E::kSUM);
    x = fc3sum->getOutput(0);
    ISoftMaxLayer* prob = net->addSoftMax(*x);
    assert(prob);
    prob->getOutput(0)->setName("prob");
    net->markOutput(*prob->getOutput(0));

    config->setMemoryPoolLimit(MemoryPoolType::kWORKSPACE, 16 << 20);  // WORKSPACE_SIZE, lenet/utils.h:18
    IHostMemory* plan = builder->buildSerializedNetwork(*net, *config);
    delete net;
    freeWeights(wm);
    return plan;
}

}  // namespace trtx_host
