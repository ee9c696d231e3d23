// YOLOv8 detection network (BASELINE config 3) through the network-definition API.
// Mirrors the reference blocks and builder:
//   convBnSiLU / bottleneck / C2F / SPPF / DFL / addYoLoLayer     yolov8/src/block.cpp:79-309
//   get_width / get_depth / buildEngineYolov8Det                  yolov8/src/model.cpp:13-25, 98-336
// Graph, weight keys ("model.<n>...") and layer order are those of the reference; the code is organised
// around small tables (backbone stages, the three detect levels) instead of one long listing.
#include <cmath>
#include <vector>

#include "common.h"
#include "models.h"

using namespace nvinfer1;

namespace trtx_host {
namespace {

int get_width(int x, float gw, int max_channels, int divisor = 8) {  // model.cpp:13-16
    const int ch = int(ceil((x * gw) / divisor)) * divisor;
    return ch >= max_channels ? max_channels : ch;
}

int get_depth(int x, float gd) {  // model.cpp:18-25 (round-half-to-even like Python's round)
    if (x == 1) return 1;
    int r = (int)round(x * gd);
    if (x * gd - int(x * gd) == 0.5 && (int(x * gd) % 2) == 0) --r;
    return std::max<int>(r, 1);
}

struct Ctx {
    INetworkDefinition* net;
    WeightMap& wm;
};

// Conv(no bias) + BN(eps 1e-3) + SiLU spelled Sigmoid * x  (block.cpp:79-96)
ITensor* convBnSiLU(Ctx& c, ITensor& in, int ch, int k, int s, int p, const std::string& lname) {
    auto* conv = c.net->addConvolutionNd(in, ch, DimsHW{k, k}, need(c.wm, lname + ".conv.weight"), noWeights());
    assert(conv);
    conv->setStrideNd(DimsHW{s, s});
    conv->setPaddingNd(DimsHW{p, p});
    ITensor* bn = addBatchNorm2d(c.net, c.wm, *conv->getOutput(0), lname + ".bn", 1e-3f)->getOutput(0);
    ITensor* sig = c.net->addActivation(*bn, ActivationType::kSIGMOID)->getOutput(0);
    auto* prod = c.net->addElementWise(*bn, *sig, ElementWiseOperation::kPROD);
    assert(prod);
    return prod->getOutput(0);
}

ITensor* bottleneck(Ctx& c, ITensor& in, int c1, int c2, bool shortcut, const std::string& lname) {  // block.cpp:98-110
    ITensor* a = convBnSiLU(c, in, c2, 3, 1, 1, lname + ".cv1");
    ITensor* b = convBnSiLU(c, *a, c2, 3, 1, 1, lname + ".cv2");
    if (shortcut && c1 == c2) return c.net->addElementWise(in, *b, ElementWiseOperation::kSUM)->getOutput(0);
    return b;
}

ITensor* C2F(Ctx& c, ITensor& in, int c2, int n, bool shortcut, float e, const std::string& lname) {  // block.cpp:126-155
    const int c_ = (int)((float)c2 * e);
    ITensor* cv1 = convBnSiLU(c, in, 2 * c_, 1, 1, 0, lname + ".cv1");
    const Dims d = cv1->getDimensions();
    const Dims3 half{d.d[0] / 2, d.d[1], d.d[2]}, unit{1, 1, 1};
    ITensor* s1 = c.net->addSlice(*cv1, Dims3{0, 0, 0}, half, unit)->getOutput(0);
    ITensor* s2 = c.net->addSlice(*cv1, Dims3{d.d[0] / 2, 0, 0}, half, unit)->getOutput(0);
    ITensor* pair[] = {s1, s2};
    ITensor* cat = c.net->addConcatenation(pair, 2)->getOutput(0);
    ITensor* y = s2;
    for (int i = 0; i < n; ++i) {
        y = bottleneck(c, *y, c_, c_, shortcut, lname + ".m." + std::to_string(i));
        ITensor* grow[] = {cat, y};
        cat = c.net->addConcatenation(grow, 2)->getOutput(0);
    }
    return convBnSiLU(c, *cat, c2, 1, 1, 0, lname + ".cv2");
}

ITensor* SPPF(Ctx& c, ITensor& in, int c1, int c2, int k, const std::string& lname) {  // block.cpp:214-237
    ITensor* x = convBnSiLU(c, in, c1 / 2, 1, 1, 0, lname + ".cv1");
    std::vector<ITensor*> parts{x};
    for (int i = 0; i < 3; ++i) {
        auto* pool = c.net->addPoolingNd(*parts.back(), PoolingType::kMAX, DimsHW{k, k});
        pool->setStrideNd(DimsHW{1, 1});
        pool->setPaddingNd(DimsHW{k / 2, k / 2});
        parts.push_back(pool->getOutput(0));
    }
    ITensor* cat = c.net->addConcatenation(parts.data(), 4)->getOutput(0);
    return convBnSiLU(c, *cat, c2, 1, 1, 0, lname + ".cv2");
}

ITensor* upsample2x(Ctx& c, ITensor& in) {  // model.cpp:143-148
    const float scale[] = {1.0f, 2.0f, 2.0f};
    auto* r = c.net->addResize(in);
    assert(r);
    r->setResizeMode(ResizeMode::kNEAREST);
    r->setScales(scale, 3);
    return r->getOutput(0);
}

ITensor* cat2(Ctx& c, ITensor* a, ITensor* b) {
    ITensor* v[] = {a, b};
    return c.net->addConcatenation(v, 2)->getOutput(0);
}

// Distribution-focal-loss decode: (64, g) -> (4,16,g) -> transpose (16,4,g) -> softmax over 16 bins ->
// 1x1 conv with weights arange(16) -> (4, g)   (block.cpp:239-257)
ITensor* DFL(Ctx& c, ITensor& in, int grid, const std::string& wkey) {
    auto* sh1 = c.net->addShuffle(in);
    sh1->setReshapeDimensions(Dims3{4, 16, grid});
    sh1->setSecondTranspose(Permutation{1, 0, 2});
    auto* sm = c.net->addSoftMax(*sh1->getOutput(0));
    auto* conv = c.net->addConvolutionNd(*sm->getOutput(0), 1, DimsHW{1, 1}, need(c.wm, wkey), noWeights());
    conv->setStrideNd(DimsHW{1, 1});
    conv->setPaddingNd(DimsHW{0, 0});
    auto* sh2 = c.net->addShuffle(*conv->getOutput(0));
    sh2->setReshapeDimensions(Dims2{4, grid});
    return sh2->getOutput(0);
}

// block.cpp:259-309 — the plugin is obtained from the registry under the reference's name
IPluginV2Layer* addYoLoLayer(Ctx& c, const std::vector<ITensor*>& dets, const std::vector<int>& strides, const Yolov8Config& cfg) {
    auto* creator = getPluginRegistry()->getPluginCreator("YoloLayer_TRT", "1");
    assert(creator && "YoloLayer_TRT creator not registered");
    std::vector<int> info = {cfg.num_class, cfg.num_points, (int)cfg.kpt_conf /* block.cpp:278: the float threshold is truncated */,
                             cfg.input_w, cfg.input_h, cfg.max_out_bbox, cfg.task == 1, cfg.task == 2, cfg.task == 3};
    info.insert(info.end(), strides.begin(), strides.end());
    PluginField field("combinedInfo", info.data(), PluginFieldType::kINT32, (int32_t)info.size());
    PluginFieldCollection fc{1, &field};
    IPluginV2* plugin = creator->createPlugin("yololayer", &fc);
    assert(plugin);
    std::vector<ITensor*> ins(dets);
    auto* layer = c.net->addPluginV2(ins.data(), (int32_t)ins.size(), *plugin);
    plugin->destroy();  // the network holds its own clone
    return layer;
}

// cv4_conv_combined (model.cpp:54-96): two 3x3 convBnSiLU + biased 1x1 conv, flattened to (out_ch, grid)
ITensor* cv4Branch(Ctx& c, ITensor& in, const std::string& lname, int grid, const Yolov8Config& cfg) {
    int mid = 0, out_ch = 0;
    if (cfg.task == 1) {  // seg: the reference's width table (:61-70)
        mid = cfg.gw <= 0.5f ? 32 : (cfg.gw == 0.75f ? 48 : (cfg.gw == 1.0f ? 64 : 80));
        out_ch = 32;
    } else {              // pose / obb: the width stored in the weights (:74-82)
        mid = (int)need(c.wm, lname + ".0.bn.weight").count;
        out_ch = cfg.task == 2 ? cfg.num_points * 3 : 1;
    }
    ITensor* a = convBnSiLU(c, in, mid, 3, 1, 1, lname + ".0");
    ITensor* b = convBnSiLU(c, *a, mid, 3, 1, 1, lname + ".1");
    auto* cv = c.net->addConvolutionNd(*b, out_ch, DimsHW{1, 1}, need(c.wm, lname + ".2.weight"), need(c.wm, lname + ".2.bias"));
    assert(cv);
    cv->setStrideNd(DimsHW{1, 1});
    auto* sh = c.net->addShuffle(*cv->getOutput(0));
    sh->setReshapeDimensions(Dims2{out_ch, grid});
    return sh->getOutput(0);
}

// Proto (model.cpp:36-52): 3x3 convBnSiLU -> ConvTranspose 2x2 stride 2 (bias) -> 3x3 -> 1x1 to 32 mask prototypes
ITensor* proto(Ctx& c, ITensor& in, const Yolov8Config& cfg) {
    const int mid = get_width(256, cfg.gw, cfg.max_channels);
    ITensor* a = convBnSiLU(c, in, mid, 3, 1, 1, "model.22.proto.cv1");
    auto* up = c.net->addDeconvolutionNd(*a, mid, DimsHW{2, 2}, need(c.wm, "model.22.proto.upsample.weight"),
                                         need(c.wm, "model.22.proto.upsample.bias"));
    assert(up);
    up->setStrideNd(DimsHW{2, 2});
    ITensor* b = convBnSiLU(c, *up->getOutput(0), mid, 3, 1, 1, "model.22.proto.cv2");
    return convBnSiLU(c, *b, 32, 1, 1, 0, "model.22.proto.cv3");
}

}  // namespace

IHostMemory* buildEngineYolov8Det(IBuilder* builder, IBuilderConfig* config, const std::string& wts, const Yolov8Config& cfg) {
    WeightMap wm = loadWeights(wts);
    INetworkDefinition* net = builder->createNetworkV2(0U);
    Ctx c{net, wm};
    const float gd = cfg.gd, gw = cfg.gw;
    const int mc = cfg.max_channels;
    auto W = [&](int x) { return get_width(x, gw, mc); };

    ITensor* data = net->addInput("images", DataType::kFLOAT, Dims3{3, cfg.input_h, cfg.input_w});
    assert(data);
    // ---- backbone (model.cpp:115-140)
    ITensor* p1 = convBnSiLU(c, *data, W(64), 3, 2, 1, "model.0");
    ITensor* p2 = convBnSiLU(c, *p1, W(128), 3, 2, 1, "model.1");
    ITensor* c2 = C2F(c, *p2, W(128), get_depth(3, gd), true, 0.5f, "model.2");
    ITensor* p3 = convBnSiLU(c, *c2, W(256), 3, 2, 1, "model.3");
    ITensor* c4 = C2F(c, *p3, W(256), get_depth(6, gd), true, 0.5f, "model.4");
    ITensor* p4 = convBnSiLU(c, *c4, W(512), 3, 2, 1, "model.5");
    ITensor* c6 = C2F(c, *p4, W(512), get_depth(6, gd), true, 0.5f, "model.6");
    ITensor* p5 = convBnSiLU(c, *c6, W(1024), 3, 2, 1, "model.7");
    ITensor* c8 = C2F(c, *p5, W(1024), get_depth(3, gd), true, 0.5f, "model.8");
    ITensor* c9 = SPPF(c, *c8, W(1024), W(1024), 5, "model.9");
    // ---- neck (model.cpp:145-182)
    ITensor* c12 = C2F(c, *cat2(c, upsample2x(c, *c9), c6), W(512), get_depth(3, gd), false, 0.5f, "model.12");
    ITensor* c15 = C2F(c, *cat2(c, upsample2x(c, *c12), c4), W(256), get_depth(3, gd), false, 0.5f, "model.15");
    ITensor* c16 = convBnSiLU(c, *c15, W(256), 3, 2, 1, "model.16");
    ITensor* c18 = C2F(c, *cat2(c, c16, c12), W(512), get_depth(3, gd), false, 0.5f, "model.18");
    ITensor* c19 = convBnSiLU(c, *c18, W(512), 3, 2, 1, "model.19");
    ITensor* c21 = C2F(c, *cat2(c, c19, c9), W(1024), get_depth(3, gd), false, 0.5f, "model.21");

    // ---- detect head (model.cpp:188-251): per level a 64-channel box branch and a num_class branch
    const int base_in = (gw == 1.25f) ? 80 : 64;
    const int base_out = (gw == 0.25f) ? std::max(64, std::min(cfg.num_class, 100)) : W(256);
    ITensor* feats[3] = {c15, c18, c21};
    ITensor* strideRef[3] = {p3, p4, p5};  // strides derived from the backbone maps (model.cpp:27-34, 258-261)
    std::vector<int> strides;
    for (ITensor* t : strideRef) strides.push_back(cfg.input_h / (int)t->getDimensions().d[1]);

    // Layers are added in the reference's own order - all three levels' convolution arms first, then the three DFL tails
    // (model.cpp:188-251, then 263-303) - so that the plan this builder serializes is byte for byte the plan the reference's
    // buildEngineYolov8Det produces through the same shim on the same .wts (tests/test_ref_builders.py).
    std::vector<ITensor*> dets, cats;
    std::vector<std::pair<ITensor*, ITensor*>> tails;
    for (int lv = 0; lv < 3; ++lv) {
        const std::string s = std::to_string(lv);
        ITensor* b = convBnSiLU(c, *feats[lv], base_in, 3, 1, 1, "model.22.cv2." + s + ".0");
        b = convBnSiLU(c, *b, base_in, 3, 1, 1, "model.22.cv2." + s + ".1");
        auto* box = net->addConvolutionNd(*b, 64, DimsHW{1, 1}, need(wm, "model.22.cv2." + s + ".2.weight"),
                                          need(wm, "model.22.cv2." + s + ".2.bias"));
        box->setStrideNd(DimsHW{1, 1});
        box->setPaddingNd(DimsHW{0, 0});
        ITensor* k = convBnSiLU(c, *feats[lv], base_out, 3, 1, 1, "model.22.cv3." + s + ".0");
        k = convBnSiLU(c, *k, base_out, 3, 1, 1, "model.22.cv3." + s + ".1");
        auto* cls = net->addConvolutionNd(*k, cfg.num_class, DimsHW{1, 1}, need(wm, "model.22.cv3." + s + ".2.weight"),
                                          need(wm, "model.22.cv3." + s + ".2.bias"));
        cls->setStrideNd(DimsHW{1, 1});
        cls->setPaddingNd(DimsHW{0, 0});
        cats.push_back(cat2(c, box->getOutput(0), cls->getOutput(0)));
    }
    for (int lv = 0; lv < 3; ++lv) {
        const std::string s = std::to_string(lv);
        // model.cpp:263-303: flatten the grid, split box/cls, DFL the box half, re-join
        const int grid = (cfg.input_h / strides[lv]) * (cfg.input_w / strides[lv]);
        auto* flat = net->addShuffle(*cats[lv]);
        flat->setReshapeDimensions(Dims2{64 + cfg.num_class, grid});
        ITensor* boxPart = net->addSlice(*flat->getOutput(0), Dims2{0, 0}, Dims2{64, grid}, Dims2{1, 1})->getOutput(0);
        ITensor* clsPart = net->addSlice(*flat->getOutput(0), Dims2{64, 0}, Dims2{cfg.num_class, grid}, Dims2{1, 1})->getOutput(0);
        ITensor* dfl = DFL(c, *boxPart, grid, "model.22.dfl.conv.weight");
        if (cfg.task == 0) {
            dets.push_back(cat2(c, dfl, clsPart));
        } else if (cfg.task == 2) {  // pose joins each level right after its DFL: [dfl(4), classes, keypoints] (model.cpp:1472-1531)
            ITensor* v[] = {dfl, clsPart, cv4Branch(c, *feats[lv], "model.22.cv4." + s, grid, cfg)};
            dets.push_back(net->addConcatenation(v, 3)->getOutput(0));
        } else {  // seg / obb finish all three DFL tails first (model.cpp:1218-1250, 2662-2697)
            tails.push_back({dfl, clsPart});
        }
    }
    for (size_t lv = 0; lv < tails.size(); ++lv) {  // seg / obb: [dfl(4), classes, mask coefficients | angle] (model.cpp:1253-1272, 2699-2718)
        const int grid = (cfg.input_h / strides[lv]) * (cfg.input_w / strides[lv]);
        ITensor* v[] = {tails[lv].first, tails[lv].second, cv4Branch(c, *feats[lv], "model.22.cv4." + std::to_string(lv), grid, cfg)};
        dets.push_back(net->addConcatenation(v, 3)->getOutput(0));
    }
    if (cfg.mark_heads)
        for (size_t i = 0; i < dets.size(); ++i) {
            dets[i]->setName(("head" + std::to_string(i)).c_str());
            net->markOutput(*dets[i]);
        }
    IPluginV2Layer* yolo = addYoLoLayer(c, dets, strides, cfg);
    assert(yolo);
    yolo->getOutput(0)->setName("output");
    net->markOutput(*yolo->getOutput(0));
    if (cfg.task == 1) {  // model.cpp:1280-1282
        ITensor* pr = proto(c, *c15, cfg);
        pr->setName("proto");
        net->markOutput(*pr);
    }

    builder->setMaxBatchSize(cfg.max_batch);
    config->setMaxWorkspaceSize(16 * (1 << 20));
    if (cfg.fp16) config->setFlag(BuilderFlag::kFP16);
    IHostMemory* plan = builder->buildSerializedNetwork(*net, *config);
    delete net;
    freeWeights(wm);
    return plan;
}

}  // namespace trtx_host
