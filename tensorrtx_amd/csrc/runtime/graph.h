// Network-definition IR: what the nvinfer1::INetworkDefinition calls of the reference builders record
// (call-site census: SURVEY.md §2.3).  Shapes are inferred eagerly so ITensor::getDimensions() works
// while the network is being built (e.g. calculateStrides, yolov8/src/model.cpp:27-34).
#pragma once
#include <stdint.h>

#include <array>
#include <memory>
#include <string>
#include <vector>

#include "trtx_hip.h"

namespace trtx {

struct Dims {
    int nb = 0;
    int64_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t volume() const {
        int64_t v = 1;
        for (int i = 0; i < nb; ++i) v *= d[i];
        return v;
    }
    bool operator==(const Dims& o) const {
        if (nb != o.nb) return false;
        for (int i = 0; i < nb; ++i)
            if (d[i] != o.d[i]) return false;
        return true;
    }
};

enum LayerKind : int32_t {
    L_INPUT = 0,  // pseudo layer for network inputs
    L_CONV = 1,
    L_DECONV = 2,
    L_ACTIVATION = 3,
    L_POOLING = 4,
    L_SCALE = 5,
    L_ELEMENTWISE = 6,
    L_CONCAT = 7,
    L_SLICE = 8,
    L_SHUFFLE = 9,
    L_RESIZE = 10,
    L_SOFTMAX = 11,
    L_FULLY_CONNECTED = 12,
    L_MATMUL = 13,
    L_CONSTANT = 14,
    L_REDUCE = 15,
    L_PLUGIN = 16,
    L_IDENTITY = 17,
};

struct PluginHolder;  // runtime/plugin.h

struct TensorDef {
    int id = -1;
    std::string name;
    Dims dims;  // implicit-batch networks: without the batch dimension
    int32_t dtype = TRTX_DTYPE_FLOAT;
    int producer = -1;  // layer index, -1 for none
    int producer_slot = 0;
    bool is_input = false;
    bool is_output = false;
};

struct LayerDef {
    int32_t kind = L_IDENTITY;
    std::string name;
    std::vector<int> inputs;
    std::vector<int> outputs;
    // ---- parameters (meaning depends on kind; unused fields keep defaults) -----------------
    int32_t nb_out = 0;                       // conv/deconv/fc: output maps
    int32_t kernel[2] = {1, 1};               // conv/deconv/pool window (h, w)
    int32_t stride[2] = {1, 1};
    int32_t padding[2] = {0, 0};
    int32_t dilation[2] = {1, 1};
    int32_t groups = 1;
    int32_t op = 0;      // activation type / pooling type / elementwise op / scale mode / reduce op / resize mode
    float alpha = 0.f;   // activation alpha
    float beta = 0.f;
    int32_t axis = -1;   // concat axis / softmax axes bitmask / reduce axes bitmask
    int32_t keep_dims = 0;
    int32_t avg_exclusive = 1;
    int32_t mm_op[2] = {0, 0};                // matmul operand ops
    Dims reshape;                             // shuffle (nb = 0 -> none)
    int32_t perm1[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    int32_t perm2[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    Dims start, size, step;                   // slice
    float scales[8] = {1, 1, 1, 1, 1, 1, 1, 1};  // resize
    int32_t nb_scales = 0;
    Dims out_dims;                            // resize explicit output dims / constant dims
    std::vector<float> w0;                    // conv/deconv/fc kernel | scale: shift | constant values
    std::vector<float> w1;                    // conv/deconv/fc bias   | scale: scale
    std::vector<float> w2;                    //                        | scale: power
    std::shared_ptr<PluginHolder> plugin;     // L_PLUGIN
};

class Network {
   public:
    explicit Network(uint32_t flags) : explicit_batch((flags & 1u) != 0) {}
    bool explicit_batch;
    int max_batch = 1;  // implicit-batch capacity (IBuilder::setMaxBatchSize)
    bool fp16 = false;  // BuilderFlag::kFP16
    int max_aux_streams = -1;  // IBuilderConfig::setMaxAuxStreams: extra streams (lanes) a context may use; -1 = runtime default
    bool int8 = false;  // BuilderFlag::kINT8: tensor_scale holds the calibrated activation scales (0 = not calibrated)
    std::vector<float> tensor_scale;  // per network tensor: real value = int8 value * scale
    // Kernel tactics chosen by timing when the plan was BUILT on a machine with a GPU (TensorRT's builder does the same behind
    // buildSerializedNetwork and stores the result in the engine): (layer signature, launch configuration) pairs, runtime/tune.cpp.
    // A plan that carries them runs the same kernels - and returns the same bits - wherever and however often it is deserialized.
    struct TacticEntry {
        int32_t sig[28];
        int32_t tac[6];
        int32_t ns[2];  // what the builder measured, in nanoseconds: the chosen configuration, the default one (-1: not measured)
    };
    std::vector<TacticEntry> tactics;
    bool tactics_timed = false;  // the builder ran the timing (an empty list then means "the defaults won everywhere")
    std::vector<TensorDef> tensors;
    std::vector<LayerDef> layers;
    std::string error;  // last shape-inference / validation error

    int add_tensor(const Dims& d, int dtype, int producer, int slot);
    int add_input(const char* name, int dtype, const Dims& d);
    // appends the layer, creates its output tensors and infers their shapes; returns layer index or -1
    int add_layer(LayerDef&& l);
    // re-run shape inference for one layer (after a setter changed a parameter)
    bool infer(int layer);
    // re-infer every layer in order; false (and `error`) if any layer is invalid — called at build time
    bool validate();
    bool mark_output(int tensor);
    int find_tensor(const std::string& name) const;
    std::vector<int> input_ids() const;
    std::vector<int> output_ids() const;

    // (de)serialisation of the definition = the engine "plan" payload
    void serialize(std::vector<uint8_t>& out, std::vector<std::array<size_t, 3>>* w_offsets = nullptr) const;
    static std::unique_ptr<Network> deserialize(const uint8_t* data, size_t size, std::string* err);
    // JSON description (tensors, layers, params, weight offsets into the serialized plan) for the test oracle
    std::string describe_json() const;
};

}  // namespace trtx
