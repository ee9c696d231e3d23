// Lowered execution plan: the network definition turned into a static schedule of fused HIP kernel
// launches over a static memory plan.  This is the MI355X-native replacement for what the reference
// hands to TensorRT's builder (IBuilder::buildSerializedNetwork, yolov8/src/model.cpp:327): layer
// fusion (Conv+BN+SiLU/ReLU+residual), layout selection (NHWC fp16 for image tensors, fp32 LINEAR for
// the reshaped heads and plugin edges), concat/slice elimination by channel-offset views, liveness-based
// buffer reuse, weight pre-packing.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../kernels/kernels.h"
#include "graph.h"
#include "plugin.h"

namespace trtx {

enum Layout : int { LAY_LINEAR = 0, LAY_NHWC = 1 };
enum StorageKind : int { ST_ARENA = 0, ST_WEIGHTS = 1, ST_BINDING = 2 };

enum OpKind : int {
    OP_CONV = 0,      // fused conv (+bias/BN, act1, residual, act2): igemm (MFMA) or direct
    OP_DECONV,
    OP_POOL,
    OP_RESIZE,
    OP_EW_NHWC,
    OP_ACT_NHWC,
    OP_SCALE_NHWC,
    OP_COPY_NHWC,
    OP_REDUCE_HW,
    OP_TO_NHWC,       // LINEAR fp32 -> NHWC
    OP_TO_LINEAR,     // NHWC -> LINEAR fp32
    OP_GATHER,        // LINEAR permute / slice / broadcast copy
    OP_SCATTER,       // LINEAR concat placement
    OP_EW_LIN,
    OP_ACT_LIN,
    OP_SCALE_LIN,
    OP_SOFTMAX,
    OP_MATMUL,
    OP_REDUCE_LIN,
    OP_PLUGIN,
    OP_COPY_LIN,      // dense copy (e.g. into an output binding)
    OP_YOLO_HEAD,     // fused DFL + YoloLayer decode on the NHWC head tensors
    OP_POOL_CHAIN,    // three chained k x k stride-1 'same' max-pools (SPPF) in one launch, three outputs
    OP_D2S,           // depth-to-space: [N,H,W,(r,q,c)] -> [N,H*bh,W*bw,c] (second half of a kernel == stride deconvolution)
    OP_ROI_ALIGN,     // detectron2 ROIAlign on the NHWC feature map, NHWC [P][res][res][C] out (fused form of the "RoiAlign" plugin)
    OP_RESERVED_47,   // (rounds 3-4: OP_CONV_CHAIN, the fused convolution chains - tools/hip/experiments/)
    OP_CONV_GROUP,    // 2..4 INDEPENDENT implicit-GEMM convolutions of one kernel instantiation in one launch (POp::group; round 4)
};
const char* op_kind_name(int k);

struct PTensor {
    int id = -1;
    int net_tensor = -1;
    std::string name;
    Dims dims;           // logical dims (implicit batch: per sample)
    bool batched = true; // carries the runtime batch as outermost dimension
    int layout = LAY_LINEAR;
    int dtype = DT_F32;
    int nfix = 0;        // NHWC with explicit batch: N taken from dims[0]; 0 = runtime batch
    int nmul = 1;        // NHWC with a leading per-sample dim (P,C,H,W): image count = batch * nmul
    int C = 0, H = 0, W = 0;
    bool pad_zeroed = false;  // channels [C, ld) of an owning NHWC tensor are guaranteed zero
    // storage
    int parent = -1;     // aliasing: this tensor lives inside `parent` ...
    int coff = 0;        // ... starting at this channel (NHWC) ...
    long eoff = 0;       // ... or element offset (LINEAR reshape/identity views)
    int ld = 0;          // resolved channel stride (NHWC)
    int storage = -1;    // resolved storage id
    int rcoff = 0;       // resolved channel offset within the storage
    long reoff = 0;      // resolved element offset within the storage (LINEAR)
    int Calloc = 0;      // owning NHWC tensor: allocated channel count
    float scale = 0.f;   // DT_I8 tensors: real value = int8 value * scale (the calibrated scale of the owning tensor)
    long sample_elems() const { return layout == LAY_NHWC ? (long)nmul * H * W * ld : (long)dims.volume(); }
};

struct Storage {
    int kind = ST_ARENA;
    size_t bytes = 0;      // total bytes at max batch
    size_t offset = 0;     // within the arena / weights blob
    int binding = -1;      // ST_BINDING
    int first_use = 1 << 30, last_use = -1;
};

struct POp {
    int kind = 0;
    std::string name;
    std::vector<int> in, out;  // plan tensor ids
    std::vector<int> extra_in; // tensors read besides `in` (conv with a folded upsample: the half-resolution source); they count for
                               // dependencies and buffer lifetimes like `in`
    int dtype = DT_F32;
    // conv / deconv
    ConvArgs conv{};
    bool igemm = false;
    bool stem = false;         // conv_stem kernel: reads the LINEAR fp32 input directly
    bool from_deconv = false;  // 1x1 conv standing in for a kernel == stride deconvolution (weights re-laid from CKRS)
    // OP_CONV_GROUP: the member convolutions, each a complete OP_CONV record (its own in / out tensors, ConvArgs, weights); the group's
    // in / out are the unions, so dependencies, lanes and buffer lifetimes see one op
    std::vector<POp> group;
    int src_layer = -1;        // network layer holding the kernel weights
    int scale_layer = -1;      // folded IScaleLayer (BatchNorm) or -1
    size_t w_off = 0, b_off = 0, s_off = 0;  // byte offsets into the device weight blob
    // generic parameters
    int i[12] = {0};
    float f[4] = {0};
    StridedView view{};
    long off0 = 0;             // element offset applied to the strided side of a gather/scatter
    bool view_batched_in = false, view_batched_in2 = false, view_batched_out = true;
    std::shared_ptr<PluginHolder> plugin;
    size_t ws_off = 0, ws_bytes = 0;
    // concurrency: lane = HIP stream the op is issued on; before it, the lane waits for the events of wait_ops;
    // signal = some op on another lane waits for this one (an event is recorded after it)
    int lane = 0;
    std::vector<int> wait_ops;
    bool signal = false;
    double flops = 0;   // algorithmic FLOP per sample (2*MAC)
    double bytes = 0;   // algorithmic bytes per sample (activations in + out) + weights
};

struct Plan {
    bool explicit_batch = false;
    bool fp16 = false;
    int max_batch = 1;
    std::vector<PTensor> tensors;
    std::vector<Storage> storages;
    std::vector<POp> ops;
    std::vector<int> binding_tensor;      // binding index -> network tensor id (inputs first, then outputs)
    std::vector<int> binding_ptensor;     // binding index -> plan tensor id
    std::vector<bool> binding_is_input;
    int num_lanes = 1;
    size_t arena_bytes = 0;
    size_t weight_bytes = 0;
    std::vector<uint8_t> weight_blob;     // host image of the device weight blob (filled by pack_weights)
    std::string error;

    std::string describe_json() const;
};

// Lower a network into a plan (pure host work; no device needed).  Returns false and sets plan.error.
bool lower_network(const Network& net, Plan* plan);
// While > 0 on the calling thread, lower_network keeps every tensor of the definition that a kINT8 engine of it would keep: the passes
// that make a tensor disappear from an fp16 plan but not from the int8 plan (fold_upsample) are off.  Set by
// run_int8_calibration around the statistics engine: the observer sees a tensor only where an op writes it, and the upsampled slice of
// a concat buffer must reach that buffer's histogram (the int8 engine requantises the upsampled feature into the shared
// scale and would clip it if the scale came from the skip slice alone).
struct CalibrationLowering {
    CalibrationLowering();
    ~CalibrationLowering();
    static bool active();
};
// Fill plan.weight_blob (folded BN, packed fp16 igemm weights, biases, constants).
bool pack_weights(const Network& net, Plan* plan);

}  // namespace trtx
