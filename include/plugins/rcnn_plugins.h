// Faster R-CNN plugin classes for reference-style host code (rcnn/rcnn.cpp:101-200 constructs these by name and
// hands them to INetworkDefinition::addPluginV2).  Each class keeps the reference's name, constructor arguments,
// tensor contract and registered plugin type; the device work is one call into the C ABI of libtrtx_hip.so.
//
//   class                 reference header                      device entry point
//   RpnDecodePlugin       rcnn/RpnDecodePlugin.h:79-86          trtx_rpn_decode
//   RpnNmsPlugin          rcnn/RpnNmsPlugin.h:52-57             trtx_rpn_nms
//   RoiAlignPlugin        rcnn/RoiAlignPlugin.h:60-68           trtx_roi_align
//   PredictorDecodePlugin rcnn/PredictorDecodePlugin.h:76-84    trtx_predictor_decode
//   BatchedNmsPlugin      rcnn/BatchedNmsPlugin.h:55-62         trtx_batched_nms
//   MaskRcnnInferencePlugin rcnn/MaskRcnnInferencePlugin.h:50-57  trtx_mask_rcnn_inference
//
// These are *user* plugins in the TensorRT sense: they live in the application (here: libtrtx_models.so), reach the
// engine through the IPluginV2 trampoline of include/NvInfer.h and are re-created at deserialization by the creators
// registered below, so the process that deserializes an R-CNN plan must have this header's creators linked in, exactly
// as with the reference.  Serialized parameter blobs use the reference's byte layouts (documented
// per class), so a blob written by a reference plugin loads here and vice versa (tests/test_ref_pinning.py).
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "NvInfer.h"
#include "trtx_hip.h"

namespace nvinfer1 {
namespace rcnn_detail {

struct BlobWriter {
    std::vector<char> bytes;
    template <typename T>
    void put(const T& v) {
        const char* p = reinterpret_cast<const char*>(&v);
        bytes.insert(bytes.end(), p, p + sizeof(T));
    }
};
struct BlobReader {
    const char* p;
    const char* end;
    BlobReader(const void* d, size_t n) : p(static_cast<const char*>(d)), end(p + n) {}
    template <typename T>
    T get() {
        T v{};
        if (p + sizeof(T) <= end) std::memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    // element count read from the blob, clamped to what the blob can still hold (a corrupt count must not allocate)
    template <typename T>
    size_t count(uint64_t n) const {
        const size_t left = p < end ? (size_t)(end - p) / sizeof(T) : 0;
        return n < left ? (size_t)n : left;
    }
};

// Everything the five plugins share: fp32 LINEAR tensors, no broadcast, trivial lifetime, blob = pack().
template <class Derived>
class Base : public IPluginV2Ext {
   public:
    int32_t initialize() noexcept override { return 0; }
    void terminate() noexcept override {}
    size_t getSerializationSize() const noexcept override { return self().pack().size(); }
    void serialize(void* buffer) const noexcept override {
        const std::vector<char> b = self().pack();
        std::memcpy(buffer, b.data(), b.size());
    }
    const char* getPluginVersion() const noexcept override { return "1"; }
    void destroy() noexcept override { delete this; }
    IPluginV2Ext* clone() const noexcept override { return new Derived(self()); }
    void setPluginNamespace(const char*) noexcept override {}
    const char* getPluginNamespace() const noexcept override { return ""; }
    DataType getOutputDataType(int32_t, const DataType*, int32_t) const noexcept override { return DataType::kFLOAT; }
    bool isOutputBroadcastAcrossBatch(int32_t, const bool*, int32_t) const noexcept override { return false; }
    bool canBroadcastInputAcrossBatch(int32_t) const noexcept override { return false; }

   protected:
    const Derived& self() const { return *static_cast<const Derived*>(this); }
};

template <class Plugin>
class Creator : public IPluginCreator {
   public:
    const char* getPluginName() const noexcept override { return Plugin::kType; }
    const char* getPluginVersion() const noexcept override { return "1"; }
    const PluginFieldCollection* getFieldNames() noexcept override { return nullptr; }
    IPluginV2* createPlugin(const char*, const PluginFieldCollection*) noexcept override { return nullptr; }
    IPluginV2* deserializePlugin(const char*, const void* data, size_t length) noexcept override {
        return new Plugin(data, length);
    }
    void setPluginNamespace(const char*) noexcept override {}
    const char* getPluginNamespace() const noexcept override { return ""; }
};

}  // namespace rcnn_detail

// inputs : objectness {A, fh, fw}, anchor deltas {4A, fh, fw}
// outputs: scores {top_n, 1}, boxes {top_n, 4} (XYXY, clipped to the image), best first
// blob   : the reference's byte layout (RpnDecodePlugin.h:41-76): i32 top_n | u64 n_anchor_floats | f32 anchors[] | f32 stride |
//          u64 fh | u64 fw | u64 image_h | u64 image_w  (size_t = u64 on LP64)
class RpnDecodePlugin : public rcnn_detail::Base<RpnDecodePlugin> {
   public:
    static constexpr const char* kType = "RpnDecode";
    RpnDecodePlugin(int top_n, const std::vector<float>& anchors, float stride, size_t image_height, size_t image_width)
        : top_n_(top_n), anchors_(anchors), stride_(stride), image_h_((int)image_height), image_w_((int)image_width) {}
    RpnDecodePlugin(const void* data, size_t length) {
        rcnn_detail::BlobReader r(data, length);
        top_n_ = r.get<int32_t>();
        anchors_.resize(r.count<float>(r.get<uint64_t>()));
        for (float& a : anchors_) a = r.get<float>();
        stride_ = r.get<float>();
        fh_ = (int)r.get<uint64_t>();
        fw_ = (int)r.get<uint64_t>();
        image_h_ = (int)r.get<uint64_t>();
        image_w_ = (int)r.get<uint64_t>();
    }
    std::vector<char> pack() const {
        rcnn_detail::BlobWriter w;
        w.put<int32_t>(top_n_);
        w.put<uint64_t>(anchors_.size());
        for (float a : anchors_) w.put<float>(a);
        w.put<float>(stride_);
        w.put<uint64_t>((uint64_t)fh_);
        w.put<uint64_t>((uint64_t)fw_);
        w.put<uint64_t>((uint64_t)image_h_);
        w.put<uint64_t>((uint64_t)image_w_);
        return w.bytes;
    }
    const char* getPluginType() const noexcept override { return kType; }
    int32_t getNbOutputs() const noexcept override { return 2; }
    Dims getOutputDimensions(int32_t index, const Dims*, int32_t) noexcept override { return Dims2(top_n_, index == 1 ? 4 : 1); }
    void configurePlugin(const Dims* in, int32_t nbInputs, const Dims*, int32_t, const DataType*, const DataType*, const bool*,
                         const bool*, PluginFormat, int32_t) noexcept override {
        if (nbInputs == 2 && in[0].nbDims == 3) {
            fh_ = (int)in[0].d[1];
            fw_ = (int)in[0].d[2];
        }
    }
    size_t getWorkspaceSize(int32_t maxBatch) const noexcept override {
        return trtx_rpn_decode_workspace(maxBatch, (int)anchors_.size() / 4, fh_, fw_);
    }
    int32_t enqueue(int32_t batch, const void* const* in, void* const* out, void* ws, hipStream_t stream) noexcept override {
        return trtx_rpn_decode(batch, static_cast<const float*>(in[0]), static_cast<const float*>(in[1]), fh_, fw_, image_h_,
                               image_w_, stride_, anchors_.data(), (int)anchors_.size() / 4, top_n_,
                               static_cast<float*>(out[0]), static_cast<float*>(out[1]), ws, getWorkspaceSize(batch), stream);
    }

   private:
    int top_n_ = 0;
    std::vector<float> anchors_;
    float stride_ = 1.f;
    int fh_ = 0, fw_ = 0, image_h_ = 0, image_w_ = 0;
};

// inputs : scores {pre, 1}, boxes {pre, 4};  output: boxes {post, 4}
// blob   : the reference's byte layout (RpnNmsPlugin.h:36-53): f32 nms_thresh | i32 post_nms_topk | u64 pre_nms_topk
class RpnNmsPlugin : public rcnn_detail::Base<RpnNmsPlugin> {
   public:
    static constexpr const char* kType = "RpnNms";
    RpnNmsPlugin(float nms_thresh, int post_nms_topk) : thresh_(nms_thresh), post_(post_nms_topk) {}
    RpnNmsPlugin(const void* data, size_t length) {
        rcnn_detail::BlobReader r(data, length);
        thresh_ = r.get<float>();
        post_ = r.get<int32_t>();
        pre_ = (int)r.get<uint64_t>();
    }
    std::vector<char> pack() const {
        rcnn_detail::BlobWriter w;
        w.put<float>(thresh_);
        w.put<int32_t>(post_);
        w.put<uint64_t>((uint64_t)pre_);
        return w.bytes;
    }
    const char* getPluginType() const noexcept override { return kType; }
    int32_t getNbOutputs() const noexcept override { return 1; }
    Dims getOutputDimensions(int32_t, const Dims*, int32_t) noexcept override { return Dims2(post_, 4); }
    void configurePlugin(const Dims* in, int32_t nbInputs, const Dims*, int32_t, const DataType*, const DataType*, const bool*,
                         const bool*, PluginFormat, int32_t) noexcept override {
        if (nbInputs == 2) pre_ = (int)in[0].d[0];
    }
    size_t getWorkspaceSize(int32_t maxBatch) const noexcept override { return trtx_sorted_nms_workspace(maxBatch, pre_); }
    int32_t enqueue(int32_t batch, const void* const* in, void* const* out, void* ws, hipStream_t stream) noexcept override {
        return trtx_rpn_nms(batch, static_cast<const float*>(in[0]), static_cast<const float*>(in[1]), pre_, post_, thresh_,
                            static_cast<float*>(out[0]), ws, getWorkspaceSize(batch), stream);
    }

   private:
    float thresh_ = 0.7f;
    int post_ = 0, pre_ = 1;
};

// inputs : boxes {P, 4}, features {C, fh, fw};  output: {P, C, res, res}
// blob   : the reference's byte layout (RoiAlignPlugin.h:39-45): i32 res | f32 spatial_scale | i32 sampling_ratio | i32 num_proposals | i32 channels | i32 fh | i32 fw
class RoiAlignPlugin : public rcnn_detail::Base<RoiAlignPlugin> {
   public:
    static constexpr const char* kType = "RoiAlign";
    RoiAlignPlugin(int pooler_resolution, float spatial_scale, int sampling_ratio, int num_proposals, int out_channels)
        : res_(pooler_resolution), scale_(spatial_scale), sampling_(sampling_ratio), proposals_(num_proposals),
          channels_(out_channels) {}
    RoiAlignPlugin(const void* data, size_t length) {
        rcnn_detail::BlobReader r(data, length);
        res_ = r.get<int32_t>();
        scale_ = r.get<float>();
        sampling_ = r.get<int32_t>();
        proposals_ = r.get<int32_t>();
        channels_ = r.get<int32_t>();
        fh_ = r.get<int32_t>();
        fw_ = r.get<int32_t>();
    }
    std::vector<char> pack() const {
        rcnn_detail::BlobWriter w;
        w.put<int32_t>(res_);
        w.put<float>(scale_);
        w.put<int32_t>(sampling_);
        w.put<int32_t>(proposals_);
        w.put<int32_t>(channels_);
        w.put<int32_t>(fh_);
        w.put<int32_t>(fw_);
        return w.bytes;
    }
    const char* getPluginType() const noexcept override { return kType; }
    int32_t getNbOutputs() const noexcept override { return 1; }
    Dims getOutputDimensions(int32_t, const Dims*, int32_t) noexcept override { return Dims4(proposals_, channels_, res_, res_); }
    void configurePlugin(const Dims* in, int32_t nbInputs, const Dims*, int32_t, const DataType*, const DataType*, const bool*,
                         const bool*, PluginFormat, int32_t) noexcept override {
        if (nbInputs == 2 && in[1].nbDims == 3) {
            fh_ = (int)in[1].d[1];
            fw_ = (int)in[1].d[2];
        }
    }
    size_t getWorkspaceSize(int32_t) const noexcept override { return 0; }
    int32_t enqueue(int32_t batch, const void* const* in, void* const* out, void*, hipStream_t stream) noexcept override {
        return trtx_roi_align(batch, static_cast<const float*>(in[0]), static_cast<const float*>(in[1]), res_, scale_, sampling_,
                              proposals_, channels_, fh_, fw_, static_cast<float*>(out[0]), stream);
    }

   private:
    int res_ = 0;
    float scale_ = 1.f;
    int sampling_ = 0, proposals_ = 0, channels_ = 0, fh_ = 0, fw_ = 0;
};

// inputs : scores {N, C, 1, 1}, deltas {N, 4C, 1, 1}, proposals {N, 4}
// outputs: scores {N, 1}, boxes {N, 4}, classes {N, 1}
// blob   : the reference's byte layout (PredictorDecodePlugin.h:43-51): u32 num_boxes | u32 num_classes | u32 image_h | u32 image_w |
//          u64 n_weights | f32 bbox_reg_weights[n]
class PredictorDecodePlugin : public rcnn_detail::Base<PredictorDecodePlugin> {
   public:
    static constexpr const char* kType = "PredictorDecode";
    PredictorDecodePlugin(int num_boxes, size_t image_height, size_t image_width, const std::vector<float>& bbox_reg_weights)
        : boxes_(num_boxes), image_h_((int)image_height), image_w_((int)image_width) {
        for (int k = 0; k < 4; ++k) w_[k] = k < (int)bbox_reg_weights.size() ? bbox_reg_weights[k] : 1.f;
    }
    PredictorDecodePlugin(const void* data, size_t length) {
        rcnn_detail::BlobReader r(data, length);
        boxes_ = r.get<int32_t>();
        classes_ = r.get<int32_t>();
        image_h_ = r.get<int32_t>();
        image_w_ = r.get<int32_t>();
        const size_t nw = r.count<float>(r.get<uint64_t>());
        for (size_t k = 0; k < nw; ++k) {
            const float v = r.get<float>();
            if (k < 4) w_[k] = v;
        }
    }
    std::vector<char> pack() const {
        rcnn_detail::BlobWriter w;
        w.put<int32_t>(boxes_);
        w.put<int32_t>(classes_);
        w.put<int32_t>(image_h_);
        w.put<int32_t>(image_w_);
        w.put<uint64_t>(4);
        for (float v : w_) w.put<float>(v);
        return w.bytes;
    }
    const char* getPluginType() const noexcept override { return kType; }
    int32_t getNbOutputs() const noexcept override { return 3; }
    Dims getOutputDimensions(int32_t index, const Dims*, int32_t) noexcept override { return Dims2(boxes_, index == 1 ? 4 : 1); }
    void configurePlugin(const Dims* in, int32_t nbInputs, const Dims*, int32_t, const DataType*, const DataType*, const bool*,
                         const bool*, PluginFormat, int32_t) noexcept override {
        if (nbInputs == 3) {
            boxes_ = (int)in[0].d[0];
            classes_ = (int)in[0].d[1];
        }
    }
    size_t getWorkspaceSize(int32_t maxBatch) const noexcept override {
        return trtx_predictor_decode_workspace(maxBatch, boxes_, classes_);
    }
    int32_t enqueue(int32_t batch, const void* const* in, void* const* out, void* ws, hipStream_t stream) noexcept override {
        return trtx_predictor_decode(batch, static_cast<const float*>(in[0]), static_cast<const float*>(in[1]),
                                     static_cast<const float*>(in[2]), boxes_, classes_, image_h_, image_w_, w_,
                                     static_cast<float*>(out[0]), static_cast<float*>(out[1]), static_cast<float*>(out[2]), ws,
                                     getWorkspaceSize(batch), stream);
    }

   private:
    int boxes_ = 0, classes_ = 1, image_h_ = 0, image_w_ = 0;
    float w_[4] = {1.f, 1.f, 1.f, 1.f};
};

// inputs : scores {N, 1}, boxes {N, 4}, classes {N, 1};  outputs: scores {D, 1}, boxes {D, 4}, classes {D, 1}
// blob   : the reference's byte layout (BatchedNmsPlugin.h:40-43): i32 nms_method | f32 nms_thresh | i32 detections_per_im | u64 count
class BatchedNmsPlugin : public rcnn_detail::Base<BatchedNmsPlugin> {
   public:
    static constexpr const char* kType = "BatchedNms";
    BatchedNmsPlugin(int nms_method, float nms_thresh, int detections_per_im)
        : method_(nms_method), thresh_(nms_thresh), dets_(detections_per_im) {}
    BatchedNmsPlugin(const void* data, size_t length) {
        rcnn_detail::BlobReader r(data, length);
        method_ = r.get<int32_t>();
        thresh_ = r.get<float>();
        dets_ = r.get<int32_t>();
        count_ = (int)r.get<uint64_t>();
    }
    std::vector<char> pack() const {
        rcnn_detail::BlobWriter w;
        w.put<int32_t>(method_);
        w.put<float>(thresh_);
        w.put<int32_t>(dets_);
        w.put<uint64_t>((uint64_t)count_);
        return w.bytes;
    }
    const char* getPluginType() const noexcept override { return kType; }
    int32_t getNbOutputs() const noexcept override { return 3; }
    Dims getOutputDimensions(int32_t index, const Dims*, int32_t) noexcept override { return Dims2(dets_, index == 1 ? 4 : 1); }
    void configurePlugin(const Dims* in, int32_t nbInputs, const Dims*, int32_t, const DataType*, const DataType*, const bool*,
                         const bool*, PluginFormat, int32_t) noexcept override {
        if (nbInputs == 3) count_ = (int)in[0].d[0];
    }
    size_t getWorkspaceSize(int32_t maxBatch) const noexcept override { return trtx_sorted_nms_workspace(maxBatch, count_); }
    int32_t enqueue(int32_t batch, const void* const* in, void* const* out, void* ws, hipStream_t stream) noexcept override {
        return trtx_batched_nms(method_, batch, static_cast<const float*>(in[0]), static_cast<const float*>(in[1]),
                                static_cast<const float*>(in[2]), count_, dets_, thresh_, static_cast<float*>(out[0]),
                                static_cast<float*>(out[1]), static_cast<float*>(out[2]), ws, getWorkspaceSize(batch), stream);
    }

   private:
    int method_ = 1;
    float thresh_ = 0.5f;
    int dets_ = 0, count_ = 1;
};

// inputs : labels {D, 1}, masks {D, C, S, S};  output: {D, 1, S, S} = sigmoid of each detection's own class plane
// blob   : the reference's byte layout (MaskRcnnInferencePlugin.h:31-35): i32 detections_per_im | i32 output_size | i32 num_classes
class MaskRcnnInferencePlugin : public rcnn_detail::Base<MaskRcnnInferencePlugin> {
   public:
    static constexpr const char* kType = "MaskRcnnInference";
    MaskRcnnInferencePlugin(int detections_per_im, int output_size) : dets_(detections_per_im), size_(output_size) {}
    MaskRcnnInferencePlugin(const void* data, size_t length) {
        rcnn_detail::BlobReader r(data, length);
        dets_ = r.get<int32_t>();
        size_ = r.get<int32_t>();
        classes_ = r.get<int32_t>();
    }
    std::vector<char> pack() const {
        rcnn_detail::BlobWriter w;
        w.put<int32_t>(dets_);
        w.put<int32_t>(size_);
        w.put<int32_t>(classes_);
        return w.bytes;
    }
    const char* getPluginType() const noexcept override { return kType; }
    int32_t getNbOutputs() const noexcept override { return 1; }
    Dims getOutputDimensions(int32_t, const Dims*, int32_t) noexcept override { return Dims4(dets_, 1, size_, size_); }
    void configurePlugin(const Dims* in, int32_t nbInputs, const Dims*, int32_t, const DataType*, const DataType*, const bool*,
                         const bool*, PluginFormat, int32_t) noexcept override {
        if (nbInputs == 2 && in[1].nbDims == 4) classes_ = (int)in[1].d[1];
    }
    size_t getWorkspaceSize(int32_t) const noexcept override { return 0; }
    int32_t enqueue(int32_t batch, const void* const* in, void* const* out, void*, hipStream_t stream) noexcept override {
        return trtx_mask_rcnn_inference(batch, static_cast<const float*>(in[0]), static_cast<const float*>(in[1]), dets_, size_,
                                        classes_, static_cast<float*>(out[0]), stream);
    }

   private:
    int dets_ = 0, size_ = 0, classes_ = 1;
};

using RpnDecodePluginCreator = rcnn_detail::Creator<RpnDecodePlugin>;
using MaskRcnnInferencePluginCreator = rcnn_detail::Creator<MaskRcnnInferencePlugin>;
using RpnNmsPluginCreator = rcnn_detail::Creator<RpnNmsPlugin>;
using RoiAlignPluginCreator = rcnn_detail::Creator<RoiAlignPlugin>;
using PredictorDecodePluginCreator = rcnn_detail::Creator<PredictorDecodePlugin>;
using BatchedNmsPluginCreator = rcnn_detail::Creator<BatchedNmsPlugin>;

}  // namespace nvinfer1
