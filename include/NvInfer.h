/*
 * NvInfer.h — header-only C++ shim that gives the C ABI of libtrtx_hip.so (include/trtx_hip.h) the
 * nvinfer1:: spelling the tensorrtx host code is written against, so a reference-style builder such as
 * createLenetEngine (lenet/lenet.cpp:36-155) or buildEngineYolov8Det (yolov8/src/model.cpp:98-336)
 * compiles against the MI355X runtime with only  cudaStream_t -> hipStream_t  / cudaMalloc -> hipMalloc
 * edits.  Covered surface = the subset the five BASELINE configs use
 * (SURVEY.md §2.3 / §8b): IBuilder, IBuilderConfig, INetworkDefinition + 18 add* calls, ITensor, I*Layer,
 * IHostMemory, IRuntime, ICudaEngine, IExecutionContext, ILogger, IPluginV2{,Ext,IOExt}, IPluginCreator,
 * getPluginRegistry(), REGISTER_TENSORRT_PLUGIN.
 *
 * Ownership follows TensorRT: layers/tensors belong to their network; builder, config, network, host
 * memory, runtime, engine and context are released with `delete p` (TRT >= 8) or `p->destroy()` (TRT 7).
 * No arithmetic happens here: every call forwards to the C ABI.
 */
#ifndef TRTX_NVINFER_SHIM_H_
#define TRTX_NVINFER_SHIM_H_

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "trtx_hip.h"

#define NV_TENSORRT_MAJOR 8
#define NV_TENSORRT_MINOR 6
#define NV_TENSORRT_PATCH 1
#define TRTX_HIP_RUNTIME 1

struct cudnnContext;   // only ever passed through as nullptr (IPluginV2Ext::attachToContext)
struct cublasContext;

namespace nvinfer1 {

enum class DataType : int32_t { kFLOAT = 0, kHALF = 1, kINT8 = 2, kINT32 = 3, kBOOL = 4, kUINT8 = 5 };
enum class ActivationType : int32_t { kRELU = 0, kSIGMOID = 1, kTANH = 2, kLEAKY_RELU = 3 };
enum class PoolingType : int32_t { kMAX = 0, kAVERAGE = 1 };
enum class ElementWiseOperation : int32_t { kSUM = 0, kPROD = 1, kMAX = 2, kMIN = 3, kSUB = 4, kDIV = 5, kPOW = 6 };
enum class ScaleMode : int32_t { kUNIFORM = 0, kCHANNEL = 1, kELEMENTWISE = 2 };
enum class ReduceOperation : int32_t { kSUM = 0, kPROD = 1, kMAX = 2, kMIN = 3, kAVG = 4 };
enum class MatrixOperation : int32_t { kNONE = 0, kTRANSPOSE = 1, kVECTOR = 2 };
enum class ResizeMode : int32_t { kNEAREST = 0, kLINEAR = 1 };
enum class BuilderFlag : int32_t { kFP16 = 0, kINT8 = 1 };
enum class NetworkDefinitionCreationFlag : int32_t { kEXPLICIT_BATCH = 0, kSTRONGLY_TYPED = 1 };
enum class TensorFormat : int32_t { kLINEAR = 0, kCHW2 = 1, kHWC8 = 2 };
using PluginFormat = TensorFormat;
enum class MemoryPoolType : int32_t { kWORKSPACE = 0 };
enum class TensorIOMode : int32_t { kNONE = 0, kINPUT = 1, kOUTPUT = 2 };
enum class PluginFieldType : int32_t { kFLOAT16 = 0, kFLOAT32 = 1, kFLOAT64 = 2, kINT8 = 3, kINT16 = 4, kINT32 = 5, kCHAR = 6, kDIMS = 7, kUNKNOWN = 8 };

class Dims {
   public:
    static constexpr int32_t MAX_DIMS = 8;
    int32_t nbDims = 0;
    int64_t d[MAX_DIMS] = {0, 0, 0, 0, 0, 0, 0, 0};
};
class Dims2 : public Dims {
   public:
    Dims2() { nbDims = 2; }
    Dims2(int64_t a, int64_t b) {
        nbDims = 2;
        d[0] = a;
        d[1] = b;
    }
};
class DimsHW : public Dims2 {
   public:
    DimsHW() = default;
    DimsHW(int64_t h, int64_t w) : Dims2(h, w) {}
    int64_t h() const { return d[0]; }
    int64_t w() const { return d[1]; }
};
class Dims3 : public Dims {
   public:
    Dims3() { nbDims = 3; }
    Dims3(int64_t a, int64_t b, int64_t c) {
        nbDims = 3;
        d[0] = a;
        d[1] = b;
        d[2] = c;
    }
};
using DimsCHW = Dims3;
class Dims4 : public Dims {
   public:
    Dims4() { nbDims = 4; }
    Dims4(int64_t a, int64_t b, int64_t c, int64_t e) {
        nbDims = 4;
        d[0] = a;
        d[1] = b;
        d[2] = c;
        d[3] = e;
    }
};

struct Permutation {
    int32_t order[Dims::MAX_DIMS];
};

class Weights {
   public:
    DataType type;
    const void* values;
    int64_t count;
};

class ILogger {
   public:
    enum class Severity : int32_t { kINTERNAL_ERROR = 0, kERROR = 1, kWARNING = 2, kINFO = 3, kVERBOSE = 4 };
    virtual void log(Severity severity, const char* msg) = 0;
    virtual ~ILogger() = default;
};

class IProfiler {
   public:
    virtual void reportLayerTime(const char* layerName, float ms) = 0;
    virtual ~IProfiler() = default;
};

// INT8 calibration interfaces (the reference: yolov8/include/calibrator.h:14-36, yolov8/src/calibrator.cpp:9-74,
// retinaface/calibrator.h).  getBatch hands DEVICE pointers for the named inputs; the cache is opaque text owned by the runtime.
enum class CalibrationAlgoType : int32_t { kLEGACY_CALIBRATION = 0, kENTROPY_CALIBRATION = 1, kENTROPY_CALIBRATION_2 = 2, kMINMAX_CALIBRATION = 3 };
class IInt8Calibrator {
   public:
    virtual int32_t getBatchSize() const = 0;
    virtual bool getBatch(void* bindings[], const char* names[], int32_t nbBindings) = 0;
    virtual const void* readCalibrationCache(size_t& length) = 0;
    virtual void writeCalibrationCache(const void* ptr, size_t length) = 0;
    virtual CalibrationAlgoType getAlgorithm() = 0;
    virtual ~IInt8Calibrator() = default;
};
class IInt8EntropyCalibrator2 : public IInt8Calibrator {
   public:
    CalibrationAlgoType getAlgorithm() override { return CalibrationAlgoType::kENTROPY_CALIBRATION_2; }
};
class IInt8EntropyCalibrator : public IInt8Calibrator {
   public:
    CalibrationAlgoType getAlgorithm() override { return CalibrationAlgoType::kENTROPY_CALIBRATION; }
};
class IInt8MinMaxCalibrator : public IInt8Calibrator {
   public:
    CalibrationAlgoType getAlgorithm() override { return CalibrationAlgoType::kMINMAX_CALIBRATION; }
};
class IGpuAllocator;

namespace shim {
inline trtx_dims to_c(const Dims& d) {
    trtx_dims o{};
    o.nb = d.nbDims;
    for (int i = 0; i < 8; ++i) o.d[i] = d.d[i];
    return o;
}
inline Dims from_c(const trtx_dims& d) {
    Dims o;
    o.nbDims = d.nb;
    for (int i = 0; i < 8; ++i) o.d[i] = d.d[i];
    return o;
}
inline const float* wptr(const Weights& w) { return static_cast<const float*>(w.values); }
inline void require_float(const Weights& w, const char* what) {
    if (w.count > 0 && w.type != DataType::kFLOAT) {
        std::fprintf(stderr, "[NvInfer shim] %s: only fp32 Weights are supported (as every .wts loader produces)\n", what);
        std::abort();
    }
}
}  // namespace shim

class INetworkDefinition;

class ITensor {
   public:
    Dims getDimensions() const noexcept {
        trtx_dims d{};
        trtx_tensor_get_dims(mNet, mId, &d);
        return shim::from_c(d);
    }
    void setName(const char* name) noexcept { trtx_tensor_set_name(mNet, mId, name); }
    const char* getName() const noexcept { return trtx_tensor_get_name(mNet, mId); }
    DataType getType() const noexcept { return DataType::kFLOAT; }
    bool isNetworkInput() const noexcept { return mIsInput; }
    int32_t id() const noexcept { return mId; }  // shim extension
    ITensor(trtx_network* n, int32_t id, bool in = false) : mNet(n), mId(id), mIsInput(in) {}

   private:
    trtx_network* mNet;
    int32_t mId;
    bool mIsInput;
};

class ILayer {
   public:
    virtual ~ILayer() = default;
    ITensor* getOutput(int32_t index) const noexcept;
    int32_t getNbOutputs() const noexcept { return trtx_layer_nb_outputs(mNet, mLayer); }
    void setName(const char* name) noexcept { trtx_layer_set_name(mNet, mLayer, name); }
    ILayer(INetworkDefinition* owner, trtx_network* n, int32_t layer) : mOwner(owner), mNet(n), mLayer(layer) {}

   protected:
    void setInts(int32_t param, std::initializer_list<int32_t> v) noexcept {
        std::vector<int32_t> t(v);
        if (trtx_layer_set_ints(mNet, mLayer, param, t.data(), (int32_t)t.size()) != TRTX_OK)
            std::fprintf(stderr, "[NvInfer shim] layer parameter rejected: %s\n", trtx_network_last_error(mNet));
    }
    void setDimsParam(int32_t param, const Dims& d) noexcept {
        const trtx_dims c = shim::to_c(d);
        if (trtx_layer_set_dims(mNet, mLayer, param, &c) != TRTX_OK)
            std::fprintf(stderr, "[NvInfer shim] layer dims rejected: %s\n", trtx_network_last_error(mNet));
    }
    INetworkDefinition* mOwner;
    trtx_network* mNet;
    int32_t mLayer;
};

#define TRTX_LAYER_CTOR(cls) \
    cls(INetworkDefinition* o, trtx_network* n, int32_t l) : ILayer(o, n, l) {}

class IConvolutionLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IConvolutionLayer)
    void setStrideNd(const Dims& s) noexcept { setInts(TRTX_P_STRIDE, {(int32_t)s.d[0], (int32_t)s.d[1]}); }
    void setPaddingNd(const Dims& p) noexcept { setInts(TRTX_P_PADDING, {(int32_t)p.d[0], (int32_t)p.d[1]}); }
    void setDilationNd(const Dims& p) noexcept { setInts(TRTX_P_DILATION, {(int32_t)p.d[0], (int32_t)p.d[1]}); }
    void setStride(const DimsHW& s) noexcept { setStrideNd(s); }      // TRT <= 7 spelling (model.cpp:207)
    void setPadding(const DimsHW& p) noexcept { setPaddingNd(p); }
    void setDilation(const DimsHW& p) noexcept { setDilationNd(p); }
    void setNbGroups(int32_t g) noexcept { setInts(TRTX_P_GROUPS, {g}); }
};
class IDeconvolutionLayer : public IConvolutionLayer {
   public:
    IDeconvolutionLayer(INetworkDefinition* o, trtx_network* n, int32_t l) : IConvolutionLayer(o, n, l) {}
};
class IFullyConnectedLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IFullyConnectedLayer)
};
class IActivationLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IActivationLayer)
    void setAlpha(float a) noexcept { trtx_layer_set_floats(mNet, mLayer, TRTX_P_ALPHA, &a, 1); }
    void setBeta(float b) noexcept { trtx_layer_set_floats(mNet, mLayer, TRTX_P_BETA, &b, 1); }
};
class IPoolingLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IPoolingLayer)
    void setStrideNd(const Dims& s) noexcept { setInts(TRTX_P_STRIDE, {(int32_t)s.d[0], (int32_t)s.d[1]}); }
    void setPaddingNd(const Dims& p) noexcept { setInts(TRTX_P_PADDING, {(int32_t)p.d[0], (int32_t)p.d[1]}); }
    void setStride(const DimsHW& s) noexcept { setStrideNd(s); }
    void setPadding(const DimsHW& p) noexcept { setPaddingNd(p); }
    void setAverageCountExcludesPadding(bool e) noexcept { setInts(TRTX_P_AVG_EXCLUSIVE, {e ? 1 : 0}); }
};
class IScaleLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IScaleLayer)
};
class IElementWiseLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IElementWiseLayer)
};
class IConcatenationLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IConcatenationLayer)
    void setAxis(int32_t a) noexcept { setInts(TRTX_P_AXIS, {a}); }
};
class ISliceLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(ISliceLayer)
};
class IShuffleLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IShuffleLayer)
    void setReshapeDimensions(const Dims& d) noexcept { setDimsParam(TRTX_P_RESHAPE, d); }
    void setFirstTranspose(const Permutation& p) noexcept {
        trtx_layer_set_ints(mNet, mLayer, TRTX_P_FIRST_TRANSPOSE, p.order, permLen());
    }
    void setSecondTranspose(const Permutation& p) noexcept {
        trtx_layer_set_ints(mNet, mLayer, TRTX_P_SECOND_TRANSPOSE, p.order, permLen(true));
    }

   private:
    // Permutation{1,0,2} leaves the trailing entries zero-initialised: pass only as many as the tensor has
    int32_t permLen(bool second = false) const noexcept;
};
class IResizeLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IResizeLayer)
    void setResizeMode(ResizeMode m) noexcept { setInts(TRTX_P_RESIZE_MODE, {(int32_t)m}); }
    void setScales(const float* scales, int32_t nb) noexcept { trtx_layer_set_floats(mNet, mLayer, TRTX_P_RESIZE_SCALES, scales, nb); }
    void setOutputDimensions(const Dims& d) noexcept { setDimsParam(TRTX_P_RESIZE_OUT_DIMS, d); }
    void setAlignCorners(bool) noexcept {}
};
class ISoftMaxLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(ISoftMaxLayer)
    void setAxes(uint32_t axes) noexcept { setInts(TRTX_P_AXIS, {(int32_t)axes}); }
};
class IMatrixMultiplyLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IMatrixMultiplyLayer)
};
class IConstantLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IConstantLayer)
};
class IReduceLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IReduceLayer)
};
class IIdentityLayer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IIdentityLayer)
};
class IPluginV2Layer : public ILayer {
   public:
    TRTX_LAYER_CTOR(IPluginV2Layer)
};

// ------------------------------------------------------------------------------------------- plugins
struct PluginTensorDesc {
    Dims dims;
    DataType type;
    TensorFormat format;
    float scale;
};
struct PluginField {
    const char* name = nullptr;
    const void* data = nullptr;
    PluginFieldType type = PluginFieldType::kUNKNOWN;
    int32_t length = 0;
    PluginField() = default;
    PluginField(const char* n, const void* d, PluginFieldType t, int32_t l) : name(n), data(d), type(t), length(l) {}
};
struct PluginFieldCollection {
    int32_t nbFields = 0;
    const PluginField* fields = nullptr;
};

class IPluginV2 {
   public:
    virtual ~IPluginV2() = default;
    virtual int32_t getNbOutputs() const = 0;
    virtual Dims getOutputDimensions(int32_t index, const Dims* inputs, int32_t nbInputDims) = 0;
    virtual int32_t initialize() = 0;
    virtual void terminate() = 0;
    virtual size_t getWorkspaceSize(int32_t maxBatchSize) const = 0;
    virtual int32_t enqueue(int32_t batchSize, const void* const* inputs, void* const* outputs, void* workspace,
                            hipStream_t stream) = 0;
    virtual size_t getSerializationSize() const = 0;
    virtual void serialize(void* buffer) const = 0;
    virtual const char* getPluginType() const = 0;
    virtual const char* getPluginVersion() const = 0;
    virtual void destroy() = 0;
    virtual IPluginV2* clone() const = 0;
    virtual void setPluginNamespace(const char* ns) = 0;
    virtual const char* getPluginNamespace() const = 0;
    // TRT <= 7 only; default keeps old-style plugins compiling
    virtual bool supportsFormat(DataType type, PluginFormat format) const {
        return type == DataType::kFLOAT && format == PluginFormat::kLINEAR;
    }
};

class IPluginV2Ext : public IPluginV2 {
   public:
    virtual DataType getOutputDataType(int32_t index, const DataType* inputTypes, int32_t nbInputs) const = 0;
    virtual bool isOutputBroadcastAcrossBatch(int32_t outputIndex, const bool* inputIsBroadcasted, int32_t nbInputs) const = 0;
    virtual bool canBroadcastInputAcrossBatch(int32_t inputIndex) const = 0;
    // the 10-argument form the R-CNN plugins learn their shapes from (rcnn/RpnDecodePlugin.h:158-171)
    virtual void configurePlugin(const Dims* inputDims, int32_t nbInputs, const Dims* outputDims, int32_t nbOutputs,
                                 const DataType* inputTypes, const DataType* outputTypes, const bool* inputIsBroadcast,
                                 const bool* outputIsBroadcast, PluginFormat floatFormat, int32_t maxBatchSize) {}
    virtual void attachToContext(cudnnContext*, cublasContext*, IGpuAllocator*) {}
    virtual void detachFromContext() {}
    IPluginV2Ext* clone() const override = 0;
};

class IPluginV2IOExt : public IPluginV2Ext {
   public:
    virtual void configurePlugin(const PluginTensorDesc* in, int32_t nbInput, const PluginTensorDesc* out, int32_t nbOutput) = 0;
    virtual bool supportsFormatCombination(int32_t pos, const PluginTensorDesc* inOut, int32_t nbInputs, int32_t nbOutputs) const = 0;
    using IPluginV2Ext::configurePlugin;
    IPluginV2IOExt* clone() const override = 0;
};

class IPluginCreator {
   public:
    virtual ~IPluginCreator() = default;
    virtual const char* getPluginName() const = 0;
    virtual const char* getPluginVersion() const = 0;
    virtual const PluginFieldCollection* getFieldNames() = 0;
    virtual IPluginV2* createPlugin(const char* name, const PluginFieldCollection* fc) = 0;
    virtual IPluginV2* deserializePlugin(const char* name, const void* serialData, size_t serialLength) = 0;
    virtual void setPluginNamespace(const char* ns) = 0;
    virtual const char* getPluginNamespace() const = 0;
};

namespace shim {

// ---- C++ plugin object -> C v-table (user plugins handed to addPluginV2 / registered creators) --------
inline void fill_vtbl(trtx_plugin_vtbl* v, IPluginV2* p);

inline int32_t tp_nb_outputs(void* s) { return static_cast<IPluginV2*>(s)->getNbOutputs(); }
inline int32_t tp_output_dims(void* s, int32_t idx, const trtx_dims* in, int32_t nb, trtx_dims* out) {
    std::vector<Dims> d(nb > 0 ? nb : 1);
    for (int i = 0; i < nb; ++i) d[i] = from_c(in[i]);
    *out = to_c(static_cast<IPluginV2*>(s)->getOutputDimensions(idx, d.data(), nb));
    return 0;
}
inline int32_t tp_configure(void* s, const trtx_dims* in, int32_t nbi, const trtx_dims* out, int32_t nbo, int32_t maxBatch) {
    IPluginV2* p = static_cast<IPluginV2*>(s);
    if (auto* io = dynamic_cast<IPluginV2IOExt*>(p)) {
        std::vector<PluginTensorDesc> di(nbi > 0 ? nbi : 1), dout(nbo > 0 ? nbo : 1);
        for (int i = 0; i < nbi; ++i) di[i] = PluginTensorDesc{from_c(in[i]), DataType::kFLOAT, TensorFormat::kLINEAR, 1.f};
        for (int i = 0; i < nbo; ++i) dout[i] = PluginTensorDesc{from_c(out[i]), DataType::kFLOAT, TensorFormat::kLINEAR, 1.f};
        io->configurePlugin(di.data(), nbi, dout.data(), nbo);
    } else if (auto* ext = dynamic_cast<IPluginV2Ext*>(p)) {
        std::vector<Dims> di(nbi > 0 ? nbi : 1), dout(nbo > 0 ? nbo : 1);
        for (int i = 0; i < nbi; ++i) di[i] = from_c(in[i]);
        for (int i = 0; i < nbo; ++i) dout[i] = from_c(out[i]);
        std::vector<DataType> ti(nbi > 0 ? nbi : 1, DataType::kFLOAT), tout(nbo > 0 ? nbo : 1, DataType::kFLOAT);
        std::unique_ptr<bool[]> bi(new bool[nbi > 0 ? nbi : 1]()), bo(new bool[nbo > 0 ? nbo : 1]());
        ext->configurePlugin(di.data(), nbi, dout.data(), nbo, ti.data(), tout.data(), bi.get(), bo.get(),
                             PluginFormat::kLINEAR, maxBatch);
    }
    return 0;
}
inline int32_t tp_initialize(void* s) { return static_cast<IPluginV2*>(s)->initialize(); }
inline void tp_terminate(void* s) { static_cast<IPluginV2*>(s)->terminate(); }
inline size_t tp_workspace(void* s, int32_t mb) { return static_cast<IPluginV2*>(s)->getWorkspaceSize(mb); }
inline int32_t tp_enqueue(void* s, int32_t batch, const void* const* in, void* const* out, void* ws, trtx_stream_t st) {
    return static_cast<IPluginV2*>(s)->enqueue(batch, in, out, ws, reinterpret_cast<hipStream_t>(st));
}
inline size_t tp_ser_size(void* s) { return static_cast<IPluginV2*>(s)->getSerializationSize(); }
inline void tp_serialize(void* s, void* b) { static_cast<IPluginV2*>(s)->serialize(b); }
inline const char* tp_type(void* s) { return static_cast<IPluginV2*>(s)->getPluginType(); }
inline const char* tp_version(void* s) { return static_cast<IPluginV2*>(s)->getPluginVersion(); }
inline int32_t tp_clone(void* s, trtx_plugin_vtbl* out) {
    IPluginV2* c = static_cast<IPluginV2*>(s)->clone();
    if (!c) return 1;
    fill_vtbl(out, c);
    return 0;
}
inline void tp_destroy(void* s) { static_cast<IPluginV2*>(s)->destroy(); }
inline void fill_vtbl(trtx_plugin_vtbl* v, IPluginV2* p) {
    v->self = p;
    v->get_nb_outputs = tp_nb_outputs;
    v->get_output_dims = tp_output_dims;
    v->configure = tp_configure;
    v->initialize = tp_initialize;
    v->terminate = tp_terminate;
    v->workspace_size = tp_workspace;
    v->enqueue = tp_enqueue;
    v->serialization_size = tp_ser_size;
    v->serialize = tp_serialize;
    v->plugin_type = tp_type;
    v->plugin_version = tp_version;
    v->clone = tp_clone;
    v->destroy = tp_destroy;
}

// ---- C v-table plugin (built-in HIP plugin) -> C++ object handed back by createPlugin --------------------
class CPluginAdapter : public IPluginV2IOExt {
   public:
    explicit CPluginAdapter(const trtx_plugin_vtbl& v) : mV(v) {}
    ~CPluginAdapter() override {
        if (mV.destroy) mV.destroy(mV.self);
    }
    const trtx_plugin_vtbl& vtbl() const { return mV; }
    int32_t getNbOutputs() const noexcept override { return mV.get_nb_outputs(mV.self); }
    Dims getOutputDimensions(int32_t index, const Dims* inputs, int32_t nb) noexcept override {
        std::vector<trtx_dims> in(nb > 0 ? nb : 1);
        for (int i = 0; i < nb; ++i) in[i] = to_c(inputs[i]);
        trtx_dims o{};
        mV.get_output_dims(mV.self, index, in.data(), nb, &o);
        return from_c(o);
    }
    int32_t initialize() noexcept override { return mV.initialize(mV.self); }
    void terminate() noexcept override { mV.terminate(mV.self); }
    size_t getWorkspaceSize(int32_t mb) const noexcept override { return mV.workspace_size(mV.self, mb); }
    int32_t enqueue(int32_t b, const void* const* in, void* const* out, void* ws, hipStream_t s) noexcept override {
        return mV.enqueue(mV.self, b, in, out, ws, reinterpret_cast<trtx_stream_t>(s));
    }
    size_t getSerializationSize() const noexcept override { return mV.serialization_size(mV.self); }
    void serialize(void* b) const noexcept override { mV.serialize(mV.self, b); }
    const char* getPluginType() const noexcept override { return mV.plugin_type(mV.self); }
    const char* getPluginVersion() const noexcept override { return mV.plugin_version(mV.self); }
    void destroy() noexcept override { delete this; }
    IPluginV2IOExt* clone() const noexcept override {
        trtx_plugin_vtbl c{};
        if (mV.clone(mV.self, &c) != 0) return nullptr;
        return new CPluginAdapter(c);
    }
    void setPluginNamespace(const char* ns) noexcept override { mNs = ns ? ns : ""; }
    const char* getPluginNamespace() const noexcept override { return mNs.c_str(); }
    DataType getOutputDataType(int32_t, const DataType*, int32_t) const noexcept override { return DataType::kFLOAT; }
    bool isOutputBroadcastAcrossBatch(int32_t, const bool*, int32_t) const noexcept override { return false; }
    bool canBroadcastInputAcrossBatch(int32_t) const noexcept override { return false; }
    void configurePlugin(const PluginTensorDesc*, int32_t, const PluginTensorDesc*, int32_t) noexcept override {}
    bool supportsFormatCombination(int32_t pos, const PluginTensorDesc* io, int32_t, int32_t) const noexcept override {
        return io[pos].format == TensorFormat::kLINEAR && io[pos].type == DataType::kFLOAT;
    }

   private:
    trtx_plugin_vtbl mV;
    std::string mNs;
};

class CCreatorAdapter : public IPluginCreator {
   public:
    explicit CCreatorAdapter(const trtx_creator_vtbl& c) : mC(c) {}
    const char* getPluginName() const noexcept override { return mC.plugin_name(mC.self); }
    const char* getPluginVersion() const noexcept override { return mC.plugin_version(mC.self); }
    const PluginFieldCollection* getFieldNames() noexcept override { return &mFC; }
    IPluginV2* createPlugin(const char* name, const PluginFieldCollection* fc) noexcept override {
        std::vector<trtx_plugin_field> f(fc && fc->nbFields > 0 ? fc->nbFields : 0);
        for (size_t i = 0; i < f.size(); ++i)
            f[i] = trtx_plugin_field{fc->fields[i].name, fc->fields[i].data, (int32_t)fc->fields[i].type, fc->fields[i].length};
        trtx_plugin_vtbl v{};
        if (!mC.create || mC.create(mC.self, name, f.data(), (int32_t)f.size(), &v) != 0) return nullptr;
        return new CPluginAdapter(v);
    }
    IPluginV2* deserializePlugin(const char* name, const void* data, size_t len) noexcept override {
        trtx_plugin_vtbl v{};
        if (mC.deserialize(mC.self, name, data, len, &v) != 0) return nullptr;
        return new CPluginAdapter(v);
    }
    void setPluginNamespace(const char* ns) noexcept override { mNs = ns ? ns : ""; }
    const char* getPluginNamespace() const noexcept override { return mNs.c_str(); }

   private:
    trtx_creator_vtbl mC;
    PluginFieldCollection mFC{};
    std::string mNs;
};

// C creator trampolines for a registered C++ IPluginCreator
inline const char* tc_name(void* s) { return static_cast<IPluginCreator*>(s)->getPluginName(); }
inline const char* tc_version(void* s) { return static_cast<IPluginCreator*>(s)->getPluginVersion(); }
inline int32_t tc_create(void* s, const char* name, const trtx_plugin_field* f, int32_t nb, trtx_plugin_vtbl* out) {
    std::vector<PluginField> pf(nb > 0 ? nb : 0);
    for (int i = 0; i < nb; ++i) pf[i] = PluginField(f[i].name, f[i].data, (PluginFieldType)f[i].type, f[i].length);
    PluginFieldCollection fc{nb, pf.data()};
    IPluginV2* p = static_cast<IPluginCreator*>(s)->createPlugin(name, &fc);
    if (!p) return 1;
    fill_vtbl(out, p);
    return 0;
}
inline int32_t tc_deserialize(void* s, const char* name, const void* data, size_t len, trtx_plugin_vtbl* out) {
    IPluginV2* p = static_cast<IPluginCreator*>(s)->deserializePlugin(name, data, len);
    if (!p) return 1;
    fill_vtbl(out, p);
    return 0;
}
}  // namespace shim

class IPluginRegistry {
   public:
    bool registerCreator(IPluginCreator& creator, const char* pluginNamespace) noexcept {
        creator.setPluginNamespace(pluginNamespace ? pluginNamespace : "");
        mCpp.push_back(&creator);
        trtx_creator_vtbl c{};
        c.self = &creator;
        c.plugin_name = shim::tc_name;
        c.plugin_version = shim::tc_version;
        c.create = shim::tc_create;
        c.deserialize = shim::tc_deserialize;
        return trtx_registry_register(&c) == TRTX_OK;
    }
    IPluginCreator* getPluginCreator(const char* type, const char* version, const char* ns = "") noexcept {
        for (IPluginCreator* c : mCpp)
            if (!std::strcmp(c->getPluginName(), type) && !std::strcmp(c->getPluginVersion(), version)) return c;
        trtx_creator_vtbl cv{};
        if (trtx_registry_get(type, version, &cv) != TRTX_OK) return nullptr;
        mAdapters.emplace_back(new shim::CCreatorAdapter(cv));
        return mAdapters.back().get();
    }

   private:
    std::vector<IPluginCreator*> mCpp;
    std::vector<std::unique_ptr<shim::CCreatorAdapter>> mAdapters;
};

}  // namespace nvinfer1

// TensorRT declares the registry accessor at GLOBAL scope (extern "C" in NvInferRuntimeCommon.h); reference code calls it unqualified
// from files without a using-directive (yolov8/src/block.cpp:263) as well as from inside `using namespace nvinfer1`.
inline nvinfer1::IPluginRegistry* getPluginRegistry() noexcept {
    static nvinfer1::IPluginRegistry r;
    return &r;
}

namespace nvinfer1 {
using ::getPluginRegistry;  // the same entity under both spellings: no ambiguity next to a using-directive

template <typename T>
class PluginRegistrar {
   public:
    PluginRegistrar() { getPluginRegistry()->registerCreator(instance, ""); }

   private:
    T instance{};
};
#define REGISTER_TENSORRT_PLUGIN(name) static nvinfer1::PluginRegistrar<name> pluginRegistrar##name {}

// ------------------------------------------------------------------------------------ network definition
class INetworkDefinition {
   public:
    INetworkDefinition(trtx_network* n) : mNet(n) {}
    virtual ~INetworkDefinition() { trtx_network_destroy(mNet); }
    void destroy() noexcept { delete this; }
    trtx_network* handle() const noexcept { return mNet; }

    ITensor* addInput(const char* name, DataType type, const Dims& dims) noexcept {
        const trtx_dims d = shim::to_c(dims);
        const int32_t id = trtx_add_input(mNet, name, (int32_t)type, &d);
        return id < 0 ? nullptr : tensor(id, true);
    }
    void markOutput(ITensor& t) noexcept { trtx_mark_output(mNet, t.id()); }

    IConvolutionLayer* addConvolutionNd(ITensor& in, int32_t nbOut, const Dims& k, Weights kernel, Weights bias) noexcept {
        shim::require_float(kernel, "addConvolutionNd");
        shim::require_float(bias, "addConvolutionNd");
        return layer<IConvolutionLayer>(trtx_add_convolution(mNet, in.id(), nbOut, (int32_t)k.d[0], (int32_t)k.d[1], shim::wptr(kernel),
                                                             kernel.count, shim::wptr(bias), bias.count));
    }
    IConvolutionLayer* addConvolution(ITensor& in, int32_t nbOut, DimsHW k, Weights kernel, Weights bias) noexcept {
        return addConvolutionNd(in, nbOut, k, kernel, bias);
    }
    IDeconvolutionLayer* addDeconvolutionNd(ITensor& in, int32_t nbOut, const Dims& k, Weights kernel, Weights bias) noexcept {
        shim::require_float(kernel, "addDeconvolutionNd");
        return layer<IDeconvolutionLayer>(trtx_add_deconvolution(mNet, in.id(), nbOut, (int32_t)k.d[0], (int32_t)k.d[1],
                                                                 shim::wptr(kernel), kernel.count, shim::wptr(bias), bias.count));
    }
    IDeconvolutionLayer* addDeconvolution(ITensor& in, int32_t nbOut, DimsHW k, Weights kernel, Weights bias) noexcept {
        return addDeconvolutionNd(in, nbOut, k, kernel, bias);
    }
    IFullyConnectedLayer* addFullyConnected(ITensor& in, int32_t nbOut, Weights kernel, Weights bias) noexcept {
        shim::require_float(kernel, "addFullyConnected");
        return layer<IFullyConnectedLayer>(
                trtx_add_fully_connected(mNet, in.id(), nbOut, shim::wptr(kernel), kernel.count, shim::wptr(bias), bias.count));
    }
    IActivationLayer* addActivation(ITensor& in, ActivationType t) noexcept {
        return layer<IActivationLayer>(trtx_add_activation(mNet, in.id(), (int32_t)t));
    }
    IPoolingLayer* addPoolingNd(ITensor& in, PoolingType t, const Dims& w) noexcept {
        return layer<IPoolingLayer>(trtx_add_pooling(mNet, in.id(), (int32_t)t, (int32_t)w.d[0], (int32_t)w.d[1]));
    }
    IPoolingLayer* addPooling(ITensor& in, PoolingType t, DimsHW w) noexcept {
        IPoolingLayer* p = addPoolingNd(in, t, w);
        if (p) p->setStrideNd(w);  // the legacy call defaults the stride to the window size
        return p;
    }
    IScaleLayer* addScale(ITensor& in, ScaleMode mode, Weights shift, Weights scale, Weights power) noexcept {
        shim::require_float(shift, "addScale");
        shim::require_float(scale, "addScale");
        shim::require_float(power, "addScale");
        return layer<IScaleLayer>(trtx_add_scale(mNet, in.id(), (int32_t)mode, shim::wptr(shift), shift.count, shim::wptr(scale),
                                                 scale.count, shim::wptr(power), power.count));
    }
    IElementWiseLayer* addElementWise(ITensor& a, ITensor& b, ElementWiseOperation op) noexcept {
        return layer<IElementWiseLayer>(trtx_add_elementwise(mNet, a.id(), b.id(), (int32_t)op));
    }
    IConcatenationLayer* addConcatenation(ITensor* const* inputs, int32_t nb) noexcept {
        std::vector<int32_t> ids(nb > 0 ? nb : 0);
        for (int i = 0; i < nb; ++i) ids[i] = inputs[i]->id();
        return layer<IConcatenationLayer>(trtx_add_concatenation(mNet, ids.data(), nb));
    }
    ISliceLayer* addSlice(ITensor& in, const Dims& start, const Dims& size, const Dims& stride) noexcept {
        const trtx_dims a = shim::to_c(start), b = shim::to_c(size), c = shim::to_c(stride);
        return layer<ISliceLayer>(trtx_add_slice(mNet, in.id(), &a, &b, &c));
    }
    IShuffleLayer* addShuffle(ITensor& in) noexcept { return layer<IShuffleLayer>(trtx_add_shuffle(mNet, in.id())); }
    IResizeLayer* addResize(ITensor& in) noexcept { return layer<IResizeLayer>(trtx_add_resize(mNet, in.id())); }
    ISoftMaxLayer* addSoftMax(ITensor& in) noexcept { return layer<ISoftMaxLayer>(trtx_add_softmax(mNet, in.id())); }
    IMatrixMultiplyLayer* addMatrixMultiply(ITensor& a, MatrixOperation opA, ITensor& b, MatrixOperation opB) noexcept {
        return layer<IMatrixMultiplyLayer>(trtx_add_matrix_multiply(mNet, a.id(), (int32_t)opA, b.id(), (int32_t)opB));
    }
    IConstantLayer* addConstant(const Dims& dims, Weights w) noexcept {
        shim::require_float(w, "addConstant");
        const trtx_dims d = shim::to_c(dims);
        return layer<IConstantLayer>(trtx_add_constant(mNet, &d, shim::wptr(w), w.count));
    }
    IReduceLayer* addReduce(ITensor& in, ReduceOperation op, uint32_t axes, bool keepDims) noexcept {
        return layer<IReduceLayer>(trtx_add_reduce(mNet, in.id(), (int32_t)op, axes, keepDims ? 1 : 0));
    }
    IIdentityLayer* addIdentity(ITensor& in) noexcept { return layer<IIdentityLayer>(trtx_add_identity(mNet, in.id())); }
    IPluginV2Layer* addPluginV2(ITensor* const* inputs, int32_t nb, IPluginV2& plugin) noexcept {
        std::vector<int32_t> ids(nb > 0 ? nb : 0);
        for (int i = 0; i < nb; ++i) ids[i] = inputs[i]->id();
        trtx_plugin_vtbl v{};
        if (auto* a = dynamic_cast<shim::CPluginAdapter*>(&plugin))
            v = a->vtbl();  // built-in HIP plugin: hand the runtime its own v-table (it clones it)
        else
            shim::fill_vtbl(&v, &plugin);
        return layer<IPluginV2Layer>(trtx_add_plugin_v2(mNet, ids.data(), nb, &v));
    }

    // shim internals (used by ILayer::getOutput)
    ITensor* tensor(int32_t id, bool isInput = false) {
        if (id < 0) return nullptr;
        if ((size_t)id >= mTensors.size()) mTensors.resize(id + 1);
        if (!mTensors[id]) mTensors[id].reset(new ITensor(mNet, id, isInput));
        return mTensors[id].get();
    }

   private:
    template <typename L>
    L* layer(int32_t idx) {
        if (idx < 0) {
            std::fprintf(stderr, "[NvInfer shim] add layer failed: %s\n", trtx_network_last_error(mNet));
            return nullptr;
        }
        L* l = new L(this, mNet, idx);
        mLayers.emplace_back(l);
        return l;
    }
    trtx_network* mNet;
    std::vector<std::unique_ptr<ILayer>> mLayers;
    std::vector<std::unique_ptr<ITensor>> mTensors;
};

inline ITensor* ILayer::getOutput(int32_t index) const noexcept {
    return mOwner->tensor(trtx_layer_output(mNet, mLayer, index));
}
inline int32_t IShuffleLayer::permLen(bool) const noexcept { return 8; }

// ------------------------------------------------------------------------------------------ runtime side
class IHostMemory {
   public:
    explicit IHostMemory(trtx_hostmem* m) : mM(m) {}
    virtual ~IHostMemory() { trtx_hostmem_destroy(mM); }
    void* data() const noexcept { return const_cast<void*>(trtx_hostmem_data(mM)); }
    size_t size() const noexcept { return trtx_hostmem_size(mM); }
    DataType type() const noexcept { return DataType::kINT8; }
    void destroy() noexcept { delete this; }

   private:
    trtx_hostmem* mM;
};

class ICudaEngine;

class IExecutionContext {
   public:
    IExecutionContext(trtx_context* c, ICudaEngine* e) : mC(c), mE(e) {}
    virtual ~IExecutionContext() { trtx_context_destroy(mC); }
    void destroy() noexcept { delete this; }
    bool enqueue(int32_t batchSize, void* const* bindings, hipStream_t stream, hipEvent_t* /*inputConsumed*/) noexcept {
        if (mProfiler) return profiled(batchSize, bindings, stream);
        return trtx_context_enqueue(mC, batchSize, bindings, reinterpret_cast<trtx_stream_t>(stream)) == TRTX_OK;
    }
    bool enqueueV2(void* const* bindings, hipStream_t stream, hipEvent_t* e) noexcept { return enqueue(1, bindings, stream, e); }
    bool execute(int32_t batchSize, void* const* bindings) noexcept {
        const bool ok = enqueue(batchSize, bindings, nullptr, nullptr);
        return ok && hipStreamSynchronize(nullptr) == hipSuccess;
    }
    bool executeV2(void* const* bindings) noexcept { return execute(1, bindings); }
    bool setTensorAddress(const char* name, void* data) noexcept { return trtx_context_set_tensor_address(mC, name, data) == TRTX_OK; }
    bool enqueueV3(hipStream_t stream) noexcept { return trtx_context_enqueue_v3(mC, reinterpret_cast<trtx_stream_t>(stream)) == TRTX_OK; }
    const ICudaEngine& getEngine() const noexcept { return *mE; }
    void setProfiler(IProfiler* p) noexcept { mProfiler = p; }
    trtx_context* handle() const noexcept { return mC; }

   private:
    bool profiled(int32_t batch, void* const* bindings, hipStream_t stream) noexcept {
        char* js = nullptr;
        if (trtx_context_profile(mC, batch, bindings, reinterpret_cast<trtx_stream_t>(stream), &js) != TRTX_OK) return false;
        // [{"name":"...","kind":"...","ms":x}, ...]
        for (const char* p = js; (p = std::strstr(p, "\"name\":\"")) != nullptr;) {
            p += 8;
            const char* e = std::strchr(p, '"');
            std::string name(p, e);
            const char* m = std::strstr(e, "\"ms\":");
            mProfiler->reportLayerTime(name.c_str(), m ? (float)std::atof(m + 5) : 0.f);
            p = e;
        }
        trtx_string_free(js);
        return true;
    }
    trtx_context* mC;
    ICudaEngine* mE;
    IProfiler* mProfiler = nullptr;
};

class ICudaEngine {
   public:
    explicit ICudaEngine(trtx_engine* e) : mE(e) {}
    // An engine that holds its plan and materialises on the device at first use.  What IBuilder::buildEngineWithConfig hands out on a
    // machine without a HIP device, so that the reference's build flow createEngine(...) -> engine->serialize() -> write file
    // (resnet/resnet50.cpp:231-246, retinaface/retina_r50.cpp:244-262, rcnn/rcnn.cpp:310-326) works where engines are only built.
    ICudaEngine(const void* plan, size_t size) : mE(nullptr), mPlan(static_cast<const uint8_t*>(plan), static_cast<const uint8_t*>(plan) + size) {}
    virtual ~ICudaEngine() { trtx_engine_destroy(mE); }
    void destroy() noexcept { delete this; }
    IExecutionContext* createExecutionContext() noexcept {
        trtx_context* c = nullptr;
        if (trtx_context_create(eng(), &c) != TRTX_OK) return nullptr;
        return new IExecutionContext(c, this);
    }
    int32_t getNbBindings() const noexcept { return trtx_engine_nb_bindings(eng()); }
    int32_t getBindingIndex(const char* name) const noexcept { return trtx_engine_binding_index(eng(), name); }
    const char* getBindingName(int32_t i) const noexcept { return trtx_engine_binding_name(eng(), i); }
    bool bindingIsInput(int32_t i) const noexcept { return trtx_engine_binding_is_input(eng(), i) != 0; }
    Dims getBindingDimensions(int32_t i) const noexcept {
        trtx_dims d{};
        trtx_engine_binding_dims(eng(), i, &d);
        return shim::from_c(d);
    }
    DataType getBindingDataType(int32_t) const noexcept { return DataType::kFLOAT; }
    int32_t getMaxBatchSize() const noexcept { return trtx_engine_max_batch(eng()); }
    int32_t getNbIOTensors() const noexcept { return getNbBindings(); }
    const char* getIOTensorName(int32_t i) const noexcept { return getBindingName(i); }
    DataType getTensorDataType(const char*) const noexcept { return DataType::kFLOAT; }
    Dims getTensorShape(const char* name) const noexcept { return getBindingDimensions(getBindingIndex(name)); }
    TensorIOMode getTensorIOMode(const char* name) const noexcept {
        const int32_t i = getBindingIndex(name);
        return i < 0 ? TensorIOMode::kNONE : (bindingIsInput(i) ? TensorIOMode::kINPUT : TensorIOMode::kOUTPUT);
    }
    size_t getDeviceMemorySize() const noexcept { return trtx_engine_device_memory(eng()); }
    IHostMemory* serialize() const noexcept {
        trtx_hostmem* m = nullptr;
        if (!mE && !mPlan.empty()) return trtx_hostmem_create(mPlan.data(), mPlan.size(), &m) == TRTX_OK ? new IHostMemory(m) : nullptr;
        return trtx_engine_serialize(mE, &m) == TRTX_OK ? new IHostMemory(m) : nullptr;
    }
    trtx_engine* handle() const noexcept { return eng(); }

   private:
    trtx_engine* eng() const noexcept {
        if (!mE && !mPlan.empty() && trtx_engine_deserialize(mPlan.data(), mPlan.size(), &mE) != TRTX_OK) mE = nullptr;
        return mE;
    }
    mutable trtx_engine* mE;
    std::vector<uint8_t> mPlan;
};

class IRuntime {
   public:
    virtual ~IRuntime() = default;
    void destroy() noexcept { delete this; }
    ICudaEngine* deserializeCudaEngine(const void* blob, size_t size, void* /*pluginFactory*/ = nullptr) noexcept {
        if (trtx_device_count() < 1) return new ICudaEngine(blob, size);  // build-only machine: the engine keeps its plan (serialize() works,
                                                                          // createExecutionContext fails: there is no CPU fallback)
        trtx_engine* e = nullptr;
        if (trtx_engine_deserialize(blob, size, &e) != TRTX_OK) return nullptr;
        return new ICudaEngine(e);
    }
};

class IBuilderConfig {
   public:
    explicit IBuilderConfig(trtx_builder* b) : mB(b) {}
    virtual ~IBuilderConfig() = default;
    void destroy() noexcept { delete this; }
    void setMaxWorkspaceSize(size_t bytes) noexcept { trtx_builder_set_workspace(mB, bytes); }
    void setMemoryPoolLimit(MemoryPoolType, size_t bytes) noexcept { trtx_builder_set_workspace(mB, bytes); }
    void setFlag(BuilderFlag f) noexcept { trtx_builder_set_flag(mB, (int32_t)f, 1); }
    void setMaxAuxStreams(int32_t n) noexcept { trtx_builder_set_max_aux_streams(mB, n); }
    void clearFlag(BuilderFlag f) noexcept { trtx_builder_set_flag(mB, (int32_t)f, 0); }
    void setInt8Calibrator(IInt8Calibrator* c) noexcept {
        trtx_calibrator_vtbl v{};
        if (c) {
            v.self = c;
            v.get_batch_size = [](void* s) -> int32_t { return static_cast<IInt8Calibrator*>(s)->getBatchSize(); };
            v.get_batch = [](void* s, void** bindings, const char* const* names, int32_t nb) -> int32_t {
                return static_cast<IInt8Calibrator*>(s)->getBatch(bindings, const_cast<const char**>(names), nb) ? 1 : 0;
            };
            v.read_cache = [](void* s, size_t* len) -> const void* { return static_cast<IInt8Calibrator*>(s)->readCalibrationCache(*len); };
            v.write_cache = [](void* s, const void* p, size_t len) { static_cast<IInt8Calibrator*>(s)->writeCalibrationCache(p, len); };
            v.get_algorithm = [](void* s) -> int32_t { return (int32_t) static_cast<IInt8Calibrator*>(s)->getAlgorithm(); };
        }
        trtx_builder_set_int8_calibrator(mB, c ? &v : nullptr);
    }

   private:
    trtx_builder* mB;
};

class IBuilder {
   public:
    explicit IBuilder(trtx_builder* b) : mB(b) {}
    virtual ~IBuilder() { trtx_builder_destroy(mB); }
    void destroy() noexcept { delete this; }
    IBuilderConfig* createBuilderConfig() noexcept { return new IBuilderConfig(mB); }
    INetworkDefinition* createNetworkV2(uint32_t flags) noexcept {
        trtx_network* n = nullptr;
        if (trtx_network_create(mB, flags, &n) != TRTX_OK) return nullptr;
        return new INetworkDefinition(n);
    }
    INetworkDefinition* createNetwork() noexcept { return createNetworkV2(0U); }
    void setMaxBatchSize(int32_t n) noexcept { trtx_builder_set_max_batch(mB, n); }
    bool platformHasFastFp16() const noexcept { return true; }
    bool platformHasFastInt8() const noexcept { return true; }  // v_mfma_i32_16x16x64_i8
    IHostMemory* buildSerializedNetwork(INetworkDefinition& net, IBuilderConfig&) noexcept {
        trtx_hostmem* m = nullptr;
        if (trtx_build_serialized(mB, net.handle(), &m) != TRTX_OK) return nullptr;
        return new IHostMemory(m);
    }
    ICudaEngine* buildEngineWithConfig(INetworkDefinition& net, IBuilderConfig& cfg) noexcept {
        std::unique_ptr<IHostMemory> m(buildSerializedNetwork(net, cfg));
        if (!m) return nullptr;
        if (trtx_device_count() < 1) return new ICudaEngine(m->data(), m->size());  // build-only machine: the engine keeps its plan
        IRuntime rt;
        return rt.deserializeCudaEngine(m->data(), m->size());
    }

   private:
    trtx_builder* mB;
};

inline IBuilder* createInferBuilder(ILogger&) noexcept {
    trtx_builder* b = nullptr;
    return trtx_builder_create(&b) == TRTX_OK ? new IBuilder(b) : nullptr;
}
inline IRuntime* createInferRuntime(ILogger&) noexcept { return new IRuntime(); }

}  // namespace nvinfer1

#endif  // TRTX_NVINFER_SHIM_H_
