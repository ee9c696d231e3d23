// Engine (device-resident packed weights + lowered plan) and execution context (activation arena +
// stream executor).  IRuntime::deserializeCudaEngine / ICudaEngine / IExecutionContext of the reference
// (yolov8/yolov8_det.cpp:42-66,98; lenet/lenet.cpp:187-243).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "plan.h"

struct trtx_engine {
    std::unique_ptr<trtx::Network> net;
    trtx::Plan plan;
    std::vector<uint8_t> blob;  // the serialized plan this engine was created from
    void* d_weights = nullptr;
    int device = -1;  // HIP device the weights live on (current device at deserialize); contexts and enqueues must run there
    int plugins_initialized = 0;  // number of plugin ops (in plan order) whose initialize() succeeded
    // what the tactic timing at deserialize decided, one record per MFMA convolution (runtime/tune.cpp; trtx_engine_tactics)
    struct TacticRecord {
        int op = -1;
        std::string chosen, dflt;
        float chosen_us = -1.f, default_us = -1.f;  // in-place timings of the chosen / the default tactic (-1: taken from the process cache)
        int candidates = 0;
    };
    std::vector<TacticRecord> tactics;
    ~trtx_engine();
};

namespace trtx {
struct CalibObserver;
}

struct trtx_context {
    trtx_engine* engine = nullptr;
    void* d_arena = nullptr;
    std::vector<void*> addr;  // setTensorAddress slots, one per binding
    // concurrency (plan.num_lanes > 1): lane 0 is the caller's stream, lanes 1.. are streams owned by the context;
    // op_event[k] is recorded after op k when another lane waits for it
    std::vector<hipStream_t> lane_stream;
    std::vector<hipEvent_t> op_event, lane_done;
    hipEvent_t start_event = nullptr;
    // hipGraph replay of the lane schedule (one graph per (batch, binding pointers) seen; see trtx_context_enqueue)
    struct CapturedGraph {
        int batch = 0;
        std::vector<void*> bindings;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        uint64_t last_use = 0;
    };
    std::vector<CapturedGraph> graphs;
    std::vector<CapturedGraph> seen;   // combinations enqueued once (eagerly) so far
    uint64_t enqueue_count = 0;
    struct trtx::CalibObserver* observer = nullptr;  // INT8 calibration run: statistics of every NHWC tensor written (int8.h)
    // trtx_context_enqueue_frames: for the duration of that call, the camera frames the stem samples instead of reading its fp32 input
    const trtx::StemFrame* frames = nullptr;
    bool tuning = false;   // tactic-timing runs: plugins, the fused detect head and RoIAlign are skipped (runtime/tune.cpp)
    int graph_state = 0;   // 0 undecided, 1 eligible, -1 never (user plugins, capture failed once, disabled)
    ~trtx_context();
};

namespace trtx {
// runs every op of the plan on `stream`; with `prof` != nullptr brackets each op with hipEvents
struct OpTiming {
    std::string name, kind;
    float ms;                // between the stream events before and after the op (includes the hand-over between two launches)
    float kernel_ms = -1.f;  // convolutions: the dispatch's own begin -> end (LaunchProbe, kernels.h); -1 where not measured
};
int32_t execute_plan(trtx_context* c, int batch, void* const* bindings, hipStream_t stream,
                     std::vector<OpTiming>* prof);
// runtime/tune.cpp.  time_now = false: apply the kernel tactics the plan carries (deserializeCudaEngine).  time_now = true: time the
// exchangeable launch configurations of every MFMA convolution in place, keep the fastest and record the choices in
// e->net->tactics (buildSerializedNetwork on a machine with a GPU).
int32_t tune_engine(trtx_engine* e, bool time_now);
// deserializeCudaEngine with the choice of the above (trtx_engine_deserialize = engine_from_plan(..., false) unless TRTX_TUNE=1 and the
// plan was built without a GPU)
int32_t engine_from_plan(const void* plan_data, size_t size, bool time_tactics, trtx_engine** out);
}  // namespace trtx
