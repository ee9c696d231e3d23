// Every environment switch of libtrtx_hip.so, in ONE place (round 5; VERDICT r4 item 7: rounds 2-4 had ~45 getenv calls inside launch functions and
// lowering passes - a lab notebook, not a product).  What is left is the documented set below (README "Environment"); the experiment switches of earlier
// rounds are gone with the experiments (tools/hip/experiments/README.md).
#pragma once
#include <string>

namespace trtx {

struct Options {
    // --- documented knobs
    int tune = -1;               // TRTX_TUNE: unset = time the tactics when a plan is BUILT on a machine with a GPU; 0 = never (static defaults); 1 = also time at
                                 //            deserialize a plan that carries no tactics
                                 //            2 = as unset, without the third look of engines built for contexts in flight (tune.cpp)
    bool tune_verbose = false;   // TRTX_TUNE_VERBOSE: per-layer tactic report on stderr
    int tune_margin = -1;        // TRTX_TUNE_MARGIN=<percent>: engines built for several contexts in flight (setMaxAuxStreams(0)): a layer leaves its default kernel only
                                 //            for a candidate at least this much faster when timed alone (unset: 3, as every engine; round 6, tune.cpp)
    std::string tactic_cache;    // TRTX_TACTIC_CACHE=<file>: tactic choices outlive the process (ITimingCache analogue)
    bool graph = false;          // TRTX_GRAPH=1: hipGraph replay of the lane schedule (opt-in: it does not pay)
    std::string calib_report;    // TRTX_CALIB_REPORT=<file>: per-tensor INT8 calibration report
    double int8_clip_limit = 1e-4;   // TRTX_INT8_CLIP_LIMIT=<share>: entropy calibration may not clip more than this share of a tensor's elements (the threshold is raised
                                 //            to that quantile; 0 = the entropy threshold as found).  Round 6, int8.cpp.
    int lanes = 0;               // TRTX_LANES=<n>: streams per execution context (0: 1 + IBuilderConfig::setMaxAuxStreams)
    // --- A/B switches: each turns ONE lowering pass or kernel family off so that the test-suite can hold the two forms against each other bit for bit
    bool group_convs = true;     // TRTX_GROUP_CONVS=0: sibling convolutions one launch each (same K order, same bits)
    bool fold_upsample = true;   // TRTX_FOLD_UPSAMPLE=0: Upsample -> Concat -> Conv1x1 keeps its resize launch
    bool ws = true;              // TRTX_CONV_NOWS=1: no weight-stationary kernel
    bool wsk = true;             // TRTX_CONV_NOWSK=1: no wave-split-K kernel
    bool gemm256 = true;         // TRTX_GEMM256=0: no 256 x 256 x 64 tile among the candidates
    bool patch = true;           // TRTX_CONV_PATCH=0: no resident-patch 3x3 kernel among the candidates / in the grouped launches
    int res = 7;                 // TRTX_CONV_RES=<mask>: resident-operand kernels (conv_res.hip) among the candidates / in the grouped launches: 1 = the 3x3 kernel, 2 = the 1x1 kernel, 4 = the 3x3 kernel in grouped launches too; 0 = none
    bool roles = true;           // TRTX_CONV_ROLES=0: no fetching / multiplying wave-role variants among the candidates (fp32 plans)
    bool f32_mfma = true;        // TRTX_F32_DIRECT=1: fp32 engines on the scalar direct kernel of rounds 1-4 (no fp32 MFMA, no fp32 stem kernel)
    bool roialign_fused = true;  // TRTX_ROIALIGN_PLUGIN=1: RoIAlign stays a plugin op (fp32 NCHW edge)
    bool roialign_fold_stride = true;   // TRTX_ROIALIGN_FOLD_STRIDE=0: RoIAlign emits all 14 x 14 bins
    bool profile_kernel_events = true;  // TRTX_PROFILE_NO_KERNEL_EVENTS=1: trtx_context_profile without per-launch start / stop events
    int op_reps = 1;             // TRTX_OP_REPS=<n>: timing tools, launches per single-op C-ABI call
    int conv_dbg = 0;            // TRTX_CONV_DBG=<mask>: ablation builds (-DTRTX_CONV_ABLATE) only; the product kernels ignore it
    int f32_stages = 0;          // TRTX_F32_NST: ablation builds only
};

// the switches as the environment has them NOW (lowering and engine creation: tests flip them inside one process)
Options read_options();
// the switches as they were when the library first asked (launch paths: read once)
const Options& options();

}  // namespace trtx
