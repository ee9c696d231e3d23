// INT8 calibration (BuilderFlag::kINT8 + IInt8EntropyCalibrator2; reference: yolov8/src/calibrator.cpp:9-74,
// yolov8/src/model.cpp:317-324, retinaface/retina_r50.cpp:219-225).  TensorRT owns the statistics collection, the entropy
// threshold search and the cache format in the reference; this file is that part of the builder.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "graph.h"

struct trtx_context;

namespace trtx {

constexpr int kCalibBins = 8192;    // bins COLLECTED per tensor: the range is only known to a power of two while batches arrive (it doubles, bins
                                    // fold pairwise), so the largest |x| ends up anywhere in the upper half of it - at least 4096 bins cover it
constexpr int kEntropyBins = 2048;  // bins SEARCHED: the collected histogram resampled onto [0, largest |x| seen] (round 4: searching 2048 bins of
                                    // a range up to twice the data's put the threshold on a 2x coarser grid than the calibrator is described with)

// statistics collected while a calibration batch runs through the fp16 plan (one slot per plan storage = per owning tensor)
struct CalibObserver {
    int mode = 0;                       // 1: |x| maxima, 2: histograms
    unsigned* d_max = nullptr;          // [storages] float bits
    unsigned long long* d_hist = nullptr;  // [storages][kCalibBins]
    std::vector<float> range;           // histogram range per storage (0 = not started)
};

// symmetric threshold that minimises the KL divergence between the 2048-bin |x| histogram and its 128-level quantisation
// (the "entropy calibration" of NVIDIA's 8-bit-inference material; TensorRT's kENTROPY_CALIBRATION_2).  Returns the threshold.
float entropy_threshold(const std::vector<double>& hist, float range);
// The clip limit on top of it (round 6): the smallest bin edge T >= thr of the |x| histogram over [0, range] beyond which at most `limit` of the mass lies
// (a bin counts as beyond T when its centre is); limit <= 0 returns thr.
float clip_limited_threshold(const std::vector<double>& hist, float range, float thr, double limit);

// "TRT-<ver>-EntropyCalibration2\n<tensor name>: <hex of the float scale bits>\n..." (the format TensorRT writes and the
// reference's calibrator stores / reloads verbatim, calibrator.cpp:56-74)
std::string write_calibration_cache(const Network& net);
bool read_calibration_cache(const void* data, size_t length, Network* net, std::string* err);
// name used for a tensor in the cache (unnamed intermediate tensors get a stable synthetic name)
std::string calib_tensor_name(const Network& net, int tensor);

// runs the calibrator's batches through an fp16 engine of `net` (needs a GPU), fills net->tensor_scale
int32_t run_int8_calibration(Network* net, const trtx_calibrator_vtbl& calib);

}  // namespace trtx
