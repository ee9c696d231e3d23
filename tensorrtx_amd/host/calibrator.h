// INT8 calibrators for the host builders, written against the nvinfer1:: shim as the reference's are.
//   Int8EntropyCalibrator2 — the reference's class (yolov8/include/calibrator.h:14-36, yolov8/src/calibrator.cpp:9-74): same
//       constructor arguments, batches read from an image directory, letterboxed to the network input, cache file read /
//       written verbatim.  OpenCV is absent here: images are binary PPM (P6) files, pre-processed on the GPU by the runtime's
//       own letterbox entry point (trtx_batch_preprocess, the f2 row) instead of the reference's cv::resize + blobFromImages.
//   CallbackCalibrator — forwards to a C v-table (trtx_host_set_calibrator): lets a test or a Python tool feed device batches.
#pragma once
#include <dirent.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <algorithm>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "NvInfer.h"

namespace trtx_host {

class Int8EntropyCalibrator2 : public nvinfer1::IInt8EntropyCalibrator2 {
   public:
    Int8EntropyCalibrator2(int batchsize, int input_w, int input_h, const char* img_dir, const char* calib_table_name,
                           const char* input_blob_name, bool read_cache = true)
        : batchsize_(batchsize), input_w_(input_w), input_h_(input_h), img_dir_(img_dir), calib_table_name_(calib_table_name),
          input_blob_name_(input_blob_name), read_cache_(read_cache) {
        input_count_ = (size_t)3 * input_w * input_h * batchsize;
        if (DIR* d = opendir(img_dir)) {
            while (dirent* e = readdir(d)) {
                const std::string n = e->d_name;
                if (n.size() > 4 && n.substr(n.size() - 4) == ".ppm") img_files_.push_back(n);
            }
            closedir(d);
        }
        std::sort(img_files_.begin(), img_files_.end());
    }
    ~Int8EntropyCalibrator2() override {
        if (owns_preprocess_) trtx_preprocess_destroy();   // never the application's own pipeline
        if (device_input_) (void)hipFree(device_input_);
    }
    int32_t getBatchSize() const noexcept override { return batchsize_; }
    bool getBatch(void* bindings[], const char* names[], int32_t nbBindings) noexcept override {
        if (img_idx_ + batchsize_ > (int)img_files_.size()) return false;
        if (!device_ready()) return false;
        std::vector<std::vector<unsigned char>> pix(batchsize_);
        std::vector<const void*> src(batchsize_);
        std::vector<int> ws(batchsize_), hs(batchsize_);
        for (int i = 0; i < batchsize_; ++i) {
            if (!read_ppm(std::string(img_dir_) + "/" + img_files_[img_idx_ + i], &pix[i], &ws[i], &hs[i])) return false;
            src[i] = pix[i].data();
        }
        img_idx_ += batchsize_;
        if (const int32_t st = trtx_batch_preprocess(src.data(), ws.data(), hs.data(), batchsize_, static_cast<float*>(device_input_), input_w_, input_h_, nullptr)) {
            fprintf(stderr, "[trtx_host] calibrator: trtx_batch_preprocess failed (%s)%s\n", trtx_status_string(st),
                    owns_preprocess_ ? "" : " - the application's own preprocess pipeline is in use; its staging ring may be smaller than the calibration batch");
            return false;
        }
        (void)hipDeviceSynchronize();
        for (int b = 0; b < nbBindings; ++b)
            if (input_blob_name_ == names[b]) bindings[b] = device_input_;
        return true;
    }
    const void* readCalibrationCache(size_t& length) noexcept override {
        calib_cache_.clear();
        std::ifstream input(calib_table_name_, std::ios::binary);
        input >> std::noskipws;
        if (read_cache_ && input.good()) std::copy(std::istream_iterator<char>(input), std::istream_iterator<char>(), std::back_inserter(calib_cache_));
        length = calib_cache_.size();
        return length ? calib_cache_.data() : nullptr;
    }
    void writeCalibrationCache(const void* cache, size_t length) noexcept override {
        std::ofstream output(calib_table_name_, std::ios::binary);
        output.write(reinterpret_cast<const char*>(cache), (std::streamsize)length);
    }

   private:
    // Device buffer and the letterbox pipeline, on the first batch (a build that finds a calibration cache never gets here and needs
    // no GPU).  trtx_preprocess_init is process-wide state: if the application has already initialised it (TRTX_ERR_STATE) its
    // pipeline is used and left alone; only a pipeline this object created is destroyed with it.
    bool device_ready() {
        if (device_input_) return true;
        if (hipMalloc(&device_input_, input_count_ * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            device_input_ = nullptr;
            fprintf(stderr, "[trtx_host] calibrator: no device memory for a %d x 3 x %d x %d input batch\n", batchsize_, input_h_, input_w_);
            return false;
        }
        const int32_t st = trtx_preprocess_init(4096 * 3112, std::max(2, batchsize_));  // kMaxInputImageSize of the reference's config.h
        owns_preprocess_ = st == TRTX_OK;
        if (st != TRTX_OK && st != TRTX_ERR_STATE) {
            fprintf(stderr, "[trtx_host] calibrator: trtx_preprocess_init failed (%s)\n", trtx_status_string(st));
            return false;
        }
        return true;
    }
    // binary PPM: "P6 <w> <h> 255\n" + RGB bytes; returned as BGR rows (what cv::imread yields and the letterbox expects)
    static bool read_ppm(const std::string& path, std::vector<unsigned char>* bgr, int* w, int* h) {
        std::ifstream f(path, std::ios::binary);
        std::string magic;
        int maxv = 0;
        f >> magic >> *w >> *h >> maxv;
        if (!f.good() || magic != "P6" || maxv != 255 || *w < 1 || *h < 1) return false;
        f.get();
        bgr->resize((size_t)*w * *h * 3);
        f.read(reinterpret_cast<char*>(bgr->data()), (std::streamsize)bgr->size());
        for (size_t i = 0; i + 2 < bgr->size(); i += 3) std::swap((*bgr)[i], (*bgr)[i + 2]);
        return f.good();
    }
    int batchsize_, input_w_, input_h_, img_idx_ = 0;
    std::string img_dir_;
    std::vector<std::string> img_files_;
    size_t input_count_ = 0;
    std::string calib_table_name_, input_blob_name_;
    bool read_cache_;
    void* device_input_ = nullptr;
    bool owns_preprocess_ = false;
    std::vector<char> calib_cache_;
};

class CallbackCalibrator : public nvinfer1::IInt8Calibrator {
   public:
    explicit CallbackCalibrator(const trtx_calibrator_vtbl& v) : v_(v) {}
    nvinfer1::CalibrationAlgoType getAlgorithm() override {
        return v_.get_algorithm ? static_cast<nvinfer1::CalibrationAlgoType>(v_.get_algorithm(v_.self)) : nvinfer1::CalibrationAlgoType::kENTROPY_CALIBRATION_2;
    }
    int32_t getBatchSize() const noexcept override { return v_.get_batch_size ? v_.get_batch_size(v_.self) : 1; }
    bool getBatch(void* bindings[], const char* names[], int32_t nb) noexcept override {
        return v_.get_batch && v_.get_batch(v_.self, bindings, names, nb) != 0;
    }
    const void* readCalibrationCache(size_t& length) noexcept override {
        length = 0;
        return v_.read_cache ? v_.read_cache(v_.self, &length) : nullptr;
    }
    void writeCalibrationCache(const void* p, size_t n) noexcept override {
        if (v_.write_cache) v_.write_cache(v_.self, p, n);
    }

   private:
    trtx_calibrator_vtbl v_;
};

}  // namespace trtx_host
