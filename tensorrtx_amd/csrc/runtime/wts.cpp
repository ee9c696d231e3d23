// .wts weight-file loader: the runtime-side twin of the reference's loadWeights
// (lenet/utils.h:49-80, yolov8/src/block.cpp:13-43).  Format (tutorials/getting_started.md:107-132):
//   line 1: <count>;  then per blob: <name> <n> <hex32> x n, each hex token the big-endian IEEE-754 bit
//   pattern of one fp32.  Whitespace-agnostic (gen_wts.py dialects use one or two spaces).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../common.h"

struct trtx_wts {
    std::vector<std::string> names;
    std::vector<std::vector<float>> blobs;
    std::map<std::string, int> index;
};

namespace {
// fast whitespace-delimited tokenizer over the whole file
struct Tok {
    const char* p;
    const char* end;
    bool next(const char** s, size_t* n) {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
        if (p >= end) return false;
        *s = p;
        while (p < end && !(*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
        *n = (size_t)(p - *s);
        return true;
    }
};
inline int hexval(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}
}  // namespace

extern "C" int32_t trtx_wts_load(const char* path, trtx_wts** out) {
    if (!path || !out) return TRTX_ERR_INVALID;
    FILE* f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "[trtx_hip] unable to open weight file %s\n", path);
        return TRTX_ERR_IO;
    }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)sz);
    if (sz > 0 && fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) {
        fclose(f);
        return TRTX_ERR_IO;
    }
    fclose(f);
    Tok t{buf.data(), buf.data() + buf.size()};
    const char* s;
    size_t n;
    if (!t.next(&s, &n)) return TRTX_ERR_IO;
    const long count = strtol(std::string(s, n).c_str(), nullptr, 10);
    if (count <= 0) return TRTX_ERR_IO;  // "Invalid weight map file." (block.cpp:24)
    auto* w = new trtx_wts();
    for (long i = 0; i < count; ++i) {
        if (!t.next(&s, &n)) {
            delete w;
            return TRTX_ERR_IO;
        }
        std::string name(s, n);
        if (!t.next(&s, &n)) {
            delete w;
            return TRTX_ERR_IO;
        }
        const long size = strtol(std::string(s, n).c_str(), nullptr, 10);
        // every value takes at least two characters ("0 "): a count the rest of the file cannot hold is a corrupt header, not an
        // allocation request
        if (size < 0 || (size_t)size > (size_t)(buf.data() + buf.size() - s) / 2 + 1) {
            delete w;
            return TRTX_ERR_IO;
        }
        std::vector<float> v((size_t)size);
        for (long k = 0; k < size; ++k) {
            if (!t.next(&s, &n) || n > 8) {
                delete w;
                return TRTX_ERR_IO;
            }
            uint32_t bits = 0;
            for (size_t c = 0; c < n; ++c) {
                const int h = hexval(s[c]);
                if (h < 0) {
                    delete w;
                    return TRTX_ERR_IO;
                }
                bits = (bits << 4) | (uint32_t)h;
            }
            memcpy(&v[(size_t)k], &bits, 4);
        }
        w->index[name] = (int)w->names.size();
        w->names.push_back(std::move(name));
        w->blobs.push_back(std::move(v));
    }
    *out = w;
    return TRTX_OK;
}

extern "C" int32_t trtx_wts_count(const trtx_wts* w) { return w ? (int32_t)w->names.size() : 0; }

extern "C" int32_t trtx_wts_entry(const trtx_wts* w, int32_t i, const char** name, const float** values,
                                  int64_t* count) {
    if (!w || i < 0 || i >= (int32_t)w->names.size()) return TRTX_ERR_INVALID;
    if (name) *name = w->names[i].c_str();
    if (values) *values = w->blobs[i].data();
    if (count) *count = (int64_t)w->blobs[i].size();
    return TRTX_OK;
}

extern "C" int32_t trtx_wts_find(const trtx_wts* w, const char* name, const float** values, int64_t* count) {
    if (!w || !name) return TRTX_ERR_INVALID;
    auto it = w->index.find(name);
    if (it == w->index.end()) return TRTX_ERR_INVALID;
    return trtx_wts_entry(w, it->second, nullptr, values, count);
}

extern "C" void trtx_wts_free(trtx_wts* w) { delete w; }
