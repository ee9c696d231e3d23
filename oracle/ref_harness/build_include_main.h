// ORACLE — test infrastructure only.  Shared tail of the builder harnesses that wrap a reference sample program whose network builder
// lives in the same file as its main() (resnet/resnet50.cpp, retinaface/retina_r50.cpp, rcnn/rcnn.cpp): the reference source is compiled
// UNMODIFIED (included with `main` renamed), run_dir is made current so that the program's own relative weight path ("../resnet50.wts")
// resolves, and the serialized engine its own build function returns is copied out.
#pragma once
#include <unistd.h>

#include <cstdlib>
#include <cstring>

#define REF_EXPORT extern "C" __attribute__((visibility("default")))

static int ref_copy_out(nvinfer1::IHostMemory* m, void** out, size_t* len) {
    if (!m) return 1;
    *len = m->size();
    *out = malloc(m->size());
    memcpy(*out, m->data(), m->size());
    m->destroy();
    return 0;
}
REF_EXPORT void ref_build_free(void* p) { free(p); }
