"""ORACLE — test infrastructure only.  Builds `oracle/_ref/`: the REFERENCE'S OWN code, compiled from the sources
where they lie under /root/reference (nothing is copied into git; `oracle/_ref/` is git-ignored but travels to the GPU
box like any other built `.so`).

Two kinds of artefact:

* ``libref_host.so`` (g++, CPU): the reference's host-side post-processing — yolov8 ``iou/cmp/nms/batch_nms``
  (yolov8/src/postprocess.cpp:71-129), yolov5 ``iou/cmp/nms`` (yolov5/src/postprocess.cpp:30-73), RetinaFace
  ``iou/cmp/nms`` (retinaface/common.hpp:91-130).  Those functions sit in translation units that also need OpenCV and
  TensorRT, so the recipe cuts the function bodies out by their signatures (regions below), writes them to
  ``oracle/_ref/gen/*.inc`` and compiles them inside the wrapper ``oracle/ref_harness/host_post.cpp``.

* ``libref_<family>.so`` (hipcc, gfx950): the reference's CUDA plugin sources, UNMODIFIED except for the lexical
  patch listed in ``PATCHES`` (chevron spacing), compiled as *user plugins* against this repo's
  ``include/NvInfer.h`` with the spelling bridge in ``oracle/ref_compat/`` (cudaStream_t -> hipStream_t, cub -> hipcub,
  thrust::cuda -> thrust::hip).  Their ``REGISTER_TENSORRT_PLUGIN`` statics register them with libtrtx_hip.so's plugin
  registry when the library is dlopened, so the tests drive the reference's own kernels through the same C-ABI plugin
  v-table (trampolines in NvInfer.h) the engine uses — on the MI355X — and compare them with the product kernels.
  yolov8/src/postprocess.cu and preprocess.cu (plain functions, no plugin) are built the same way.

The build needs /root/reference and therefore only runs in the build container (``__graft_entry__.build()`` calls it
when the directory exists); on the GPU box the prebuilt files are used.  Run directly: ``python oracle/ref_build.py``.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("TRTX_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")
GEN = os.path.join(OUT, "gen")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# (output .inc, reference file, first line regex, regex of the first line NOT taken)
REGIONS = [
    ("yolov8_nms.inc", "yolov8/src/postprocess.cpp", r"^static float iou\(", r"^void process_decode_ptr_host\("),
    ("yolov8_nms_obb.inc", "yolov8/src/postprocess.cpp", r"^std::tuple<float, float, float> convariance_matrix\(", r"^static std::vector<cv::Point> get_corner\("),
    ("yolov5_nms.inc", "yolov5/src/postprocess.cpp", r"^static float iou\(", r"^void draw_bbox\("),
    ("retina_nms.inc", "retinaface/common.hpp", r"^static float iou\(", r"^// Load weights from files"),
    ("retina_types.inc", "retinaface/decode.h", r"^namespace decodeplugin", r"^namespace nvinfer1"),
]

# Lexical patches applied to the generated copy of a reference source (never to /root/reference).  Semantics-neutral.
PATCHES = {
    # clang-format split the kernel-launch chevrons (`<< <` ... `>> >`); nvcc tolerates that, hipcc does not
    "*": [(r"<<\s+<", "<<<"), (r">>\s+>", ">>>")],
    # the file's own, documented precision switch ("set USE_INT8 or USE_FP16 or USE_FP32"): the INT8 default needs calibration images
    "retinaface/retina_r50.cpp": [(r"#define USE_INT8", "#define USE_FP16")],
    # yolov4's plugin is written against TensorRT 7's IPluginV2::enqueue (`void** outputs`); TensorRT 8 - the API include/NvInfer.h
    # spells, and what the yolov8 / retinaface plugins reach through TRT_CONST_ENQUEUE - made that `void* const*`
    "yolov4/mish.h": [(r"void\*\* outputs", "void* const* outputs")],
    "yolov4/mish.cu": [(r"void\*\* outputs", "void* const* outputs")],
}

# family -> (sources, extra include dirs relative to the reference, headers that need a patched copy)
FAMILIES = {
    "yolov8_plugin": (["yolov8/plugin/yololayer.cu"], ["yolov8/plugin", "yolov8/include"], []),
    "yolov8_post": (["yolov8/src/postprocess.cu", "yolov8/src/preprocess.cu"], ["yolov8/include"], []),
    "yolov5_plugin": (["yolov5/plugin/yololayer.cu"], ["yolov5/plugin", "yolov5/src"], []),
    "retinaface_plugin": (["retinaface/decode.cu"], ["retinaface"], []),
    "yolov4_plugin": (["yolov4/mish.cu"], ["yolov4"], ["yolov4/mish.h"]),   # Mish_TRT (round 4)
    "rcnn_plugins": (["rcnn/RpnDecode.cu", "rcnn/RpnNms.cu", "rcnn/RoiAlign.cu", "rcnn/PredictorDecode.cu", "rcnn/BatchedNms.cu",
                      "rcnn/MaskRcnnInference.cu"], ["rcnn"], []),
}


# Reference network BUILDERS (plain consumers of the nvinfer1 API), compiled unmodified against include/NvInfer.h and linked with
# libtrtx_hip.so: name -> (sources, include dirs, harness under oracle/ref_harness/, extra -D flags).  tests/test_ref_builders.py requires
# the product's own host builders (tensorrtx_amd/host/*.cpp) to emit the same networks.
BUILDERS = {
    # yololayer.cu comes along because yololayer.h registers the plugin creator in every TU that includes it (block.cpp does)
    "yolov8": (["yolov8/src/block.cpp", "yolov8/src/model.cpp", "yolov8/plugin/yololayer.cu"], ["yolov8/include", "yolov8/plugin"], "build_yolov8.cpp", []),
    # sample programs whose builder shares a file with main(): the harness includes the (generated copy of the) .cpp with main renamed
    "lenet": (["lenet/lenet.cpp"], ["lenet"], "build_lenet.cpp", []),
    "resnet50": (["resnet/resnet50.cpp"], ["resnet"], "build_resnet50.cpp", []),
    "retinaface": (["retinaface/retina_r50.cpp", "retinaface/decode.cu"], ["retinaface"], "build_retinaface.cpp", []),
    # the reference's DEFAULT RetinaFace build (USE_INT8, its own calibrator reading a calibration table); no precision patch for this one
    "retinaface_int8": (["retinaface/retina_r50.cpp", "retinaface/decode.cu", "retinaface/calibrator.cpp"], ["retinaface"], "build_retinaface_int8.cpp",
                        ["REF_COMPAT_HOST_MALLOC_FALLBACK"]),
    "rcnn": (["rcnn/rcnn.cpp", "rcnn/RpnDecode.cu", "rcnn/RpnNms.cu", "rcnn/RoiAlign.cu", "rcnn/PredictorDecode.cu", "rcnn/BatchedNms.cu",
              "rcnn/MaskRcnnInference.cu"], ["rcnn"], "build_rcnn.cpp", []),
}
# sources that a harness #includes (they carry a main()): generated like the others, not compiled on their own
INCLUDED_BY_HARNESS = {"lenet/lenet.cpp", "resnet/resnet50.cpp", "retinaface/retina_r50.cpp", "rcnn/rcnn.cpp"}


def _patched(rel, only_generic=False):
    with open(os.path.join(REF, rel)) as f:
        text = f.read()
    for pat, rep in PATCHES["*"] + ([] if only_generic else PATCHES.get(rel, [])):
        text = re.sub(pat, rep, text)
    return text


def _write_if_changed(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if os.path.exists(path) and open(path).read() == text:
        return
    with open(path, "w") as f:
        f.write(text)


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def extract_regions():
    for out, rel, first, stop in REGIONS:
        lines = open(os.path.join(REF, rel)).read().split("\n")
        a = next(i for i, l in enumerate(lines) if re.search(first, l))
        b = next(i for i, l in enumerate(lines) if i > a and re.search(stop, l))
        _write_if_changed(os.path.join(GEN, out), "// generated from %s:%d-%d by oracle/ref_build.py (not tracked)\n" % (rel, a + 1, b)
                          + "\n".join(lines[a:b]) + "\n")


def build_host():
    extract_regions()
    so = os.path.join(OUT, "libref_host.so")
    src = os.path.join(HERE, "ref_harness", "host_post.cpp")
    deps = [src] + [os.path.join(GEN, r[0]) for r in REGIONS]
    if _newer(so, deps):
        # -ffp-contract=off: the reference is built by gcc/nvcc host passes without FMA contraction on x86-64
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + GEN, "-I" + REF, src, "-o", so])
    return so


def build_family(name):
    srcs, incs, patched_headers = FAMILIES[name]
    so = os.path.join(OUT, "libref_%s.so" % name)
    gdir = os.path.join(GEN, name)
    gen_srcs = []
    for rel in srcs + patched_headers:
        dst = os.path.join(gdir, os.path.basename(rel))
        _write_if_changed(dst, "// generated from %s by oracle/ref_build.py (lexical patches only; not tracked)\n" % rel + _patched(rel))
        if rel in srcs:
            gen_srcs.append(dst)
    harness = os.path.join(HERE, "ref_harness", name + ".cpp")
    extra = [harness] if os.path.exists(harness) else []
    deps = gen_srcs + extra + [os.path.join(ROOT, "include", "NvInfer.h"), os.path.join(HERE, "ref_compat", "ref_prelude.h")]
    if not _newer(so, deps):
        return so
    lib_dir = os.path.join(ROOT, "tensorrtx_amd", "lib")
    # -ffp-contract=off: nvcc's default (-fmad=true) fuses some multiply-adds, WHICH ones is the compiler's choice; the pin is on
    # the uncontracted IEEE arithmetic the oracle and the product plugins state (their Makefiles use the same flag)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-w", "-ffp-contract=off",
           "-include", os.path.join(HERE, "ref_compat", "ref_prelude.h"),
           "-I" + gdir,  # patched copies shadow the originals
           "-I" + os.path.join(HERE, "ref_compat"), "-I" + os.path.join(ROOT, "include")]
    cmd += ["-I" + os.path.join(REF, i) for i in incs]
    for s in gen_srcs + extra:
        cmd += ["-x", "hip", s]
    cmd += ["-L" + lib_dir, "-ltrtx_hip", "-Wl,-rpath,$ORIGIN/../../tensorrtx_amd/lib", "-o", so]
    subprocess.check_call(cmd)
    return so


def build_builder(name):
    srcs, incs, harness, defs = BUILDERS[name]
    so = os.path.join(OUT, "libref_build_%s.so" % name)
    gdir = os.path.join(GEN, "build_" + name)
    gen_srcs = []
    for rel in srcs:
        dst = os.path.join(gdir, os.path.basename(rel))
        # (the precision switch of retina_r50.cpp is NOT applied to the builder that pins the reference's default INT8 configuration)
        _write_if_changed(dst, "// generated from %s by oracle/ref_build.py (lexical patches only; not tracked)\n" % rel + _patched(rel, only_generic=name.endswith("_int8")))
        if rel not in INCLUDED_BY_HARNESS:
            gen_srcs.append(dst)
    hsrc = os.path.join(HERE, "ref_harness", harness)
    deps = [os.path.join(gdir, os.path.basename(r)) for r in srcs] + [hsrc, os.path.join(HERE, "ref_harness", "build_include_main.h"),
            os.path.join(ROOT, "include", "NvInfer.h"), os.path.join(HERE, "ref_compat", "ref_prelude.h"),
            os.path.join(HERE, "ref_compat", "opencv2", "opencv.hpp")]
    if not _newer(so, deps):
        return so
    lib_dir = os.path.join(ROOT, "tensorrtx_amd", "lib")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-w",
           "-include", os.path.join(HERE, "ref_compat", "ref_prelude.h"), "-I" + gdir, "-I" + os.path.join(HERE, "ref_compat"),
           "-I" + os.path.join(HERE, "ref_harness"), "-I" + os.path.join(ROOT, "include")] + ["-D" + d for d in defs]
    cmd += ["-I" + os.path.join(REF, i) for i in incs]
    for s_ in gen_srcs + [hsrc]:
        cmd += ["-x", "hip", s_]
    cmd += ["-L" + lib_dir, "-ltrtx_hip", "-Wl,-rpath,$ORIGIN/../../tensorrtx_amd/lib", "-o", so]
    subprocess.check_call(cmd)
    return so


def build_all(verbose=False):
    if not os.path.isdir(REF):
        if verbose:
            print("[ref_build] %s not present: using prebuilt oracle/_ref" % REF)
        return False
    os.makedirs(OUT, exist_ok=True)
    build_host()
    for name in FAMILIES:
        build_family(name)
    for name in BUILDERS:
        build_builder(name)
    return True


if __name__ == "__main__":
    ok = build_all(verbose=True)
    print("oracle/_ref:", sorted(f for f in os.listdir(OUT) if f.endswith(".so")) if os.path.isdir(OUT) else "absent")
    sys.exit(0 if ok or os.path.isdir(OUT) else 1)
