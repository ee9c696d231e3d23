"""The reference's own network builders, compiled UNMODIFIED against this repository's nvinfer1 shim (oracle/ref_build.py ->
oracle/_ref/libref_build_<family>.so), as the pin for the product's host builders (tensorrtx_amd/host/*.cpp).

Each case names: the reference entry point that is run, the seeded synthetic .wts it is run on, and the options of the product builder that
describe the same configuration.  The bar is byte equality of the serialized plans: same layers in the same creation order, same tensor
names, same parameters, same folded weights to the last bit.  CPU only (the shim serializes a network without a GPU).

Test infrastructure: nothing here is imported by the product.
"""
import ctypes
import hashlib
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile

from tests.util import synth_wts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_builder_plans.json")

# name -> (library, synthetic weights, product model + options)
CASES = {
    # lenet/lenet.cpp:164 APIToModel -> createLenetEngine :52
    "lenet": dict(lib="lenet", wts="lenet", model="lenet", opts=dict(batch=1)),
    # resnet/resnet50.cpp:231 APIToModel -> createEngine :154 (DataType::kFLOAT, no fp16 flag, 224 x 224)
    "resnet50": dict(lib="resnet50", wts="resnet50", model="resnet50", opts=dict(batch=1, fp16=0, h=224, w=224)),
    # retinaface/retina_r50.cpp:244 APIToModel -> createEngine :101 (USE_FP16; decodeplugin::INPUT_H x INPUT_W = 480 x 640)
    # The one documented divergence: the reference's Decode_TRT plugin serializes NOTHING (decode.cu:19-26, its input size is a compile-time
    # constant) while the product's carries the network size as two ints so that other sizes deserialize (builtin_plugins.cpp; an empty
    # blob still means 480 x 640).  `decode_state` = the state the product must have written; it is cut out before comparing.
    "retinaface_r50": dict(lib="retinaface", wts="retinaface_r50", model="retinaface_r50", opts=dict(batch=1, fp16=1, h=480, w=640),
                           decode_state=(480, 640)),
    # The reference's DEFAULT RetinaFace build: `#define USE_INT8` (retina_r50.cpp:12, :219-225) = BuilderFlag::kINT8 alone + ITS Int8EntropyCalibrator2
    # (retinaface/calibrator.cpp) reading "r50_int8calib.table"; the table (TensorRT's text format) is written by the test with one scale per
    # tensor of the network, so no image and no GPU is needed.  The product builds the same network with the same table through its calibrator.
    "retinaface_r50_int8": dict(lib="retinaface_int8", wts="retinaface_r50", model="retinaface_r50", opts=dict(batch=1, fp16=0, int8=1, h=480, w=640),
                                decode_state=(480, 640), calib_table="r50_int8calib.table"),
    # rcnn/rcnn.cpp:310 BuildRcnnModel -> createEngine_rcnn :250 after calculateSize :349 (480 x 640 image -> 800 x 1067 input), "fp16"
    "faster_rcnn_r50c4": dict(lib="rcnn", wts="rcnn_r50c4", model="rcnn_r50c4", opts=dict(batch=1, fp16=1, h=800, w=1067), mask=0),
    # the same with MASK_ON (rcnn.cpp:41, :277-296: mask head + MaskRcnnInference plugin, second output)
    "mask_rcnn_r50c4": dict(lib="rcnn", wts="rcnn_r50c4", model="rcnn_r50c4", opts=dict(batch=1, fp16=1, h=800, w=1067, mask=1), mask=1),
    # yolov8/src/model.cpp:98 / :1057 / :1310 / :2499 with the n-scale arguments of yolov8_det.cpp:57-62 (gd 0.33, gw 0.25, max 1024)
    "yolov8n_det": dict(lib="yolov8", wts="yolov8n", model="yolov8n", opts=dict(batch=1, fp16=1, h=640, w=640), task=0),
    "yolov8n_seg": dict(lib="yolov8", wts="yolov8n_seg", model="yolov8n", opts=dict(batch=1, fp16=1, h=640, w=640, task=1), task=1),
    "yolov8n_pose": dict(lib="yolov8", wts="yolov8n_pose", model="yolov8n", opts=dict(batch=1, fp16=1, h=640, w=640, task=2, classes=1), task=2),
    "yolov8n_obb": dict(lib="yolov8", wts="yolov8n_obb", model="yolov8n", opts=dict(batch=1, fp16=1, h=640, w=640, task=3, classes=15), task=3),
}


def lib_path(case):
    return os.path.join(REF_DIR, f"libref_build_{CASES[case]['lib']}.so")


def calibration_table(case: str) -> bytes:
    """A TensorRT-format calibration cache with one (seeded, distinct) scale per tensor of the case's network: names from the fp16 plan."""
    import numpy as np
    from tensorrtx_amd import engine

    c = CASES[case]
    wts, _ = synth_wts(c["wts"])
    opts = dict(c["opts"], fp16=1)
    opts.pop("int8", None)
    names = [t["name"] or f"(Unnamed Tensor* {t['id']})" for t in engine.describe_plan(engine.build_plan(c["model"], wts, **opts))["tensors"]]
    scales = np.random.default_rng(7).uniform(0.01, 0.2, len(names)).astype(np.float32)
    return b"TRT-8601-EntropyCalibration2\n" + b"".join(f"{n}: {struct.unpack('<I', struct.pack('<f', float(s)))[0]:08x}\n".encode() for n, s in zip(names, scales))


def _load(case):
    from tensorrtx_amd import capi

    capi.lib()  # libref_build_*.so NEEDs libtrtx_hip.so (the shim's C-ABI): map it first, by path
    return ctypes.CDLL(lib_path(case))


def reference_plan(case: str) -> bytes:
    """Run the reference's builder for `case` on the seeded synthetic weights; returns the plan it serialized.

    In a child process: the reference libraries register THEIR plugin creators (REGISTER_TENSORRT_PLUGIN at load time) in the process-wide
    registry and change the working directory; neither may leak into the process that builds and lowers the product's plans."""
    synth_wts(CASES[case]["wts"])  # create the cached weights here, once, not in the child
    if "calib_table" in CASES[case]:
        calibration_table(case)
    fd, path = tempfile.mkstemp(prefix="trtx_refplan_", suffix=".plan")
    os.close(fd)
    try:
        r = subprocess.run([sys.executable, "-m", "tests.ref_builder_cases", case, path], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, f"reference builder for {case} failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}"
        with open(path, "rb") as f:
            return f.read()
    finally:
        os.unlink(path)


def _reference_plan_here(case: str) -> bytes:
    c = CASES[case]
    wts, _ = synth_wts(c["wts"])
    L = _load(case)
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="trtx_refbuild_")
    try:
        if c["lib"] in ("lenet", "resnet50", "retinaface", "retinaface_int8"):
            # these programs open a fixed relative path from their working directory
            rel = {"lenet": "../models/lenet.wts", "resnet50": "../resnet50.wts", "retinaface": "../retinaface.wts", "retinaface_int8": "../retinaface.wts"}[c["lib"]]
            run = os.path.join(tmp, "build")
            os.makedirs(run)
            dst = os.path.normpath(os.path.join(run, rel))
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copy(wts, dst)
            if "calib_table" in c:   # the table the reference's calibrator will find in its working directory
                with open(os.path.join(run, c["calib_table"]), "wb") as f:
                    f.write(calibration_table(case))
            rc = getattr(L, f"ref_build_{c['lib']}")(run.encode(), 1, ctypes.byref(out), ctypes.byref(n))
        elif c["lib"] == "rcnn":
            rc = L.ref_build_rcnn(wts.encode(), 1, b"fp16", c["mask"], ctypes.byref(out), ctypes.byref(n))
        else:
            rc = L.ref_build_yolov8(c["task"], wts.encode(), ctypes.c_float(0.33), ctypes.c_float(0.25), 1024, 1, ctypes.byref(out),
                                    ctypes.byref(n))
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
    assert rc == 0, f"reference builder for {case} returned {rc}"
    data = ctypes.string_at(out, n.value)
    L.ref_build_free(out)
    return data


def product_plan(case: str) -> bytes:
    """The product's host builder on the same weights (with the documented plugin-state extension, if any, checked and removed)."""
    from tensorrtx_amd import engine

    c = CASES[case]
    wts, _ = synth_wts(c["wts"])
    if "calib_table" in c:
        from tensorrtx_amd import calibrator
        with calibrator.Calibrator(cache=calibration_table(case)).installed():
            plan = engine.build_plan(c["model"], wts, **c["opts"])
    else:
        plan = engine.build_plan(c["model"], wts, **c["opts"])
    if "decode_state" in c:
        marker = struct.pack("<i", 10) + b"Decode_TRT" + struct.pack("<i", 1) + b"1"
        at = plan.rfind(marker) + len(marker)
        assert at >= len(marker), "Decode_TRT plugin record not found"
        (n,) = struct.unpack_from("<Q", plan, at)
        assert n == 8 and at + 8 + n == len(plan), "the plugin record is expected to close the plan"
        assert struct.unpack_from("<ii", plan, at + 8) == c["decode_state"]
        plan = plan[:at] + struct.pack("<Q", 0)
    return plan


def wts_sha(case: str) -> str:
    wts, _ = synth_wts(CASES[case]["wts"])
    with open(wts, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def load_golden() -> dict:
    with open(GOLDEN) as f:
        return json.load(f)


if __name__ == "__main__":
    with open(sys.argv[2], "wb") as f:
        f.write(_reference_plan_here(sys.argv[1]))
