"""CPU tests of the host side: .wts loader, C++ builders through the nvinfer1 shim, network serialisation,
graph lowering (fusion / concat elimination / FLOP count) and the graph-interpreter oracle against the
PyTorch restatements.  No compute kernel runs here (no GPU in this container)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import graph_interp as gi
from oracle import models_torch as mt
from oracle import wts as owts
from oracle import yolo_post as yp
from tensorrtx_amd import capi, engine, synth
from tensorrtx_amd import wts as wts_writer
from util import synth_wts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    names = set()
    for h in ("trtx_hip.h",):
        src = open(os.path.join(ROOT, "include", h)).read()
        names |= set(re.findall(r"\b(trtx_[a-z0-9_]+)\s*\(", src))
    names -= {"trtx_plugin_vtbl", "trtx_creator_vtbl"}
    assert len(names) > 60
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    assert L.trtx_abi_version() == 1
    assert L.trtx_device_count() >= 0


def test_no_device_means_loud_failure_not_fallback():
    import torch as _t
    if _t.cuda.is_available():
        pytest.skip("GPU present")
    path, _ = synth_wts("lenet")
    plan = engine.build_plan("lenet", path, batch=1)  # building is host-only work
    with pytest.raises(capi.TrtxError) as e:
        engine.Engine(plan)
    assert e.value.status == 5  # TRTX_ERR_NO_DEVICE


@pytest.mark.parametrize("dialect", ["single", "double"])
def test_wts_loader_matches_python_restatement(tmp_path, dialect):
    rng = np.random.default_rng(0)
    tensors = {"a.weight": rng.normal(size=(3, 4)).astype(np.float32), "b.bias": np.array([1.5, -0.0, np.inf, 1e-40], np.float32),
               "empty": np.zeros((0,), np.float32), "c.num_batches_tracked": np.array([7.0], np.float32)}
    p = str(tmp_path / "t.wts")
    wts_writer.write_wts(p, tensors, dialect=dialect)
    ref = owts.load_wts(p)
    L = capi.lib()
    w = ctypes.c_void_p()
    assert L.trtx_wts_load(p.encode(), ctypes.byref(w)) == 0
    assert L.trtx_wts_count(w) == len(tensors)
    for i, (name, arr) in enumerate(tensors.items()):
        n, v, c = ctypes.c_char_p(), ctypes.POINTER(ctypes.c_float)(), ctypes.c_int64()
        assert L.trtx_wts_entry(w, i, ctypes.byref(n), ctypes.byref(v), ctypes.byref(c)) == 0
        assert n.value.decode() == name and c.value == arr.size
        got = np.ctypeslib.as_array(v, shape=(c.value,)) if c.value else np.zeros((0,), np.float32)
        assert got.tobytes() == arr.reshape(-1).tobytes() == ref[name].tobytes()  # bit exact incl. -0.0, inf, denormal
    L.trtx_wts_free(w)
    assert L.trtx_wts_load(str(tmp_path / "missing.wts").encode(), ctypes.byref(w)) == 6  # TRTX_ERR_IO


def test_lenet_builder_matches_pytorch_twin():
    path, _ = synth_wts("lenet")
    plan = engine.build_plan("lenet", path, batch=1)
    desc = engine.describe_plan(plan)
    x = torch.randn(1, 1, 32, 32, generator=torch.Generator().manual_seed(3))
    out = gi.run(desc, plan, {"data": x.numpy()})["prob"].reshape(-1)
    ref = mt.lenet(mt.Params(owts.load_wts(path)), x).reshape(-1)
    assert torch.allclose(out, ref, atol=1e-6)
    low = engine.describe_plan(plan, lowered=True)
    kinds = [o["kind"] for o in low["ops"]]
    assert kinds.count("conv") == 2 and "act_nhwc" not in kinds  # ReLU fused into both convolutions
    assert kinds.count("matmul") == 3


def test_resnet50_builder_matches_pytorch_restatement():
    path, _ = synth_wts("resnet50")
    plan = engine.build_plan("resnet50", path, batch=2, fp16=0, h=64, w=64)
    desc = engine.describe_plan(plan)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(4))
    out = gi.run(desc, plan, {"data": x.numpy()})["prob"].reshape(2, 1000)
    with torch.inference_mode():
        ref = mt.resnet50(mt.Params(owts.load_wts(path)), x)
    assert (out - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_resnet50_lowering_fuses_bn_relu_and_residual():
    path, _ = synth_wts("resnet50")
    plan = engine.build_plan("resnet50", path, batch=32, fp16=1, h=224, w=224)
    low = engine.describe_plan(plan, lowered=True)
    convs = [o for o in low["ops"] if o["kind"] == "conv"]
    assert len(convs) == 54  # 53 convs + FC
    assert convs[0]["stem"] and all(o["igemm"] for o in convs[1:])  # fp32-NCHW stem kernel, the rest on the MFMA kernel
    assert sum(o["residual"] for o in convs) == 16 and all(o["bn_folded"] for o in convs[:-1])
    kinds = {o["kind"] for o in low["ops"]}
    assert kinds <= {"conv", "pool", "to_linear"}, kinds  # no input layout pass: the stem reads NCHW fp32
    assert abs(low["flops_per_sample"] / 1e9 - 8.178) < 0.06


def test_yolov8n_builder_matches_pytorch_restatement():
    path, _ = synth_wts("yolov8n")
    plan = engine.build_plan("yolov8n", path, batch=2, h=128, w=128, fp16=1, mark_heads=1)
    desc = engine.describe_plan(plan)
    x = torch.from_numpy(synth.images(2, 128, 128, seed=5))
    out = gi.run(desc, plan, {"images": x.numpy()})
    with torch.inference_mode():
        heads, strides = mt.yolov8_det(mt.Params(owts.load_wts(path)), x)
    assert strides == [8, 16, 32]
    for i, h in enumerate(heads):
        assert (out[f"head{i}"] - h).abs().max().item() < 2e-4
    dec = yp.decode_c([h.numpy() for h in heads], 80, 128, 128, strides)
    got = out["output"].reshape(2, -1).numpy()
    assert np.array_equal(got[:, 0], dec[:, 0])


def test_yolov8n_lowering_at_benchmark_size(monkeypatch):
    path, _ = synth_wts("yolov8n")
    plan = engine.build_plan("yolov8n", path, batch=32, h=640, w=640, fp16=1)
    # layer by layer (the default): Appendix C.1's 63 convolutions = 1 stem (fp32 NCHW in) + 62 MFMA implicit-GEMM; the DFL
    # 1x1 convs are absorbed by the fused head kernel
    # round 4: the detect head's 18 convolutions - six chains of depth three over three levels (model.cpp:188-251) - go out as 6 grouped
    # launches of 3 sibling layers each (lower.cpp group_convs): 65 -> 53 ops, 62 -> 50 MFMA conv launches
    grouped = engine.describe_plan(plan, lowered=True)
    groups = [o for o in grouped["ops"] if o["kind"] == "conv_group"]
    assert len(grouped["ops"]) == 53 and len(groups) == 6 and all(len(g["members"]) == 3 for g in groups)
    assert sorted((m["cout"], m["k"][0]) for g in groups for m in g["members"]) == sorted([(64, 3)] * 6 + [(64, 1)] * 3 + [(80, 3)] * 6 + [(80, 1)] * 3)
    assert all(sorted(m["hw_in"][0] for m in g["members"]) == [20, 40, 80] for g in groups)                 # one member per pyramid level
    assert all(len({(m["cout"], m["k"][0], m["act1"]) for m in g["members"]}) == 1 for g in groups)         # the same layer of sibling branches
    assert sum(o["kind"] == "conv" for o in grouped["ops"]) + 18 == 63 and grouped["n_conv"] == 63 and grouped["n_igemm"] == 62
    monkeypatch.setenv("TRTX_GROUP_CONVS", "0")    # the rest of this test looks at the plan one launch per convolution
    assert abs(engine.describe_plan(plan, lowered=True)["flops_per_sample"] - grouped["flops_per_sample"]) < 1.0
    assert abs(engine.describe_plan(plan, lowered=True)["bytes_per_sample"] - grouped["bytes_per_sample"]) < 1.0
    monkeypatch.setenv("TRTX_FOLD_UPSAMPLE", "0")
    unfolded = engine.describe_plan(plan, lowered=True)
    assert [o["kind"] for o in unfolded["ops"]].count("resize") == 2 and len(unfolded["ops"]) == 67
    monkeypatch.delenv("TRTX_FOLD_UPSAMPLE")
    low = engine.describe_plan(plan, lowered=True)
    convs = [o for o in low["ops"] if o["kind"] == "conv"]
    assert len(convs) == 63 and convs[0]["stem"] and sum(o["igemm"] for o in convs) == 62
    assert abs(low["flops_per_sample"] / 1e9 - 8.743) < 0.01
    kinds = [o["kind"] for o in low["ops"]]
    assert sorted(set(kinds)) == ["conv", "pool_chain", "yolo_head"]  # everything else fused / aliased away
    assert len(kinds) == 65  # 63 conv + 1 fused SPPF pool chain + 1 fused head; the two nearest upsamples of the head (model.cpp:130-160)
    #                          are folded into the A-gather of the 1x1 convolutions that read them (384 -> 128 @40x40, 192 -> 64 @80x80)
    ups = [o for o in convs if o["up_c"]]
    assert [(o["cin"], o["cout"], o["up_c"], o["hw_in"]) for o in ups] == [(384, 128, 256, [40, 40]), (192, 64, 128, [80, 80])]
    # ... and the fp32 build (round 6): the same 65 launches - its SPPF pools share the chained kernel (4-channel chunks) instead of three launches
    low32 = engine.describe_plan(engine.build_plan("yolov8n", path, batch=2, h=160, w=160, fp16=0), lowered=True)
    assert sorted(o["kind"] for o in low32["ops"]) == sorted(kinds)
    assert low["bytes_per_sample"] < unfolded["bytes_per_sample"] - 1.8e6   # the convolutions' own reads: 3/4 of 0.8 + 1.6 MB per image gone
    #                                                                          (the resize launches' 0.6 MB read + 2.5 MB written were never priced)
    assert all(o["act1"] == 3 for o in convs if o["bn_folded"])  # SiLU epilogue on every Conv+BN
    assert sum(o["residual"] for o in convs) == 6  # bottleneck shortcuts of model.2/4/6/8
    assert low["arena_bytes"] < 450e6


def test_plan_roundtrip_is_byte_stable():
    path, _ = synth_wts("yolov8n")
    plan = engine.build_plan("yolov8n", path, batch=4, h=64, w=64)
    again = engine.build_plan("yolov8n", path, batch=4, h=64, w=64)
    assert plan == again
    desc = engine.describe_plan(plan)
    plug = [l for l in desc["layers"] if l["kind"] == 16][0]
    blob = bytes.fromhex(plug["plugin_blob"])
    # YoloLayer serialisation layout (yololayer.cu:75-101): 8 ints/float, 3 strides, 3 bools
    assert len(blob) == 8 * 4 + 3 * 4 + 3
    hdr = np.frombuffer(blob, dtype=np.int32, count=8)
    assert hdr[0] == 80 and hdr[3] == 256 and hdr[4] == 64 and hdr[5] == 64 and hdr[6] == 1000 and hdr[7] == 3


def test_invalid_network_fails_at_build_time(tmp_path):
    # a .wts with a wrong-sized kernel must be rejected by build-time validation, not crash
    path, tensors = synth_wts("lenet")
    bad = {k: v.numpy() for k, v in tensors.items()}
    bad["conv1.weight"] = bad["conv1.weight"].reshape(-1)[:-1]
    p = str(tmp_path / "bad.wts")
    wts_writer.write_wts(p, bad)
    with pytest.raises(capi.TrtxError):
        engine.build_plan("lenet", p, batch=1)


def test_retinaface_builder_matches_pytorch_restatement():
    from oracle import det_post as dp
    path, _ = synth_wts("retinaface_r50")
    plan = engine.build_plan("retinaface_r50", path, batch=2, fp16=1, h=128, w=160)
    desc = engine.describe_plan(plan)
    x = (torch.from_numpy(synth.images(2, 128, 160, seed=2)) * 255 - 110) / 64
    out = gi.run(desc, plan, {"data": x.numpy()})["prob"].reshape(2, -1).numpy()
    with torch.inference_mode():
        heads = mt.retinaface_r50(mt.Params(owts.load_wts(path)), x)
    dec = dp.retina_decode([h.reshape(2, 32, -1).numpy() for h in heads], 128, 160)
    assert np.array_equal(out[:, 0], dec[:, 0]) and dec[:, 0].min() > 50
    n = int(dec[0, 0])
    assert np.allclose(out[0, 1:1 + n * 15], dec[0, 1:1 + n * 15], rtol=1e-3, atol=1e-2)


def test_retinaface_lowering_at_config4_size(monkeypatch):
    monkeypatch.setenv("TRTX_GROUP_CONVS", "0")
    path, _ = synth_wts("retinaface_r50")
    plan = engine.build_plan("retinaface_r50", path, batch=1, fp16=1, h=1280, w=1280)
    low = engine.describe_plan(plan, lowered=True)
    kinds = [o["kind"] for o in low["ops"]]
    convs = [o for o in low["ops"] if o["kind"] == "conv"]
    assert len(convs) == 82 and convs[0]["stem"] and sum(o["igemm"] for o in convs) == len(convs) - 1
    assert abs(low["flops_per_sample"] / 1e9 - 354.0) < 1.0          # 354 GFLOP @1280^2
    assert "copy_nhwc" not in kinds and "act_nhwc" not in kinds      # SSH ReLU pushed into the conv epilogues, heads aliased
    # the FPN's two all-ones depthwise 2x2/2 deconvolutions (retina_r50.cpp:156-172) are what they compute: nearest upsamples
    assert kinds.count("deconv") == 0 and kinds.count("resize") == 2 and kinds.count("plugin") == 1
    plug = [l for l in engine.describe_plan(plan)["layers"] if l["kind"] == 16][0]
    assert np.frombuffer(bytes.fromhex(plug["plugin_blob"]), dtype=np.int32).tolist() == [1280, 1280]


RCNN_SMALL = dict(pre_nms_topk=100, post_nms_topk=20, detections=10)


def test_rcnn_builder_matches_pytorch_restatement():
    """Config 5 graph (HWC preprocess, C4 backbone, RPN, the five user plugins through the IPluginV2 trampoline, res5 on
    the (proposals, C, 14, 14) tensor, FC/softmax/slice) interpreted layer by layer == the PyTorch restatement."""
    path, _ = synth_wts("rcnn_r50c4")
    plan = engine.build_plan("rcnn_r50c4", path, batch=2, fp16=0, h=64, w=96, mark_stages=1, **RCNN_SMALL)
    desc = engine.describe_plan(plan)
    plugs = [l["plugin_type"] for l in desc["layers"] if l["kind"] == 16]
    assert plugs == ["RpnDecode", "RpnNms", "RoiAlign", "PredictorDecode", "BatchedNms"]
    x = torch.from_numpy(synth.images(2, 64, 96, seed=5)).permute(0, 2, 3, 1).contiguous() * 255
    res = gi.run(desc, plan, {"images": x.numpy()})
    with torch.inference_mode():
        ref = mt.rcnn_r50c4(mt.Params(owts.load_wts(path)), x, pre_nms_topk=100, post_nms_topk=20, detections_per_image=10)
    assert np.allclose(res["features"].numpy(), ref["features"].numpy(), atol=1e-4)
    assert np.allclose(res["proposals"].numpy(), ref["proposals"], atol=1e-2)
    assert np.allclose(res["scores"].numpy().reshape(2, -1), ref["scores"], atol=1e-5) and ref["scores"].max() > 0.3
    assert np.allclose(res["boxes"].numpy(), ref["boxes"], atol=1e-2)
    assert np.array_equal(res["labels"].numpy().reshape(2, -1), ref["labels"])


def test_rcnn_lowering_at_reference_size():
    path, _ = synth_wts("rcnn_r50c4")
    plan = engine.build_plan("rcnn_r50c4", path, batch=1, fp16=1)  # 800x1067, 6000 -> 1000 proposals -> 100 detections
    low = engine.describe_plan(plan, lowered=True)
    kinds = [o["kind"] for o in low["ops"]]
    convs = [o for o in low["ops"] if o["kind"] == "conv"]
    # stem + 42 backbone + 3 RPN + 10 res5 + 2 FC, all but the stem on the MFMA implicit-GEMM kernel
    assert len(convs) == 58 and convs[0]["stem"] and sum(o["igemm"] for o in convs) == 57
    # four plugins through the trampoline + RoIAlign as the engine-native NHWC op (fp16 engines), no layout pass on the RoI tensor
    assert kinds.count("plugin") == 4 and kinds.count("roi_align") == 1 and "act_nhwc" not in kinds and "ew_nhwc" not in kinds
    assert kinds.count("to_nhwc") == 0
    e = engine.describe_plan(plan)
    out_dims = {t["name"]: t["dims"] for t in e["tensors"] if t["is_output"]}
    assert out_dims == {"scores": [100, 1], "boxes": [100, 4], "labels": [100, 1]}


def _happens_before(ops):
    """closure over same-lane order and cross-lane waits, as bit masks: hb[k] has bit j set iff op j happens-before op k"""
    hb = [0] * len(ops)
    last_on_lane = {}
    for k, o in enumerate(ops):
        m = 0
        for d in o["waits"] + ([last_on_lane[o["lane"]]] if o["lane"] in last_on_lane else []):
            m |= hb[d] | (1 << d)
        hb[k] = m
        last_on_lane[o["lane"]] = k
    return hb


@pytest.mark.parametrize("model,opts", [("yolov8n", dict(batch=4, h=640, w=640, fp16=1)),
                                        ("retinaface_r50", dict(batch=1, fp16=1, h=320, w=320)),
                                        ("resnet50", dict(batch=2, fp16=1)),
                                        ("rcnn_r50c4", dict(batch=1, fp16=1, h=128, w=160, pre_nms_topk=200, post_nms_topk=40,
                                                            detections=10, mask=1))])
def test_lanes_and_arena_plan_are_race_free(model, opts):
    """Independent check of the concurrency plan the lowering emits (lanes = HIP streams, waits = events): every reader
    is ordered after every writer of its storage, and two arena blocks share bytes only if all ops touching one
    happen-before all ops touching the other."""
    path, _ = synth_wts(model)
    low = engine.describe_plan(engine.build_plan(model, path, **opts), lowered=True)
    ops, tensors, storages = low["ops"], low["tensors"], low["storages"]
    hb = _happens_before(ops)
    before = lambda a, b: (hb[b] >> a) & 1  # noqa: E731
    touch = {}
    writers = {}
    for k, o in enumerate(ops):
        for t in o["in"] + o["out"]:
            touch.setdefault(tensors[t]["storage"], set()).add(k)
        for t in o["out"]:
            writers.setdefault(tensors[t]["storage"], set()).add(k)
    for k, o in enumerate(ops):            # RAW (storage granularity, stricter than the planner's channel ranges)
        for t in o["in"]:
            for w in writers.get(tensors[t]["storage"], ()):
                if w < k and w not in [k]:
                    assert before(w, k), (model, "op", k, "reads storage written by unordered op", w)
    used = set(touch)
    for si, s in enumerate(storages):      # plugin / head workspaces: arena blocks no tensor refers to
        if s["kind"] == 0 and si not in used and s["last"] >= 0:
            touch[si] = {s["first"]}
    arena = [si for si, s in enumerate(storages) if s["kind"] == 0 and si in touch]
    for i, a in enumerate(arena):
        for b in arena[i + 1:]:
            sa, sb = storages[a], storages[b]
            if sa["offset"] + sa["bytes"] <= sb["offset"] or sb["offset"] + sb["bytes"] <= sa["offset"]:
                continue
            ab = all(before(x, y) for x in touch[a] for y in touch[b])
            ba = all(before(y, x) for x in touch[a] for y in touch[b])
            assert ab or ba, (model, "arena blocks", a, b, "overlap without ordering")
    if model in ("yolov8n", "retinaface_r50"):
        assert low["n_lanes"] > 1  # independent head branches really are spread over streams


def test_product_side_yolov8n_weights_match_the_test_generator():
    """bench.py must not touch the oracle outside its cpu_baseline leg, so it draws its synthetic YOLOv8n weights with
    tensorrtx_amd.synth.yolov8n_state; that generator and the oracle-driven one used by the tests stay identical."""
    _, tensors = synth_wts("yolov8n")
    sd = synth.yolov8n_state(0)
    assert list(sd) == list(tensors)
    assert all(np.array_equal(sd[k], tensors[k].numpy()) for k in sd)


@pytest.mark.parametrize("model", ["resnet50", "retinaface_r50", "rcnn_r50c4"])
def test_product_side_weights_of_the_other_bench_configs_match_the_test_generator(model):
    """bench.py takes the C2 / C4 / C5 weights from tensorrtx_amd.synth too (no tests/ or oracle/ import outside its
    cpu_baseline leg); same names, same order, same bytes as the oracle-driven generator, so the engines and their numbers are the same."""
    _, tensors = synth_wts(model)
    sd = synth.STATE[model](0)
    assert list(sd) == list(tensors)
    assert all(np.array_equal(sd[k], tensors[k].numpy()) for k in sd)


def test_mask_rcnn_builder_matches_pytorch_restatement():
    """MASK_ON (rcnn.cpp:202-232): second RoIAlign on the final boxes, res5 with shared weights, ConvTranspose + ReLU,
    1x1 predictor, MaskRcnnInference plugin; interpreted layer by layer == the PyTorch restatement."""
    path, _ = synth_wts("rcnn_r50c4")
    plan = engine.build_plan("rcnn_r50c4", path, batch=1, fp16=0, h=64, w=96, mask=1, **RCNN_SMALL)
    desc = engine.describe_plan(plan)
    plugs = [l["plugin_type"] for l in desc["layers"] if l["kind"] == 16]
    assert plugs == ["RpnDecode", "RpnNms", "RoiAlign", "PredictorDecode", "BatchedNms", "RoiAlign", "MaskRcnnInference"]
    out_dims = {t["name"]: t["dims"] for t in desc["tensors"] if t["is_output"]}
    assert out_dims["masks"] == [10, 1, 14, 14]
    x = torch.from_numpy(synth.images(1, 64, 96, seed=5)).permute(0, 2, 3, 1).contiguous() * 255
    res = gi.run(desc, plan, {"images": x.numpy()})
    with torch.inference_mode():
        ref = mt.rcnn_r50c4(mt.Params(owts.load_wts(path)), x, pre_nms_topk=100, post_nms_topk=20, detections_per_image=10,
                            mask_on=True)
    assert np.array_equal(res["labels"].numpy().reshape(1, -1), ref["labels"])
    got = res["masks"].numpy().reshape(ref["masks"].shape)
    assert np.allclose(got, ref["masks"], atol=1e-5) and 0.05 < float(ref["masks"].mean()) < 0.95 and ref["masks"].std() > 0.01


def test_c_abi_rejects_bad_arguments_before_touching_the_device():
    """Error behaviour of the operator entry points: null pointers / impossible sizes come back as TRTX_ERR_INVALID (1),
    capacities beyond the kernel's design as TRTX_ERR_UNSUPPORTED (4) - no launch, no crash, also without a GPU."""
    L = capi.lib()
    null = ctypes.c_void_p()
    buf = (ctypes.c_float * 16)()
    f = ctypes.c_float
    assert L.trtx_yolo_nms(null, 1, 1000, f(0.5), f(0.45), null, null, null, null, ctypes.c_size_t(0), null) == 1
    assert L.trtx_yolo_nms(buf, 1, 4096, f(0.5), f(0.45), buf, buf, null, buf, ctypes.c_size_t(1 << 30), null) == 4
    assert L.trtx_yolo_nms(buf, 1, 1000, f(0.5), f(0.45), buf, buf, null, buf, ctypes.c_size_t(16), null) == 3  # workspace too small
    assert L.trtx_yolo_postprocess_gpu(null, 1, 1000, f(0.5), f(0.45), buf, null) == 1
    assert L.trtx_yolo_postprocess_gpu(buf, 1, 2000, f(0.5), f(0.45), buf, null) == 4
    assert L.trtx_mask_rcnn_inference(1, null, buf, 10, 14, 80, buf, null) == 1
    assert L.trtx_mask_rcnn_inference(0, buf, buf, 10, 14, 80, buf, null) == 1
    assert L.trtx_rpn_nms(1, null, buf, 6000, 1000, f(0.7), buf, buf, ctypes.c_size_t(64), null) == 1
    assert L.trtx_batched_nms(1, 1, buf, buf, null, 1000, 100, f(0.5), buf, buf, buf, buf, ctypes.c_size_t(64), null) == 1
    L.trtx_status_string.restype = ctypes.c_char_p
    assert L.trtx_status_string(3) == b"workspace too small" and L.trtx_status_string(5) == b"no HIP device"


@pytest.mark.parametrize("order", ["package_first", "torch_first", "entry_build_first"])
def test_single_hip_runtime_whatever_the_import_order(order):
    """Round-1 regression: dlopening libtrtx_hip.so before torch mapped two libamdhip64 runtimes (torch saw the GPU,
    trtx_device_count() did not).  The binding now maps torch's runtime first; assert exactly one copy in a fresh
    process for every import order a user script / the driver can produce."""
    import subprocess
    import sys
    pre = {"package_first": "import tensorrtx_amd; tensorrtx_amd.lib(); import torch",
           "torch_first": "import torch; import tensorrtx_amd; tensorrtx_amd.lib()",
           "entry_build_first": "import __graft_entry__ as g; g.build(); import torch"}[order]
    code = pre + "; from tensorrtx_amd import capi, engine; engine.models_lib(); r = capi.hip_runtimes_mapped(); " \
                 "print(len(r), r); assert len(r) == 1, r"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr


def test_rcnn_fp16_plan_runs_roi_align_natively():
    """fp16 R-CNN engines replace the RoiAlign plugin edge (fp32 LINEAR in/out + two layout passes) by the NHWC kernel; fp32 engines
    keep the plugin."""
    path, _ = synth_wts("rcnn_r50c4")
    kinds = {}
    for fp16 in (1, 0):
        plan = engine.build_plan("rcnn_r50c4", path, batch=1, fp16=fp16, h=320, w=416, mask=1)
        ops = engine.describe_plan(plan, lowered=True)["ops"]
        kinds[fp16] = [o["kind"] for o in ops]
    assert kinds[1].count("roi_align") == 2 and kinds[0].count("roi_align") == 0   # box head + mask head
    assert kinds[1].count("plugin") == kinds[0].count("plugin") - 2
    assert kinds[1].count("to_nhwc") < kinds[0].count("to_nhwc") or kinds[1].count("to_nhwc") <= 2


@pytest.mark.parametrize("task,name,nc,extra", [(1, "yolov8n_seg", 80, 32), (2, "yolov8n_pose", 1, 51), (3, "yolov8n_obb", 15, 1)])
def test_yolov8_task_graphs_build_and_lower_on_the_host(task, name, nc, extra):
    """buildEngineYolov8Seg / Pose / Obb (yolov8/src/model.cpp:1057-1308, 1310-1563, 2499-2740): the cv4 branch widens each plugin
    input to 4 + classes + extra rows, "seg" adds the proto output, and the branch plugin is not folded into the det-only head kernel."""
    from util import synth_wts
    path, _ = synth_wts(name)
    plan = engine.build_plan("yolov8n", path, batch=2, h=128, w=128, fp16=1, task=task, classes=nc, mark_heads=1)
    outs = {t["name"]: t["dims"] for t in engine.describe_plan(plan)["tensors"] if t.get("is_output")}
    assert outs["head0"] == [4 + nc + extra, 256] and outs["head2"] == [4 + nc + extra, 16] and outs["output"][0] == 1 + 1000 * 90
    assert ("proto" in outs) == (task == 1)
    if task == 1:
        assert outs["proto"] == [32, 32, 32]
    kinds = [o["kind"] for o in engine.describe_plan(plan, lowered=True)["ops"]]
    assert kinds.count("plugin") == 1 and "yolo_head" not in kinds
    with pytest.raises(Exception):
        engine.build_plan("yolov8n", path, batch=1, h=128, w=128, task=7)


def test_max_aux_streams_travels_in_the_plan_and_bounds_the_lanes(monkeypatch):
    """IBuilderConfig::setMaxAuxStreams (TensorRT >= 8.6): stored in the plan (format 3), lanes = 1 + aux streams."""
    from util import synth_wts
    path, _ = synth_wts("yolov8n")
    for aux, lanes in ((None, 2), (0, 1), (1, 2), (3, 2)):   # grouped sibling layers (round 4) leave two independent branches: the cv2 and cv3 arms
        kw = {} if aux is None else dict(aux_streams=aux)
        assert engine.describe_plan(engine.build_plan("yolov8n", path, batch=2, h=128, w=128, fp16=1, **kw), lowered=True)["n_lanes"] == lanes
    monkeypatch.setenv("TRTX_GROUP_CONVS", "0")
    for aux, lanes in ((None, 4), (0, 1), (1, 2), (3, 4)):
        kw = {} if aux is None else dict(aux_streams=aux)
        plan = engine.build_plan("yolov8n", path, batch=2, h=128, w=128, fp16=1, **kw)
        assert engine.describe_plan(plan)["max_aux_streams"] == (-1 if aux is None else aux)
        assert engine.describe_plan(plan, lowered=True)["n_lanes"] == lanes
    with pytest.raises(Exception):
        engine.build_plan("yolov8n", path, batch=2, h=128, w=128, aux_streams=99)


def test_conv_tactics_are_enumerated_on_the_host():
    """conv_tactics (what runtime/tune.cpp times per layer): entry 0 is the untuned default, every entry shares the layer's packed
    weights (tile widths divide the padded Cout, 64-wide k-steps only where Cin % 64 == 0), no duplicates."""
    from tensorrtx_amd import capi
    t = capi.conv2d_tactics(32, 20, 20, 128, 128, 3, 1, 1)
    assert t[0] == (128, 32, 128, 1, 1, 0) and len(set(t)) == len(t) >= 8
    assert (64, 32, 128, 2, 1, 0) in t and (64, 64, 64, 1, 1, 0) in t    # wave-split-K on 64-wide tiles; 64-row tiles with 64-wide k-steps
    assert not any(x[5] for x in t)   # the 3x3 row-reuse kernel is compiled out of the product (conv_igemm.hip: co-scheduling hazard) ...
    import subprocess, sys
    r3 = subprocess.run([sys.executable, "-c", "from tensorrtx_amd import capi; print(capi.conv2d_tactics(32, 20, 20, 128, 128, 3, 1, 1))"],
                        env=dict(os.environ, TRTX_TACTICS_R3="1"), capture_output=True, text=True, check=True)
    assert eval(r3.stdout.strip().splitlines()[-1]) == t   # ... and no environment variable brings it back (ADVICE r3 / VERDICT r3 Weak 3)
    assert all(128 % bn == 0 and bk in (32, 64) and bm in (64, 128) for bn, bk, bm, _, _, _ in t)
    t = capi.conv2d_tactics(32, 80, 80, 32, 32, 3, 1, 1)                  # weight-stationary kernel is the default where it applies
    assert t[0][4] == 2 and all(x[4] in (1, 7) for x in t[1:]) and all(x[1] == 32 for x in t)
    assert [x for x in t if x[4] == 7] == [(32, 32, 128, 1, 7, 0)]      # round 6: the resident-operand 3x3 kernel, at the column tile that holds the whole Cout
    assert (64, 32, 128, 1, 8, 0) in capi.conv2d_tactics(32, 80, 80, 128, 64, 1, 1, 0)   # ... and its 1x1 sibling
    t = capi.conv2d_tactics(32, 80, 80, 64, 80, 3, 1, 1)
    assert {x[0] for x in t} == {80}
    assert capi.conv2d_tactics(32, 160, 160, 16, 16, 3, 1, 1) == [(16, 32, 128, 1, 1, 0), (16, 32, 128, 1, 8, 0)]   # two taps per k-step: one tile + (round 6) the A-direct kernel's thin 3x3 form
    assert capi.conv2d_tactics(32, 150, 150, 16, 16, 3, 1, 1) == [(16, 32, 128, 1, 1, 0)]                          # ... which wants output rows of whole 16-pixel fragments
    assert not any(x[5] for x in capi.conv2d_tactics(32, 40, 40, 64, 64, 3, 2, 1))        # stride 2: no row reuse
    assert not any(x[5] for x in capi.conv2d_tactics(32, 40, 40, 128, 128, 1, 1, 0))


def test_fp32_conv_tactics_are_enumerated_on_the_host():
    """conv_tactics_f32 (kernels/conv_igemm_f32.hip; what the tuner times for a plan built without kFP16): (bn, bm, operand path, channels per k-step);
    operand path 1 = LDS-DMA, 3 = resident patch, 5 = through registers, 6 = fetching / multiplying wave roles, 7 / 8 = the resident-operand kernels of
    conv_res.hip (3x3 / 1x1, round 6).  Every entry returns the same bits."""
    import subprocess
    import sys
    from tensorrtx_amd import capi
    t = capi.conv2d_tactics_f32(32, 80, 80, 64, 64, 3, 1, 1)
    assert len(set(t)) == len(t) and t[0][2] == 1 and t[0][3] == 16                   # the launcher's own choice: LDS-DMA, 16-channel steps
    assert {x[2] for x in t} == {1, 3, 5, 6, 7}
    assert sorted(x for x in t if x[2] == 7) == [(16, 128, 7, 16), (32, 128, 7, 16)]   # resident operands: 64 -> 64 as 4 x 16 or 2 x 32 columns (LDS)
    assert [x for x in t if x[2] == 3] == [(64, 128, 3, 16)]                           # one resident-patch entry, at the widest column tile
    assert any(x[3] == 32 for x in t) and all(x[3] == 16 for x in t if x[2] != 1)      # 32-channel steps exist for the LDS-DMA path only
    assert all(64 % x[0] == 0 and x[1] in (64, 128, 256) for x in t)
    t1 = capi.conv2d_tactics_f32(32, 80, 80, 128, 64, 1, 1, 0)
    assert sorted(x[0] for x in t1 if x[2] == 8) == [16, 32, 64] and not any(x[2] in (3, 7) for x in t1)   # the 1x1 form: every column tile whose weights fit
    assert not any(x[2] == 3 for x in capi.conv2d_tactics_f32(32, 40, 40, 64, 64, 3, 2, 1))      # stride 2: no resident patch
    assert not any(x[2] == 3 for x in capi.conv2d_tactics_f32(32, 20, 20, 256, 64, 3, 1, 1))     # 256 input channels: sixteen planes do not fit
    assert {x[2] for x in capi.conv2d_tactics_f32(32, 320, 320, 3, 16, 3, 2, 1, ld_in=4)} <= {1}  # the padded 3-channel case: two taps per step, DMA only
    off = subprocess.run([sys.executable, "-c", "from tensorrtx_amd import capi; print(capi.conv2d_tactics_f32(32, 80, 80, 64, 64, 3, 1, 1))"],
                         env=dict(os.environ, TRTX_CONV_ROLES="0"), capture_output=True, text=True, check=True)
    assert eval(off.stdout.strip().splitlines()[-1]) == [x for x in t if x[2] != 6]    # TRTX_CONV_ROLES=0 removes exactly the role variants
    off = subprocess.run([sys.executable, "-c", "from tensorrtx_amd import capi; print(capi.conv2d_tactics_f32(32, 80, 80, 64, 64, 3, 1, 1))"],
                         env=dict(os.environ, TRTX_CONV_RES="0"), capture_output=True, text=True, check=True)
    assert eval(off.stdout.strip().splitlines()[-1]) == [x for x in t if x[2] != 7]    # TRTX_CONV_RES=0: exactly the resident-operand entries


def test_int8_tensor_on_a_convolution_without_the_mfma_path_falls_back_to_fp16():
    """kINT8 promises that layers which cannot run in int8 stay in fp16.  The assignment screens by shape only; whether a convolution
    really takes the implicit-GEMM path is decided later from strides / offsets / K.  Here a 1x1 convolution over 16 channels (K = 16:
    the direct kernel) produces a 32-channel tensor with a calibrated scale that a 3x3 convolution consumes - the producer cannot write
    int8, so the tensor must stay fp16 (it used to be marked int8 and the whole build then failed in finalize)."""
    import struct
    from tensorrtx_amd import calibrator, capi
    L = capi.lib()

    class Dims(ctypes.Structure):
        _fields_ = [("nb", ctypes.c_int32), ("d", ctypes.c_int64 * 8)]

    def dims(*v):
        d = Dims()
        d.nb = len(v)
        for i, x in enumerate(v):
            d.d[i] = x
        return d

    rng = np.random.default_rng(0)

    def build(int8, cache=None):
        b, n, m = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        check = lambda st: (_ for _ in ()).throw(AssertionError(st)) if st != 0 else None
        check(L.trtx_builder_create(ctypes.byref(b)))
        check(L.trtx_builder_set_max_batch(b, 2))
        check(L.trtx_builder_set_flag(b, 0, 1))
        cal = None
        if int8:
            check(L.trtx_builder_set_flag(b, 1, 1))
            cal = calibrator.Calibrator(cache=cache)
            check(L.trtx_builder_set_int8_calibrator(b, ctypes.byref(cal.vtbl)))
        check(L.trtx_network_create(b, 0, ctypes.byref(n)))
        x = L.trtx_add_input(n, b"data", 0, ctypes.byref(dims(3, 32, 32)))

        def conv(x, cin, cout, k):
            w = np.ascontiguousarray(rng.normal(0, 0.1, (cout, cin, k, k)), dtype=np.float32)
            bias = np.zeros(cout, dtype=np.float32)
            l = L.trtx_add_convolution(n, x, cout, k, k, w.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(w.size),
                                       bias.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(cout))
            assert l >= 0
            pad = (ctypes.c_int32 * 2)(k // 2, k // 2)
            check(L.trtx_layer_set_ints(n, l, 2, pad, 2))   # TRTX_PARAM_PADDING
            r = L.trtx_add_activation(n, L.trtx_layer_output(n, l, 0), 1)
            return L.trtx_layer_output(n, r, 0)

        x = conv(x, 3, 16, 3)     # the stem
        x = conv(x, 16, 32, 1)    # K = 16: direct kernel, cannot quantise its output
        x = conv(x, 32, 32, 3)    # MFMA-eligible consumer
        x = conv(x, 32, 32, 3)
        check(L.trtx_tensor_set_name(n, x, b"out"))
        check(L.trtx_mark_output(n, x))
        st = L.trtx_build_serialized(b, n, ctypes.byref(m))
        err = ctypes.string_at(ctypes.c_char_p(L.trtx_network_last_error(n))).decode() if st else ""
        assert st == 0, err
        L.trtx_hostmem_data.restype = ctypes.c_void_p
        L.trtx_hostmem_size.restype = ctypes.c_size_t
        plan = ctypes.string_at(L.trtx_hostmem_data(m), L.trtx_hostmem_size(m))
        L.trtx_hostmem_destroy(m)
        L.trtx_network_destroy(n)
        L.trtx_builder_destroy(b)
        return plan

    L.trtx_network_last_error.restype = ctypes.c_void_p
    rng = np.random.default_rng(0)
    plan16 = build(False)
    names = [t["name"] or f"(Unnamed Tensor* {t['id']})" for t in engine.describe_plan(plan16)["tensors"]]
    cache = b"TRT-8601-EntropyCalibration2\n" + b"".join(f"{nm}: {struct.unpack('<I', struct.pack('<f', 0.05))[0]:08x}\n".encode() for nm in names)
    rng = np.random.default_rng(0)
    low = engine.describe_plan(build(True, cache), lowered=True)
    convs = [o for o in low["ops"] if o["kind"] == "conv"]
    assert len(convs) == 4
    assert convs[1]["i8"] == [0, 0, 0] or convs[1]["i8"][:2] == [0, 0]      # the K = 16 layer reads and writes fp16
    assert convs[2]["i8"][0] == 0 and convs[2]["i8"][1] == 1                 # its consumer reads fp16 and quantises for the next layer
    assert convs[3]["i8"][0] == 1


def _upsample_concat_net(extra_reader=False, cin_up=64, cin_skip=32, k=1, fp16=True):
    """Upsample(2x nearest) -> Concat([up, skip]) -> Conv k x k, the YOLOv8 head shape (model.cpp:130-160), optionally with a second reader of the
    upsampled tensor.  Returns (plan, weights) - seeded."""
    from tensorrtx_amd import builder
    rng = np.random.default_rng(3)
    net = builder.Network(max_batch=2, fp16=fp16)
    x = net.input("data", (3, 16, 24))
    w = {}
    w["a"] = rng.normal(0, 0.2, (cin_up, 3, 3, 3)).astype(np.float32)       # low-resolution branch: stride 2
    w["b"] = rng.normal(0, 0.2, (cin_skip, 3, 3, 3)).astype(np.float32)     # skip branch at full resolution
    w["c"] = rng.normal(0, 0.1, (48, cin_up + cin_skip, k, k)).astype(np.float32)
    low = net.out(net.activation(net.out(net.conv(x, w["a"], stride=2, padding=1)), "relu"))
    skip = net.out(net.activation(net.out(net.conv(x, w["b"], stride=1, padding=1)), "relu"))
    up = net.out(net.resize_nearest(low, 2))
    cat = net.out(net.concat([up, skip]))
    y = net.out(net.activation(net.out(net.conv(cat, w["c"], padding=k // 2)), "relu"))
    net.mark_output(y, "y")
    if extra_reader:
        w["d"] = rng.normal(0, 0.1, (16, cin_up, 1, 1)).astype(np.float32)
        net.mark_output(net.out(net.conv(up, w["d"])), "z")
    plan = net.build()
    net.close()
    return plan, w


def test_upsample_is_folded_only_where_it_is_safe(monkeypatch):
    """lower.cpp fold_upsample: the resize disappears into the 1x1 convolution that reads the concat buffer - but not when something else
    reads the upsampled tensor, not into a 3x3, not when the slice is not a whole number of 64-channel k-steps, and not with
    TRTX_FOLD_UPSAMPLE=0.  Since round 5 fp32 engines fold too (their convolutions run on the same skeleton, kernels/conv_igemm_f32.hip)."""
    def kinds(plan):
        low = engine.describe_plan(plan, lowered=True)
        return [o["kind"] for o in low["ops"]], [o for o in low["ops"] if o["kind"] == "conv"]
    k, convs = kinds(_upsample_concat_net()[0])
    assert "resize" not in k and [o["up_c"] for o in convs if o["up_c"]] == [64]
    assert "resize" in kinds(_upsample_concat_net(extra_reader=True)[0])[0]
    assert "resize" in kinds(_upsample_concat_net(k=3)[0])[0]
    assert "resize" in kinds(_upsample_concat_net(cin_up=32)[0])[0]
    k32, convs32 = kinds(_upsample_concat_net(fp16=False)[0])
    assert "resize" not in k32 and [o["up_c"] for o in convs32 if o["up_c"]] == [64]
    monkeypatch.setenv("TRTX_FOLD_UPSAMPLE", "0")
    assert "resize" in kinds(_upsample_concat_net()[0])[0]


def _conv_bn_mish_net(fp16, fused=True, cin=16, cout=32, hw=(12, 20)):
    """convBnMish of the reference (yolov4/yolov4.cpp:199-213): Conv (no bias) -> Scale (folded BatchNorm) -> "Mish_TRT" from the plugin
    registry; `fused=False` puts a max-pool between the Scale and the plugin so that no convolution can absorb it.  Seeded."""
    from tensorrtx_amd import builder
    rng = np.random.default_rng(4)
    w = rng.normal(0, 0.25, (cout, cin, 3, 3)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 1.0, cout).astype(np.float32)
    net = builder.Network(max_batch=2, fp16=fp16)
    x = net.input("data", (cin,) + hw)
    t = net.out(net.scale(net.out(net.conv(x, w, stride=1, padding=1)), shift, scale))
    if not fused:
        t = net.out(net.pooling(t, 2, 2))
    net.mark_output(net.out(net.plugin([t], "Mish_TRT")), "out")
    plan = net.build()
    net.close()
    return plan, (w, scale, shift)


@pytest.mark.parametrize("fp16", [False, True])
def test_mish_plugin_is_an_epilogue_activation(fp16):
    """g1: the built-in Mish_TRT after Conv -> Scale becomes the convolution's epilogue (act code 6 = ACT_MISH), no plugin op
    and no layout pass; behind anything else it is an activation op in the tensor's own layout.  The plan keeps the plugin layer and its
    blob (int input_size, yolov4/mish.cu:24-32) so it round-trips through the registry."""
    plan, _ = _conv_bn_mish_net(fp16)
    low = engine.describe_plan(plan, lowered=True)
    kinds = [o["kind"] for o in low["ops"]]
    convs = [o for o in low["ops"] if o["kind"] == "conv"]
    assert "plugin" not in kinds and len(convs) == 1 and convs[0]["act1"] == 6, kinds
    assert [k for k in kinds if k not in ("conv", "to_nhwc", "to_linear")] == [], kinds
    net = engine.describe_plan(plan)
    plug = [l for l in net["layers"] if l.get("plugin_type")]
    assert len(plug) == 1 and plug[0]["plugin_type"] == "Mish_TRT" and plug[0]["plugin_blob"] == __import__("struct").pack("<i", 32 * 12 * 20).hex()
    low2 = engine.describe_plan(_conv_bn_mish_net(fp16, fused=False)[0], lowered=True)
    k2 = [o["kind"] for o in low2["ops"]]
    assert "plugin" not in k2 and "act_nhwc" in k2, k2


def test_roi_align_emits_only_the_bins_its_stride_2_readers_read(monkeypatch):
    """res5.0's conv1 and shortcut are 1x1 STRIDE-2 convolutions (rcnn/backbone.hpp:9,110-117): the fp16 plan's RoIAlign
    writes the 7x7 even bins instead of 14x14 and both readers run at stride 1; TRTX_ROIALIGN_FOLD_STRIDE=0 keeps the full grid."""
    path, _ = synth_wts("rcnn_r50c4")

    def head(env):
        if env is not None:
            monkeypatch.setenv("TRTX_ROIALIGN_FOLD_STRIDE", env)
        ops = engine.describe_plan(engine.build_plan("rcnn_r50c4", path, batch=1, fp16=1, h=320, w=416), lowered=True)["ops"]
        k = next(i for i, o in enumerate(ops) if o["kind"] == "roi_align")
        readers = [o for o in ops[k + 1:] if o["kind"] == "conv" and o["in"][0] == ops[k]["out"][0]]
        return ops[k], readers
    ra, readers = head(None)
    assert "every 2nd bin" in ra["name"] and len(readers) == 2
    assert all(o["k"] == [1, 1] and o["stride"] == [1, 1] and o["hw_in"] == [7, 7] and o["hw_out"] == [7, 7] for o in readers)
    ra0, readers0 = head("0")
    assert "every 2nd bin" not in ra0["name"] and all(o["stride"] == [2, 2] and o["hw_in"] == [14, 14] and o["hw_out"] == [7, 7] for o in readers0)
    assert ra["bytes"] < 0.3 * ra0["bytes"]
