"""ONE table of numeric parity bounds for the GPU tests, and the rule that keeps them honest.

Every GPU parity test reports what it measured through `check()`; check() appends the numbers to gpurun_out/parity_metrics.jsonl (copied
to profiles/rNN_parity_metrics.jsonl at round end, several runs concatenated) and asserts the ones that have a bound here.  The CPU test
tests/test_parity_bounds.py then requires every bound to sit within 1.5x of the WORST value on record - in both directions: a bound the
record violates is wrong, and a bound looser than 1.5x the worst measurement protects nothing.

kind "max": value must stay below the bound; worst = largest measured;   bound <= 1.5 * worst.
kind "min": value must stay above the bound (fractions, IoU: 1.0 is perfect); worst = smallest measured, compared as SHORTFALLS from 1:
            (1 - bound) <= 1.5 * (1 - worst) + slack.  `slack` is an absolute allowance for discrete metrics whose recorded worst is
            perfect (a matched fraction of 1.0 over N items): one item of N.

fp16 numbers move a little from run to run (tactics are timed when a plan is BUILT, and kernels that split or reorder K round differently;
one plan is reproducible - see tests/test_gpu_tactics.py); the record therefore holds several runs.  The north_star's 1e-4 logit / 1e-3
IoU tolerance is what the fp32 builds are asserted at; the fp16 bounds are the measured cost of fp16 storage through 60+ layers.
"""
import json
import os

# (test, case) -> {metric: (kind, bound[, slack])}
BOUNDS = {
    ("lenet_fp32", None): {"max_abs_err": ("max", 3.1e-7)},
    ("resnet50_64", "fp32"): {"max_abs_err": ("max", 1.7e-3)},          # logits up to 720: 2.3e-6 relative
    ("resnet50_64", "fp16"): {"max_abs_err": ("max", 1.1)},             # 1.5e-3 relative
    ("resnet50_224_b32", None): {"max_abs_err": ("max", 1.9)},          # logits up to 1787: 1.1e-3 relative
    ("yolov8n_fp32_128", None): {"head_max_abs_err": ("max", 1.85e-5)},  # north_star: 1e-4 on O(10) logits
    ("yolov8n_fp16_640", None): {"cls_logit_max_abs_err": ("max", 0.115), "box_ltrb_max_abs_err": ("max", 0.03),
                                 "matched_fraction": ("min", 0.9979), "min_iou": ("min", 0.995), "max_conf_err": ("max", 0.0155)},
    ("yolov8n_fp16_640_fused", None): {"matched_fraction": ("min", 0.9979), "min_iou": ("min", 0.9955), "max_conf_err": ("max", 0.0155)},
    ("yolov8n_fp16_640_b32", None): {"matched_fraction": ("min", 0.9983), "min_iou": ("min", 0.9982), "max_conf_err": ("max", 0.0079)},
    ("retinaface_r50_fp16", "256x320"): {"matched_fraction": ("min", 0.9997, 1 / 4177), "min_iou": ("min", 0.9845),
                                         "max_box_err": ("max", 1.27), "max_conf_err": ("max", 0.0087)},
    ("retinaface_r50_fp16", "1280x1280"): {"matched_fraction": ("min", 0.99997, 1 / 41696), "min_iou": ("min", 0.9868),
                                           "max_box_err": ("max", 1.69), "max_conf_err": ("max", 0.01145)},
    ("rcnn_fp32", None): {"feat_err": ("max", 1.15e-5), "score_err": ("max", 4.9e-6), "proposals_matched": ("min", 0.98, 1 / 50),
                          "detections_matched": ("min", 0.95, 1 / 20)},
    ("rcnn_fp16", "320x416"): {"feat_rel_err": ("max", 2.66e-3), "proposals_matched": ("min", 0.9925), "detections_matched": ("min", 0.97),
                               "top_score_err": ("max", 4.0e-4)},
    ("rcnn_fp16", "800x1067"): {"feat_rel_err": ("max", 2.6e-3), "proposals_matched": ("min", 0.994), "detections_matched": ("min", 0.955),
                                "top_score_err": ("max", 7.8e-4)},
    ("rcnn_fp16", "800x1333"): {"feat_rel_err": ("max", 2.84e-3), "proposals_matched": ("min", 0.9895), "detections_matched": ("min", 0.955),
                                "top_score_err": ("max", 1.04e-3)},
    ("mask_rcnn_fp32", None): {"mask_err": ("max", 1.9e-6)},
    ("mask_rcnn_fp16", None): {"mask_err": ("max", 1.3e-3)},
    ("yolov8n_int8_320", None): {"head_max_abs_err_fp16": ("max", 0.087), "head_mean_rel_err_int8": ("max", 0.094)},
}

RECORD = os.path.join("gpurun_out", "parity_metrics.jsonl")


def record(test, case=None, **values):
    os.makedirs(os.path.dirname(RECORD), exist_ok=True)
    with open(RECORD, "a") as f:
        f.write(json.dumps(dict(test=test, case=case, **values)) + "\n")


def holds(kind, bound, value):
    return value < bound if kind == "max" else value >= bound


def check(test, case=None, **values):
    """Record `values`, then assert every one that has a bound (all of them reported on failure)."""
    record(test, case, **values)
    bounds = BOUNDS.get((test, case), {})
    missing = [m for m in bounds if m not in values]
    assert not missing, f"{test}/{case}: bounded metrics not reported: {missing}"
    bad = {m: (values[m], bounds[m][:2]) for m in bounds if not holds(bounds[m][0], bounds[m][1], values[m])}
    assert not bad, f"{test}/{case}: outside the parity bounds (value, (kind, bound)): {bad}"
