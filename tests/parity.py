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

Round 5: leaving the fitted band FAILS the test again (TRTX_PARITY_DRIFT=warn for exploratory builds) - see check().
Round 4 (VERDICT r3 Weak 2): the fitted table is a DRIFT ALARM, not a tolerance - it would pass at whatever accuracy the product
had when the record was taken, and re-running the generator absorbs a regression.  The tolerance is `CEILINGS` below: hand-written,
derived from the north_star and from the arithmetic of the precision (never from a product measurement), never touched by
tools/parity_bounds_from_record.py, ASSERTED by check().  A value inside the tolerance but outside the fitted band raises a ParityDrift warning
and is recorded (test "parity_drift"): results of an fp16 build move a little with the tactics timed on its box, and an alarm is not a verdict.  tests/test_parity_bounds.py requires a ceiling for every bounded metric
and that the generator's block is the only part of this file the tool may rewrite.
"""
import json
import math
import os

# (test, case) -> {metric: (kind[, slack])}: WHICH metrics are bounded and how; the numbers live in the table below
SPEC = {
    ("lenet_fp32", None): {"max_abs_err": ("max",)},
    ("resnet50_64", "fp32"): {"max_abs_err": ("max",)},                 # logits up to 720
    ("resnet50_64", "fp16"): {"max_abs_err": ("max",)},
    ("resnet50_224_b32", None): {"max_abs_err": ("max",)},              # logits up to 1787
    ("yolov8n_fp32_128", None): {"head_max_abs_err": ("max",)},         # north_star: 1e-4 on O(10) logits
    ("yolov8n_fp32_640_b32", None): {"head_max_abs_err_vs_fp64": ("max",), "head_max_abs_err": ("max",), "matched_fraction": ("min", 1 / 945), "min_iou": ("min",),
                                     "max_conf_err": ("max",)},   # the fp32
    # build at the bench configuration (round 5: v_mfma_f32_16x16x4_f32 convolutions) - THE row BASELINE's tolerance is asserted on
    ("yolov8n_fp16_640", None): {"cls_logit_max_abs_err": ("max",), "box_ltrb_max_abs_err": ("max",), "matched_fraction": ("min",),
                                 "min_iou": ("min",), "max_conf_err": ("max",)},
    ("yolov8n_fp16_640_fused", None): {"matched_fraction": ("min",), "min_iou": ("min",), "max_conf_err": ("max",)},
    ("yolov8n_fp16_640_b32", None): {"matched_fraction": ("min", 1 / 881), "min_iou": ("min",), "max_conf_err": ("max",)},
    ("retinaface_r50_fp16", "256x320"): {"matched_fraction": ("min", 1 / 4177), "min_iou": ("min",), "max_box_err": ("max",), "max_conf_err": ("max",)},
    ("retinaface_r50_fp16", "1280x1280"): {"matched_fraction": ("min", 1 / 41696), "min_iou": ("min",), "max_box_err": ("max",), "max_conf_err": ("max",)},
    ("rcnn_fp32", None): {"feat_err": ("max",), "score_err": ("max",), "proposals_matched": ("min", 1 / 50), "detections_matched": ("min", 1 / 20)},
    ("rcnn_fp16", "320x416"): {"feat_rel_err": ("max",), "proposals_matched": ("min", 1 / 200), "detections_matched": ("min", 1 / 50), "top_score_err": ("max",)},
    ("rcnn_fp16", "800x1067"): {"feat_rel_err": ("max",), "proposals_matched": ("min", 1 / 1000), "detections_matched": ("min", 1 / 100), "top_score_err": ("max",)},
    ("rcnn_fp16", "800x1333"): {"feat_rel_err": ("max",), "proposals_matched": ("min", 1 / 1000), "detections_matched": ("min", 1 / 100), "top_score_err": ("max",)},
    ("mask_rcnn_fp32", None): {"mask_err": ("max",)},
    ("mask_rcnn_fp16", None): {"mask_err": ("max",)},
    ("yolov8n_int8_320", None): {"head_max_abs_err_fp16": ("max",), "head_mean_rel_err_int8": ("max",)},
    # INT8 at detection level (seeded RANDOM weights: the candidates are the tail of heavy-tailed activations, which entropy calibration
    # clips - min-max calibration shows what the int8 kernels themselves cost)
    ("yolov8n_int8_640", "vs_fp32_oracle"): {"matched_iou50": ("min",), "mean_iou": ("min",), "mean_conf_err": ("max",)},
    ("yolov8n_int8_640", "vs_fp16_engine"): {"matched_iou50": ("min",), "mean_iou": ("min",), "mean_conf_err": ("max",)},
    ("yolov8n_int8_640", "minmax_vs_fp32_oracle"): {"matched_iou50": ("min",), "matched_iou90": ("min",), "mean_iou": ("min",), "mean_conf_err": ("max",)},
    ("retinaface_r50_int8", "vs_fp32_oracle"): {"matched_iou50": ("min",), "mean_iou": ("min",), "mean_conf_err": ("max",)},
    ("retinaface_r50_int8", "vs_fp16_engine"): {"matched_iou50": ("min",), "mean_iou": ("min",), "mean_conf_err": ("max",)},
    ("retinaface_r50_int8", "minmax_vs_fp32_oracle"): {"matched_iou50": ("min", 1 / 3830), "mean_iou": ("min",), "mean_conf_err": ("max",)},
    # int8 KERNEL parity: the engine against its own plan interpreted on the CPU at the plan's scales (oracle/lowered_int8.py) - no calibrator in the comparison
    ("yolov8n_int8_640", "entropy2_engine_vs_plan_interpreter"): {"matched_iou90": ("min", 1 / 900), "matched_iou90_reverse": ("min", 1 / 900), "mean_conf_err": ("max",)},
    ("yolov8n_int8_640", "minmax_engine_vs_plan_interpreter"): {"matched_iou90": ("min", 1 / 900), "matched_iou90_reverse": ("min", 1 / 900), "mean_conf_err": ("max",)},
    ("retinaface_r50_int8", "int8_engine_vs_plan_interpreter"): {"matched_iou50": ("min", 1 / 3830), "mean_iou": ("min",), "mean_conf_err": ("max",)},
    ("retinaface_r50_int8", "int8_minmax_engine_vs_plan_interpreter"): {"matched_iou50": ("min", 1 / 3830), "mean_iou": ("min",), "mean_conf_err": ("max",)},
}

# The numbers: written by `python tools/parity_bounds_from_record.py --write profiles/rNN_parity_metrics.jsonl` at 1.4x the worst value of the
# record (shortfall from 1 for "min" metrics, plus the metric's slack), never by hand; tests/test_parity_bounds.py re-checks them.
# BEGIN GENERATED VALUES
VALUES = {
    ('lenet_fp32', None): {'max_abs_err': 2.93e-07},
    ('resnet50_64', 'fp32'): {'max_abs_err': 0.00159},
    ('resnet50_64', 'fp16'): {'max_abs_err': 1.09},
    ('resnet50_224_b32', None): {'max_abs_err': 1.78},
    ('yolov8n_fp32_128', None): {'head_max_abs_err': 1.74e-05},
    ('yolov8n_fp32_640_b32', None): {'head_max_abs_err_vs_fp64': 8.59e-05, 'head_max_abs_err': 0.0001, 'matched_fraction': 0.9989417989417989, 'min_iou': 0.99999791, 'max_conf_err': 5.18e-06},
    ('yolov8n_fp16_640', None): {'cls_logit_max_abs_err': 0.108, 'box_ltrb_max_abs_err': 0.0281, 'matched_fraction': 0.99807, 'min_iou': 0.99553, 'max_conf_err': 0.0142},
    ('yolov8n_fp16_640_fused', None): {'matched_fraction': 0.99807, 'min_iou': 0.99572, 'max_conf_err': 0.0142},
    ('yolov8n_fp16_640_b32', None): {'matched_fraction': 0.99886, 'min_iou': 0.99834, 'max_conf_err': 0.00736},
    ('retinaface_r50_fp16', '256x320'): {'matched_fraction': 0.99976, 'min_iou': 0.9864, 'max_box_err': 1.15, 'max_conf_err': 0.00808},
    ('retinaface_r50_fp16', '1280x1280'): {'matched_fraction': 0.999976, 'min_iou': 0.9885, 'max_box_err': 1.52, 'max_conf_err': 0.0107},
    ('rcnn_fp32', None): {'feat_err': 1.07e-05, 'score_err': 4.59e-06, 'proposals_matched': 0.98, 'detections_matched': 0.95},
    ('rcnn_fp16', '320x416'): {'feat_rel_err': 0.00248, 'proposals_matched': 0.988, 'detections_matched': 0.952, 'top_score_err': 0.000796},
    ('rcnn_fp16', '800x1067'): {'feat_rel_err': 0.00261, 'proposals_matched': 0.9878, 'detections_matched': 0.962, 'top_score_err': 0.000464},
    ('rcnn_fp16', '800x1333'): {'feat_rel_err': 0.00257, 'proposals_matched': 0.9906, 'detections_matched': 0.962, 'top_score_err': 0.000582},
    ('mask_rcnn_fp32', None): {'mask_err': 1.76e-06},
    ('mask_rcnn_fp16', None): {'mask_err': 0.00139},
    ('yolov8n_int8_320', None): {'head_max_abs_err_fp16': 0.0813, 'head_mean_rel_err_int8': 0.0324, 'head_max_err_over_span_int8': 0.221},
    ('yolov8n_int8_640', 'vs_fp32_oracle'): {'matched_iou50': 0.9372, 'mean_iou': 0.9176, 'mean_conf_err': 0.151},
    ('yolov8n_int8_640', 'vs_fp16_engine'): {'matched_iou50': 0.9371, 'mean_iou': 0.918, 'mean_conf_err': 0.151},
    ('yolov8n_int8_640', 'minmax_vs_fp32_oracle'): {'matched_iou50': 0.924, 'matched_iou90': 0.825, 'mean_iou': 0.9473, 'mean_conf_err': 0.0877},
    ('retinaface_r50_int8', 'vs_fp32_oracle'): {'matched_iou50': 0.9725, 'mean_iou': 0.889, 'mean_conf_err': 0.0444},
    ('retinaface_r50_int8', 'vs_fp16_engine'): {'matched_iou50': 0.9725, 'mean_iou': 0.889, 'mean_conf_err': 0.0445},
    ('retinaface_r50_int8', 'minmax_vs_fp32_oracle'): {'matched_iou50': 0.9335, 'mean_iou': 0.871, 'mean_conf_err': 0.053},
    ('yolov8n_int8_640', 'entropy2_engine_vs_plan_interpreter'): {'matched_iou90': 0.9109, 'matched_iou90_reverse': 0.9061, 'mean_conf_err': 0.0547},
    ('yolov8n_int8_640', 'minmax_engine_vs_plan_interpreter'): {'matched_iou90': 0.9259, 'matched_iou90_reverse': 0.9137, 'mean_conf_err': 0.0701},
    ('retinaface_r50_int8', 'int8_engine_vs_plan_interpreter'): {'matched_iou50': 0.99315, 'mean_iou': 0.886, 'mean_conf_err': 0.0483},
    ('retinaface_r50_int8', 'int8_minmax_engine_vs_plan_interpreter'): {'matched_iou50': 0.95, 'mean_iou': 0.88, 'mean_conf_err': 0.0557},
}
# END GENERATED VALUES

# ---- spec-derived ceilings (hard asserts; by hand; see the module docstring) ------------------------------------------------------
EPS16 = 2.0 ** -11   # relative error of ONE round-to-nearest into fp16 (half an ulp)


def fp16_walk(sites, magnitude, safety=1.5):
    """|error| of a value of size `magnitude` that passed `sites` independent fp16 roundings (weights and activations of kFP16: fp16
    storage, fp32 accumulate), accumulated as a random walk, times a safety factor: EPS16 * sqrt(sites) * magnitude * safety.
    YOLOv8n: 133 rounding sites (63 weight tensors + 70 activations), logits up to 16 -> 0.135; tools/fp16_budget.py (a CPU emulation of
    exactly that arithmetic, independent of the GPU product) measures 0.064 and shows it spread over all sites (profiles/r04_fp16_budget.txt)."""
    return safety * EPS16 * math.sqrt(sites) * magnitude


NS_LOGIT, NS_IOU = 1e-4, 1e-3   # the north_star: 1e-4 on logits of O(10) (= 1e-5 of the largest logit), 1e-3 box IoU; met by fp32 builds
FP16_IOU, FP16_MATCH = 1e-2, 5e-3   # the fp16 allowance stated in DESIGN 2: 10x the IoU figure, 0.5 % of oracle candidates may flip at a threshold
CEILINGS = {
    ("lenet_fp32", None): {"max_abs_err": NS_LOGIT / 10},                       # probabilities <= 1
    ("resnet50_64", "fp32"): {"max_abs_err": NS_LOGIT / 10 * 720},               # logits up to 720
    ("resnet50_64", "fp16"): {"max_abs_err": fp16_walk(107, 720)},               # 53 convs + fc: 54 weight + 53 activation sites -> 5.5
    ("resnet50_224_b32", None): {"max_abs_err": fp16_walk(107, 1787)},           # fp16, logits up to 1787 -> 13.5
    ("yolov8n_fp32_128", None): {"head_max_abs_err": NS_LOGIT},
    ("yolov8n_fp32_640_b32", None): {"head_max_abs_err_vs_fp64": NS_LOGIT, "head_max_abs_err": NS_LOGIT, "matched_fraction": 1.0 - 1 / 945, "min_iou": 1 - NS_IOU,
                                     "max_conf_err": NS_LOGIT},   # logits up to 30.  The north_star's 1e-4 against BOTH references - the graph evaluated in double and the fp32
    # oracle (the criterion bench.py's `met` flag uses; round 5 had widened this one by the oracle's own measured distance from the double value, a ceiling
    # derived from a measurement: ADVICE r5).  (A confidence is sigmoid(logit), slope <= 1/4; one candidate of ~900 may sit on the 0.1 threshold.)
    ("yolov8n_fp16_640", None): {"cls_logit_max_abs_err": fp16_walk(133, 16), "box_ltrb_max_abs_err": fp16_walk(133, 16) / 4,   # DFL: expectation over
                                 # softmax(16 logits) in cells, d(expectation)/d(logit) <= 1/4 of the bin span per unit logit for a unimodal side
                                 "matched_fraction": 1 - FP16_MATCH, "min_iou": 1 - FP16_IOU, "max_conf_err": fp16_walk(133, 16) / 4},   # sigmoid' <= 1/4
    ("yolov8n_fp16_640_fused", None): {"matched_fraction": 1 - FP16_MATCH, "min_iou": 1 - FP16_IOU, "max_conf_err": fp16_walk(133, 16) / 4},
    ("yolov8n_fp16_640_b32", None): {"matched_fraction": 1 - FP16_MATCH, "min_iou": 1 - FP16_IOU, "max_conf_err": fp16_walk(133, 16) / 4},
    # RetinaFace: anchors of 16 px at stride 8 - one pixel of box error is IoU 0.88 on the smallest face; boxes = prior + 0.1 * delta * size
    ("retinaface_r50_fp16", "256x320"): {"matched_fraction": 1 - FP16_MATCH, "min_iou": 1 - 2 * FP16_IOU, "max_box_err": 2.0, "max_conf_err": fp16_walk(140, 16) / 4},
    ("retinaface_r50_fp16", "1280x1280"): {"matched_fraction": 1 - FP16_MATCH, "min_iou": 1 - 2 * FP16_IOU, "max_box_err": 2.0, "max_conf_err": fp16_walk(140, 16) / 4},
    ("rcnn_fp32", None): {"feat_err": NS_LOGIT, "score_err": NS_LOGIT, "proposals_matched": 0.98, "detections_matched": 0.95},   # one top-k tie per 50 / 20
    # R-CNN fp16: judged stage by stage; one flipped top-k / NMS decision upstream legitimately changes what follows
    ("rcnn_fp16", "320x416"): {"feat_rel_err": EPS16 * math.sqrt(90) * 1.5, "proposals_matched": 0.98, "detections_matched": 0.93, "top_score_err": fp16_walk(110, 16) / 4},
    ("rcnn_fp16", "800x1067"): {"feat_rel_err": EPS16 * math.sqrt(90) * 1.5, "proposals_matched": 0.98, "detections_matched": 0.93, "top_score_err": fp16_walk(110, 16) / 4},
    ("rcnn_fp16", "800x1333"): {"feat_rel_err": EPS16 * math.sqrt(90) * 1.5, "proposals_matched": 0.98, "detections_matched": 0.93, "top_score_err": fp16_walk(110, 16) / 4},
    ("mask_rcnn_fp32", None): {"mask_err": NS_LOGIT / 10},                        # sigmoid outputs <= 1
    ("mask_rcnn_fp16", None): {"mask_err": fp16_walk(110, 16) / 4},
    # (The detection-level int8 figures move by +-0.03 under changes of a scale in its fourth digit - 0.977 / 0.946 and 0.996 / 0.953 matched
    # for two min-max builds whose thresholds differ by 0.05 % (rounds 3 / 4): the candidates of these random-weight networks are tail events.
    # The 0.90 below leaves that margin under every measured value.)
    # INT8.  One int8 rounding is 1/254 of the tensor's range (vs 2^-11 of the VALUE for fp16), so no logit-level figure is stated for it;
    # the tolerance is at DETECTION level: with either calibrator at least 90 % of the fp32 oracle's candidates must be found again at
    # IoU > 0.5, the head must stay within 10 % mean relative error, and its worst logit error is bounded by the range
    # of the logits themselves (an error as large as the logit range would mean a dead head).
    ("yolov8n_int8_320", None): {"head_max_abs_err_fp16": fp16_walk(133, 16), "head_mean_rel_err_int8": 0.10, "head_max_err_over_span_int8": 0.5},
    # YOLOv8n + entropy calibration on the seeded RANDOM weights.  Rounds 4-5: the plain KL threshold found 81-83 % of the candidates again, ALL of the loss
    # being clipping (tools/int8_budget.py, a CPU emulation with the KL thresholds: the 8-bit grid alone 99.6 %), and the row was asserted at 0.71.  Round 6
    # (VERDICT r5 item 7): the calibration never clips more than 1e-4 of a tensor (TRTX_INT8_CLIP_LIMIT; the KL threshold is raised to that quantile - a rule
    # stated before it was measured: the usual 99.99th percentile) -> 95.5 % at IoU 0.5, 77 % at IoU 0.9 (plain KL: 82.6 % / 48 %; profiles/r06_int8_clip_limit.txt),
    # and the row is asserted at the same 0.90 as every other int8 row.
    ("yolov8n_int8_640", "vs_fp32_oracle"): {"matched_iou50": 0.90, "mean_iou": 0.90, "mean_conf_err": 0.20},
    ("yolov8n_int8_640", "vs_fp16_engine"): {"matched_iou50": 0.90, "mean_iou": 0.90, "mean_conf_err": 0.20},
    ("yolov8n_int8_640", "minmax_vs_fp32_oracle"): {"matched_iou50": 0.90, "matched_iou90": 0.75, "mean_iou": 0.90, "mean_conf_err": 0.10},
    ("retinaface_r50_int8", "vs_fp32_oracle"): {"matched_iou50": 0.90, "mean_iou": 0.80, "mean_conf_err": 0.15},
    ("retinaface_r50_int8", "vs_fp16_engine"): {"matched_iou50": 0.90, "mean_iou": 0.80, "mean_conf_err": 0.15},
    ("retinaface_r50_int8", "minmax_vs_fp32_oracle"): {"matched_iou50": 0.90, "mean_iou": 0.85, "mean_conf_err": 0.10},
    # Kernel parity at MODEL level.  Construct by construct the int8 engine equals its plan interpreted on the CPU bit for bit
    # (tests/test_gpu_int8.py::test_int8_engine_equals_its_plan_interpreted_on_the_cpu - that is the assertion of kernel parity).  Through the 60 requantisations of
    # a whole detector that exactness does not survive: an fp16-ulp difference anywhere (summation order of an fp16 layer chosen by the tactic timing, the stem's
    # v_exp against torch.sigmoid) moves one value across an int8 rounding boundary, one int8 step is 1/254 of a tensor's range, and from there the two evaluations
    # are two samples of the same quantisation noise.  What the model-level rows assert is that the engine is an order of magnitude closer to ITS PLAN than to the
    # fp32 oracle (entropy calibration, same images: 83 % at IoU 0.5 / conf err 0.25 against the oracle; 94 % at IoU 0.9 / 0.03 against the plan): the gap to
    # fp32 is the calibrator's clipping, not the kernels.
    ("yolov8n_int8_640", "entropy2_engine_vs_plan_interpreter"): {"matched_iou90": 0.90, "matched_iou90_reverse": 0.90, "mean_conf_err": 0.06},
    ("yolov8n_int8_640", "minmax_engine_vs_plan_interpreter"): {"matched_iou90": 0.90, "matched_iou90_reverse": 0.90, "mean_conf_err": 0.08},
    ("retinaface_r50_int8", "int8_engine_vs_plan_interpreter"): {"matched_iou50": 0.95, "mean_iou": 0.88, "mean_conf_err": 0.06},
    ("retinaface_r50_int8", "int8_minmax_engine_vs_plan_interpreter"): {"matched_iou50": 0.95, "mean_iou": 0.88, "mean_conf_err": 0.06},
}
SPEC[("yolov8n_int8_320", None)]["head_max_err_over_span_int8"] = ("max",)   # the worst head element was reported and unbounded.  Bounded
# relative to the span of the logits: under entropy calibration a clipped activation moves single logits by a third of the span (measured
# 19 of 55), half the span would be a dead head; the mean error (6 %) is what the 8-bit grid costs

BOUNDS = {k: {m: (spec[0], VALUES[k][m]) + tuple(spec[1:]) for m, spec in ms.items() if k in VALUES and m in VALUES[k]} for k, ms in SPEC.items()}

RECORD = os.path.join("gpurun_out", "parity_metrics.jsonl")


class ParityDrift(UserWarning):
    """a measured value left the band fitted to the committed record while staying inside the spec-derived tolerance"""


def record(test, case=None, **values):
    os.makedirs(os.path.dirname(RECORD), exist_ok=True)
    with open(RECORD, "a") as f:
        f.write(json.dumps(dict(test=test, case=case, **values)) + "\n")


def holds(kind, bound, value):
    return value < bound if kind == "max" else value >= bound


def check(test, case=None, **values):
    """Record `values`, then assert every one that has a bound (all of them reported on failure)."""
    record(test, case, **values)
    spec = SPEC.get((test, case), {})
    missing = [m for m in spec if m not in values]
    assert not missing, f"{test}/{case}: bounded metrics not reported: {missing}"
    # 1. the tolerance: spec-derived ceilings
    ceil = CEILINGS.get((test, case), {})
    over = {m: (values[m], (spec[m][0], ceil[m])) for m in spec if m in ceil and not holds(spec[m][0], ceil[m], values[m])}
    assert not over, f"{test}/{case}: OUTSIDE THE TOLERANCE (value, (kind, spec-derived ceiling)): {over}"
    # 2. the drift alarm: bounds fitted to the committed record
    bounds = BOUNDS.get((test, case), {})
    bad = {m: (values[m], bounds[m][:2]) for m in bounds if not holds(bounds[m][0], bounds[m][1], values[m])}
    if bad:
        # Round 5 (VERDICT r4 Weak 2): a FAILURE again.  Round 4 had demoted this to a warning that no pytest configuration turned into an error,
        # so an fp16 engine that lost 4x of its IoU margin - or an extra rounding slipped into one epilogue - passed green under ceilings 5x looser than what is
        # measured.  The band (1.4x the worst of several recorded runs on different boxes) has held over rounds 3-5; a numeric change that leaves it is either a
        # regression or an intended change whose new runs belong in the record (tools/parity_bounds_from_record.py).  TRTX_PARITY_DRIFT=warn restores the alarm
        # for exploratory builds.
        record("parity_drift", f"{test}/{case}", **{m: v[0] for m, v in bad.items()})
        msg = (f"{test}/{case}: inside the tolerance but outside the bounds fitted to the record (value, (kind, bound)): {bad} - a numeric change; if intended, add "
               "the new runs to the record and regenerate (tools/parity_bounds_from_record.py)")
        if os.environ.get("TRTX_PARITY_DRIFT") == "warn":
            import warnings
            warnings.warn(ParityDrift(msg))
        else:
            raise AssertionError("PARITY DRIFT: " + msg)
