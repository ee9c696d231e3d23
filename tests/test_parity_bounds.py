"""The parity bounds of tests/parity.py against the committed record of what the GPU tests measured (profiles/r03_parity_metrics.jsonl,
several runs): every bound holds on every recorded run, and no bound is looser than 1.5x the worst recorded value.  CPU test."""
import glob
import json
import os

import pytest

from tests import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_parity_metrics.jsonl")))
    rows = []
    with open(files[-1]) as f:           # the latest round's record
        for line in f:
            if line.strip():
                rows.append(json.loads(line))
    return os.path.basename(files[-1]), rows


@pytest.mark.parametrize("key", sorted(parity.BOUNDS, key=str), ids=lambda k: f"{k[0]}-{k[1]}")
def test_bound_is_within_one_and_a_half_times_the_worst_measurement(key):
    name, rows = _records()
    mine = [r for r in rows if r["test"] == key[0] and r.get("case") == key[1]]
    assert len(mine) >= 2, f"{name} holds {len(mine)} runs of {key}: the record needs several (fp16 results move with the build-time tactic choice)"
    for metric, spec in parity.BOUNDS[key].items():
        kind, bound = spec[0], spec[1]
        slack = spec[2] if len(spec) > 2 else 0.0
        vals = [r[metric] for r in mine]
        if kind == "max":
            worst = max(vals)
            assert worst < bound, f"{key} {metric}: recorded {worst} violates the bound {bound}"
            assert bound <= 1.5 * worst + slack, f"{key} {metric}: bound {bound} is looser than 1.5 x the worst recorded value {worst}"
        else:
            worst = min(vals)
            assert worst >= bound, f"{key} {metric}: recorded {worst} violates the bound {bound}"
            # (the generator rounds the bound outward to 3 digits: 1 % on the slack)
            assert (1 - bound) <= 1.5 * (1 - worst) + 1.01 * slack + 1e-12, f"{key} {metric}: bound {bound} is looser than 1.5 x the worst recorded shortfall (worst {worst})"
