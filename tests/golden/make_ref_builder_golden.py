"""Writes tests/golden/ref_builder_plans.json: for every case of tests/ref_builder_cases.py, the SHA-256 (and size) of the plan the
REFERENCE'S OWN builder serialized for the seeded synthetic weights, plus the SHA-256 of that weights file.  Needs oracle/_ref (built by
oracle/ref_build.py where /root/reference exists).  The committed digests keep the product's host builders pinned where oracle/_ref is
absent.  Run from the repository root:  python tests/golden/make_ref_builder_golden.py
"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import ref_builder_cases as rb  # noqa: E402

out = {}
for case in rb.CASES:
    plan = rb.reference_plan(case)
    out[case] = {"plan_sha256": hashlib.sha256(plan).hexdigest(), "plan_bytes": len(plan), "wts_sha256": rb.wts_sha(case)}
    print(case, out[case], flush=True)
with open(rb.GOLDEN, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
    f.write("\n")
