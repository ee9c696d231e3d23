"""Generates tests/golden/ref_plugins.npz and tests/golden/ref_host_nms.npz: outputs of the REFERENCE'S OWN code
(oracle/_ref, built from /root/reference by oracle/ref_build.py) on the seeded cases of tests/ref_cases.py.

  host part (CPU, runs anywhere oracle/_ref/libref_host.so exists):
      python tests/golden/make_ref_golden.py host
  plugin part (needs the MI355X: the reference's CUDA plugins compiled by hipcc run on the GPU):
      gpurun -- python tests/golden/make_ref_golden.py plugins gpurun_out/ref_golden
      cp gpurun_out/ref_golden/ref_plugins.npz tests/golden/

The committed fixtures let the oracle be checked against reference-computed results on machines where neither
/root/reference nor oracle/_ref exists.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def host(out_dir):
    import ref_host_cases as hc
    from oracle import ref
    res = {}
    for name, rows in hc.yolov8_cases().items():
        for b, row in enumerate(rows):
            res[f"yolov8/{name}/{b}"] = ref.yolov8_nms(row, 0.5, 0.45)[:, :6]
    for name, rows in hc.retina_cases().items():
        for b, row in enumerate(rows):
            res[f"retina/{name}/{b}"] = ref.retina_nms(row, 0.4)
    for name, rows in hc.yolov5_cases().items():
        for b, row in enumerate(rows):
            res[f"yolov5/{name}/{b}"] = ref.yolov5_nms(row, 0.5, 0.45)[:, :6]
    np.savez_compressed(os.path.join(out_dir, "ref_host_nms.npz"), **res)
    print("wrote", len(res), "arrays")


def plugins(out_dir, only_missing=False):
    """only_missing ("plugins_missing"): keep every array of the committed file and add the cases it does not hold yet"""
    import torch
    import ref_cases as rc
    dev = torch.device("cuda:0")
    res = {}
    if only_missing:
        with np.load(os.path.join(ROOT, "tests", "golden", "ref_plugins.npz")) as z:
            res = {k: z[k] for k in z.files}
    for case, _, _ in rc.all_cases():
        if only_missing and any(k.startswith(case.name + "/") for k in res):
            continue
        outs = case.canon(rc.run_reference(case, dev))
        for k, o in enumerate(outs):
            res[f"{case.name}/{k}"] = np.asarray(o)
        print(case.name, [np.asarray(o).shape for o in outs])
    os.makedirs(out_dir, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, "ref_plugins.npz"), **res)


if __name__ == "__main__":
    what = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden")
    {"host": host, "plugins": plugins, "plugins_missing": lambda o: plugins(o, only_missing=True)}[what](out)
