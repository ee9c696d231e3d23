cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for res in 7 0 7 0; do
  rm -f gpurun_out/parity_metrics.jsonl
  TRTX_CONV_RES=$res TRTX_PARITY_DRIFT=warn timeout 600 python -m pytest tests/test_gpu_rcnn.py -m gpu -q -k "stagewise" 2>&1 | tail -1
  python - $res <<'P'
import json, sys
for l in open("gpurun_out/parity_metrics.jsonl"):
    r = json.loads(l)
    if "rcnn_fp16" in r.get("key", r.get("name", "")) or "rcnn" in json.dumps(r)[:200]:
        print("res", sys.argv[1], json.dumps(r)[:400])
P
done
