export TRTX_CONV_BK32=1
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_engine.py -m gpu -q -x 2>&1 | tail -5
python tools/conv_ab.py wsk 13,14,15,16,30,31,32,33 2>&1 | tail -1
TRTX_CONV_NOWSK=1 python tools/conv_ab.py nowsk 13,14,15,16,30,31,32,33 2>&1 | tail -1
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), d['ms_per_step'], d['roofline']['conv_ms_per_step'])"
