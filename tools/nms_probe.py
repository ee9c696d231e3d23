import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrtx_amd import capi, synth
from oracle import yolo_post as yp
dev = torch.device("cuda:0")
ins = synth.yolo_head_tensors(32, seed=0)
dec = torch.from_numpy(yp.decode_c(ins, 80, 640, 640, [8, 16, 32])).to(dev)
L = capi.lib()
ki = torch.zeros((32, 1000), dtype=torch.int32, device=dev); kc = torch.zeros(32, dtype=torch.int32, device=dev)
kd = torch.zeros((32, 1000, 6), device=dev); dbg = torch.zeros(8, dtype=torch.int64, device=dev)
for _ in range(3):
    L.trtx_yolo_nms_probe(capi._p(dec), 32, 1000, ctypes.c_float(0.5), ctypes.c_float(0.45), capi._p(ki), capi._p(kc), capi._p(kd), capi._p(dbg), capi._stream())
torch.cuda.synchronize()
print("phase ticks (100MHz): load, sort, gather, greedy, compact, n_valid:", dbg.cpu().tolist(), "count", dec[0,0].item())
