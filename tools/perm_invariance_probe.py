"""Which conv shapes give batch-position-dependent results?  y(x[perm]) must equal y(x)[perm] bit for bit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrtx_amd import capi
dev = torch.device("cuda:0")
B = 32
perm = torch.tensor([(7 * i + 3) % B for i in range(B)], device=dev)
shapes = [(16,32,3,2,320),(32,32,1,1,160),(16,16,3,1,160),(48,32,1,1,160),(32,64,3,2,160),(64,64,1,1,80),(32,32,3,1,80),(128,64,1,1,80),
          (64,128,3,2,80),(128,128,1,1,40),(64,64,3,1,40),(256,128,1,1,40),(128,256,3,2,40),(256,256,1,1,20),(128,128,3,1,20),(384,256,1,1,20),
          (512,256,1,1,20),(384,128,1,1,40),(192,128,1,1,40),(192,64,1,1,80),(96,64,1,1,80),(64,64,3,2,80),(128,128,3,2,40),(64,64,3,1,80),
          (64,80,3,1,80),(80,80,3,1,80),(80,80,1,1,80),(128,64,3,1,40),(128,80,3,1,40),(80,80,3,1,40),(256,64,3,1,20),(64,64,3,1,20),(256,80,3,1,20),(80,80,3,1,20)]
for cin, cout, k, s, h in shapes:
    g = torch.Generator().manual_seed(cin * 1000 + cout + k)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).numpy()
    pk, cp, kp, bn = capi.pack_conv_weights_f16(w, cin_pad=cin)
    wp = torch.from_numpy(pk.view(np.int16)).to(dev)
    bias = torch.zeros(cp, device=dev)
    x = torch.randn(B, h, h, cin, generator=g).half().to(dev)
    y1 = capi.conv2d_nhwc_f16(x, wp, bias, cout, k, k, s, k // 2, "silu")
    y2 = capi.conv2d_nhwc_f16(x[perm].contiguous(), wp, bias, cout, k, k, s, k // 2, "silu")
    torch.cuda.synchronize()
    d = (y2.float() - y1[perm].float()).abs()
    print(f"{cin:4d}->{cout:4d} k{k} s{s} {h:4d}^2  max diff {d.max().item():.6f}  images differing {(d.reshape(B, -1).max(1).values > 0).sum().item()}")
