import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrtx_amd import capi
dev = torch.device("cuda:0")
B, cin, cout, h = 32, 256, 256, 20
perm = torch.tensor([(7 * i + 3) % B for i in range(B)], device=dev)
g = torch.Generator().manual_seed(cin * 1000 + cout + 1)
w = (torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).numpy()
pk, cp, kp, bn = capi.pack_conv_weights_f16(w, cin_pad=cin)
wp = torch.from_numpy(pk.view(np.int16)).to(dev)
bias = torch.zeros(cp, device=dev)
x = torch.randn(B, h, h, cin, generator=g).half().to(dev)
y1 = capi.conv2d_nhwc_f16(x, wp, bias, cout, 1, 1, 1, 0, "none").reshape(B, h * h, cout)
y2 = capi.conv2d_nhwc_f16(x[perm].contiguous(), wp, bias, cout, 1, 1, 1, 0, "none").reshape(B, h * h, cout)
torch.cuda.synchronize()
ref = (x.reshape(B, h * h, cin).float() @ torch.from_numpy(w.reshape(cout, cin)).to(dev).half().float().t())
d = (y2.float() - y1[perm].float())
idx = d.nonzero()
print("differing elements:", idx.shape[0])
for j, r, c in idx[:40].tolist():
    src = perm[j].item()
    g1 = src * 400 + r; g2 = j * 400 + r
    print(f"img {src} row {r} ch {c}: y1 {y1[src, r, c].item():.6f} (tile {g1 // 128} row-in-tile {g1 % 128}) y2 {y2[j, r, c].item():.6f} (tile {g2 // 128} row-in-tile {g2 % 128}) fp32 ref {ref[src, r, c].item():.7f}")
