"""1x1 conv = GEMM with IDENTICAL input rows: every output row must be bitwise equal.  Prints the rows that differ from row 0."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrtx_amd import capi
dev = torch.device("cuda:0")
for cin, cout, hw, B in ((256, 256, 20, 32), (256, 256, 16, 1), (256, 128, 16, 1), (256, 64, 16, 1), (64, 256, 16, 1), (256, 256, 16, 8)):
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).numpy()
    pk, cp, kp, bn = capi.pack_conv_weights_f16(w, cin_pad=cin)
    wp = torch.from_numpy(pk.view(np.int16)).to(dev)
    bias = torch.zeros(cp, device=dev)
    row = torch.randn(cin, generator=g).half()
    x = row.reshape(1, 1, 1, cin).expand(B, hw, hw, cin).contiguous().to(dev)
    y = capi.conv2d_nhwc_f16(x, wp, bias, cout, 1, 1, 1, 0, "none")
    torch.cuda.synchronize()
    y = y.reshape(-1, cout).float().cpu()
    bad = (y != y[0]).any(1).nonzero().flatten().tolist()
    print(f"{cin}->{cout} M={y.shape[0]} bn={bn}: rows differing from row 0: {len(bad)}", bad[:24], "channels:", sorted(set((y[bad] != y[0]).nonzero()[:, 1].tolist()))[:16] if bad else "")
