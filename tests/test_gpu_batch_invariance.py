"""Size-independent property at BASELINE.json's full sizes: an image's result does not depend on its position in the batch.
Permuting the input images must permute every output bit for bit (same engine, same kernels): this catches tiles that leak across
image borders, per-image plugin state indexed wrongly, and position-dependent rounding in unrolled epilogues."""
import numpy as np
import pytest
import torch

from tensorrtx_amd import engine, synth
from util import synth_wts

pytestmark = pytest.mark.gpu


def _run(e, x, batch, gpu):
    bufs = []
    for i in range(e.nb_bindings):
        if e.is_input[i]:
            bufs.append(x.to(gpu))
        else:
            bufs.append(torch.zeros(batch * int(np.prod(e.dims[i])), dtype=torch.float32, device=gpu))
    e.enqueue(batch, bufs)
    torch.cuda.synchronize()
    return {e.names[i]: bufs[i].cpu().numpy().reshape(batch, -1) for i in range(e.nb_bindings) if not e.is_input[i]}


def _perm(batch):
    return torch.tensor([(3 * i + 1) % batch for i in range(batch)]) if batch % 3 else torch.tensor(list(range(1, batch)) + [0])


def _check(model, batch, h, w, x, gpu, **opts):
    path, _ = synth_wts(model)
    e = engine.Engine(engine.build_plan(model, path, batch=batch, h=h, w=w, fp16=1, **opts))
    p = _perm(batch)
    a, b = _run(e, x, batch, gpu), _run(e, x[p].contiguous(), batch, gpu)
    e.close()
    for name in a:
        assert np.isfinite(a[name]).all(), name
        assert np.array_equal(b[name], a[name][p.numpy()]), f"{model}: output '{name}' depends on the batch position"
    return a


def test_resnet50_224_batch32(gpu):
    """BASELINE configs[1]"""
    x = torch.rand(32, 3, 224, 224, generator=torch.Generator().manual_seed(21))
    out = _check("resnet50", 32, 224, 224, x, gpu)
    assert len(set(out["prob"].argmax(1).tolist())) >= 1


def test_retinaface_r50_1280_batch8(gpu):
    """BASELINE configs[3]: decode plugin output [1 + 67200 * 15] per image"""
    x = (torch.from_numpy(synth.images(8, 1280, 1280, seed=22)) * 255 - 110) / 64
    out = _check("retinaface_r50", 8, 1280, 1280, x, gpu)
    assert (out["prob"][:, 0] > 100).all()


def test_rcnn_r50c4_1333x800_batch4(gpu):
    """BASELINE configs[4]: the whole plugin chain (top-k, NMS, RoIAlign on 4000 RoIs, soft-NMS) per image"""
    g = torch.Generator().manual_seed(23)
    x = torch.rand(4, 800, 1333, 3, generator=g) * 255.0
    out = _check("rcnn_r50c4", 4, 800, 1333, x, gpu)
    assert (out["scores"].max(1) > 0).all()
