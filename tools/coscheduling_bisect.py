"""bisecting co-scheduling hazards (round 4; profiles/r04_r3_bisect.txt): one engine, three contexts in flight for several rounds, every output (yolov8n:
every head tensor) compared bit for bit with the lone-context run of the same engine; prints how many elements differ and by how much.
    python tools/coscheduling_bisect.py ROUNDS [poison|-] [MODEL B H W]
Another build of the library is selected with TRTX_HIP_LIB (tensorrtx_amd/capi.py); the row-reuse kernel's builds are
    hipcc ... -DTRTX_EXPERIMENTAL_R3 -DR3_VARIANT=k -c kernels/conv_igemm.hip, linked with the other objects of tensorrtx_amd/csrc/build,
and forced onto the layers with TRTX_FORCE_R3=1|2 [TRTX_FORCE_R3_BN=32|64|80|128] TRTX_TUNE=0 TRTX_GROUP_CONVS=0."""
import os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tensorrtx_amd import capi, engine, synth
from util import synth_wts

def main(rounds, poison, model="yolov8n", B=8, H=640, W=640):
    gpu = torch.device("cuda:0")
    path, _ = synth_wts(model)
    opts = dict(mark_heads=1) if model == "yolov8n" else {}
    e = engine.Engine(engine.build_plan(model, path, batch=B, h=H, w=W, fp16=1, aux_streams=0, **opts))
    names = sorted({t["tactic"] for t in e.tactics()})
    print("tactics in the plan:", names)
    n = 3
    if model == "yolov8n":
        xs = [torch.from_numpy(synth.images(B, H, W, seed=40 + k)).to(gpu) for k in range(n)]
    else:
        g = torch.Generator().manual_seed(60)
        xs = [(torch.rand(B, H, W, 3, generator=g) * 255.0).to(gpu) for _ in range(n)]
    watched = [i for i in range(e.nb_bindings) if not e.is_input[i] and (model != "yolov8n" or e.names[i].startswith("head"))]
    outs_of = lambda: {i: torch.zeros(B * int(np.prod(e.dims[i])), dtype=torch.float32, device=gpu) for i in range(e.nb_bindings) if not e.is_input[i]}
    want = []
    for x in xs:
        o = outs_of()
        e.enqueue(B, [x if e.is_input[i] else o[i] for i in range(e.nb_bindings)])
        torch.cuda.synchronize()
        want.append({i: t.cpu() for i, t in o.items()})
    o = outs_of()
    e.enqueue(B, [xs[0] if e.is_input[i] else o[i] for i in range(e.nb_bindings)])
    torch.cuda.synchronize()
    rep = all(torch.equal(o[i].cpu(), want[0][i]) for i in watched)
    ctxs = [e] + [e.create_context() for _ in range(n - 1)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    side = torch.cuda.Stream()
    outs = [outs_of() for _ in range(n)]
    torch.cuda.synchronize()
    for r in range(rounds):
        if poison:
            with torch.cuda.stream(side):
                capi.poison_lds(sync=False)
        for j in range(n):
            k = (j + r) % n
            ctxs[j].enqueue(B, [xs[k] if e.is_input[i] else outs[j][i] for i in range(e.nb_bindings)], stream=streams[j].cuda_stream)
    torch.cuda.synchronize()
    bad = []
    for j in range(n):
        k = (j + rounds - 1) % n
        for i in watched:
            got, ref = outs[j][i].cpu(), want[k][i]
            d = (got - ref).abs()
            nz = int((torch.nan_to_num(d) > 0).sum()) + int(torch.isnan(got).sum())
            if nz:
                bad.append(f"ctx{j} {e.names[i]}: {nz} of {got.numel()} differ, max {float(torch.nan_to_num(d).max()):.4g}, nan {int(torch.isnan(got).sum())}")
    print(f"{model} b{B} {H}x{W} | serial-repeatable {rep} | rounds {rounds} poison {poison} | " + ("IDENTICAL" if not bad else f"{len(bad)} tensors differ"))
    for b in bad[:9]:
        print("   ", b)
    e.close()

if __name__ == "__main__":
    a = sys.argv[1:]
    rounds = int(a[0]) if a else 8
    poison = len(a) > 1 and a[1] == "poison"
    if len(a) > 2:
        main(rounds, poison, a[2], int(a[3]), int(a[4]), int(a[5]))
    else:
        main(rounds, poison)
