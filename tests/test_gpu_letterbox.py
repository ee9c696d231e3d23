"""f2: letterbox pre-processing (reference yolov8/src/preprocess.cu:7-127).  Product HIP kernel == C oracle == the reference's own
warpaffine kernel (oracle/_ref/libref_yolov8_post.so, compiled unmodified by hipcc), bit for bit."""
import ctypes
import os

import numpy as np
import pytest

from oracle import preproc as opre
from oracle import ref
from tensorrtx_amd import preproc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIZES = [((480, 640), (640, 640)), ((640, 480), (640, 640)), ((37, 53), (64, 64)), ((1080, 1920), (640, 640)), ((100, 100), (96, 127)),
         ((33, 200), (640, 384))]  # (src h, w) -> (dst h, w): landscape, portrait, up-scaling, odd destination width


def _img(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)


def test_letterbox_matrix_matches_oracle():
    for (sh, sw), (dh, dw) in SIZES + [((3000, 3000), (640, 640)), ((1, 1), (32, 32))]:
        assert np.array_equal(preproc.letterbox_matrix(sw, sh, dw, dh), opre.letterbox_matrix(sw, sh, dw, dh))


def test_oracle_letterbox_equals_reference_kernel_output():
    """committed output of the reference's own kernel (run on the MI355X) for a small image"""
    p = os.path.join(GOLD, "ref_letterbox.npz")
    if not os.path.exists(p):
        pytest.skip("tests/golden/ref_letterbox.npz not generated yet")
    z = np.load(p)
    assert np.array_equal(opre.letterbox(z["img"], 64, 64), z["out"])


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", SIZES)
def test_letterbox_bit_exact_vs_oracle(gpu, src, dst):
    import torch
    img = _img(src[0], src[1], seed=src[0] + dst[1])
    got = preproc.letterbox_batch([torch.from_numpy(img).to(gpu)], dst[1], dst[0]).cpu().numpy()[0]
    want = opre.letterbox(img, dst[1], dst[0])
    assert np.array_equal(got, want)
    assert (want == np.float32(128) / np.float32(255)).any() or src[0] * dst[1] == src[1] * dst[0]  # a border exists unless aspect ratios match


@pytest.mark.gpu
def test_letterbox_mixed_batch_one_launch_and_host_fed_ring(gpu):
    import torch
    imgs = [_img(h, w, seed=h) for (h, w), _ in SIZES[:5]] * 3      # 15 images of five sizes
    want = np.stack([opre.letterbox(im, 320, 256) for im in imgs])
    got = preproc.letterbox_batch([torch.from_numpy(im).to(gpu) for im in imgs], 320, 256).cpu().numpy()
    assert np.array_equal(got, want)
    preproc.preprocess_init(1920 * 1080, ring_depth=6)
    try:
        out = torch.empty((5, 3, 256, 320), dtype=torch.float32, device=gpu)
        for rep in range(4):  # the ring wraps around: slots are reused only after their warp has finished
            chunk = imgs[rep * 3:rep * 3 + 5]
            preproc.batch_preprocess(chunk, 320, 256, out)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), np.stack([opre.letterbox(im, 320, 256) for im in chunk]))
    finally:
        preproc.preprocess_destroy()


@pytest.mark.gpu
def test_letterbox_equals_the_reference_kernel(gpu):
    """the reference's cuda_preprocess (memcpy to its pinned buffer, H2D, warpaffine_kernel) vs trtx_letterbox_batch"""
    import torch
    if not ref.available("libref_yolov8_post.so"):
        pytest.skip("oracle/_ref/libref_yolov8_post.so not present")
    L = ref.family_lib("yolov8_post")
    L.ref_yolov8_preprocess_init(1920 * 1080)
    try:
        for (sh, sw), (dh, dw) in SIZES:
            img = _img(sh, sw, seed=7 + sh)
            want = torch.zeros((3, dh, dw), dtype=torch.float32, device=gpu)
            L.ref_yolov8_preprocess(img.ctypes.data_as(ctypes.c_void_p), sw, sh, ctypes.c_void_p(want.data_ptr()), dw, dh,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            got = preproc.letterbox_batch([torch.from_numpy(img).to(gpu)], dw, dh)[0]
            assert torch.equal(got, want), ((sh, sw), (dh, dw))
            assert np.array_equal(want.cpu().numpy(), opre.letterbox(img, dw, dh))
        if os.environ.get("TRTX_WRITE_GOLDEN"):
            img = _img(37, 53, seed=99)
            want = torch.zeros((3, 64, 64), dtype=torch.float32, device=gpu)
            L.ref_yolov8_preprocess(img.ctypes.data_as(ctypes.c_void_p), 53, 37, ctypes.c_void_p(want.data_ptr()), 64, 64,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez_compressed("gpurun_out/ref_letterbox.npz", img=img, out=want.cpu().numpy())
    finally:
        L.ref_yolov8_preprocess_destroy()


@pytest.mark.gpu
def test_enqueue_frames_is_letterbox_then_enqueue_bit_for_bit(gpu):
    """f2 as specified: the letterbox fused into the stem (trtx_context_enqueue_frames: the first layer samples the uint8 HWC BGR frames
    itself, kernels/conv_stem.hip) against the two-step path the reference's demo takes (cuda_batch_preprocess, then enqueue;
    yolov8_det.cpp:146-160) - frames of mixed sizes and aspect ratios (upscaled, downscaled, portrait, already at network size), every
    engine output compared as bit patterns."""
    import torch
    from tensorrtx_amd import engine, preproc
    from util import synth_wts
    path, _ = synth_wts("yolov8n")
    B, S = 6, 640
    plan = engine.build_plan("yolov8n", path, batch=B, h=S, w=S, fp16=1, mark_heads=1)
    rng = np.random.default_rng(5)
    sizes = [(480, 640), (1080, 1920), (640, 640), (333, 517), (900, 400), (97, 1201)]   # (h, w)
    frames = []
    for h, w in sizes:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        img[h // 4: h // 2, w // 4: w // 2] = rng.integers(0, 256, size=3, dtype=np.uint8)   # a flat patch: exact bilinear plateaus
        frames.append(torch.from_numpy(img).to(gpu))
    e = engine.Engine(plan)
    try:
        def outputs():
            return [None if e.is_input[i] else torch.full((B * int(np.prod(e.dims[i])),), float("nan"), dtype=torch.float32, device=gpu) for i in range(e.nb_bindings)]
        two = outputs()
        x = preproc.letterbox_batch(frames, S, S)
        e.enqueue(B, [x if t is None else t for t in two])
        one = outputs()
        e.enqueue_frames(B, frames, one)
        torch.cuda.synchronize()
        for i in range(e.nb_bindings):
            if e.is_input[i]:
                continue
            a, b = one[i].view(torch.int32), two[i].view(torch.int32)
            assert torch.equal(a, b), (e.names[i], int((a != b).sum()))
        assert two[e.names.index("output")].reshape(B, -1)[:, 0].sum() > 0      # (noise frames: some images have no candidate at all)
        # a batch smaller than max_batch, through a second execution context
        ctx = e.create_context()
        one3, two3 = outputs(), outputs()
        ctx.enqueue(3, [preproc.letterbox_batch(frames[3:], S, S) if t is None else t for t in two3])
        ctx.enqueue_frames(3, frames[3:], one3)
        torch.cuda.synchronize()
        k = e.names.index("output")
        n = 3 * int(np.prod(e.dims[k]))
        assert torch.equal(one3[k][:n].view(torch.int32), two3[k][:n].view(torch.int32))
    finally:
        e.close()


@pytest.mark.gpu
def test_enqueue_frames_rejects_an_fp32_engine_before_anything_is_enqueued(gpu):
    """fp32 engines have a stem op since round 5 but no fused-letterbox form of it - trtx_context_enqueue_frames says UNSUPPORTED up front
    (it used to pass the validation and fail inside the plan's execution with lanes and events already set up); the engine stays usable."""
    import torch
    from tensorrtx_amd import engine, preproc
    from util import synth_wts
    path, _ = synth_wts("yolov8n")
    plan = engine.build_plan("yolov8n", path, batch=1, h=160, w=160, fp16=0)
    e = engine.Engine(plan)
    try:
        frame = torch.randint(0, 256, (120, 200, 3), dtype=torch.uint8, device=gpu)
        outs = [None if e.is_input[i] else torch.zeros(int(np.prod(e.dims[i])), dtype=torch.float32, device=gpu) for i in range(e.nb_bindings)]
        with pytest.raises(Exception):
            e.enqueue_frames(1, [frame], outs)
        x = preproc.letterbox_batch([frame], 160, 160)
        e.enqueue(1, [x if t is None else t for t in outs])      # the two-step path still runs
        torch.cuda.synchronize()
    finally:
        e.close()
