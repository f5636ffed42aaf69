"""K10 / K11 parity: the HIP resize / crop / chunk kernels (through the C ABI) must be BIT-EXACT against
the CPU oracle (oracle/resample.c, itself pinned to Pillow in test_oracle_resample.py)."""
import numpy as np
import pytest
import torch

from oracle import preprocess as OP

pytestmark = pytest.mark.gpu

SIZES = [(224, 224), (225, 224), (224, 225), (300, 200), (200, 300), (480, 640), (640, 480), (1000, 37), (37, 1000),
         (64, 64), (17, 23), (1, 1), (1, 500), (1201, 1600), (2048, 1536), (223, 223), (449, 449)]


def _imgs(sizes, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]


@pytest.fixture(scope="module")
def pre():
    from marqo_amd.engine.preprocess import ImagePreprocessor
    return ImagePreprocessor("cuda:0", 224)


def test_resize_crop_bit_exact(pre):
    imgs = _imgs(SIZES)
    out = pre.resize_crop_u8(imgs).cpu().numpy()
    assert out.shape == (len(imgs), 224, 224, 3)
    for i, im in enumerate(imgs):
        ref = OP.clip_resize_crop_u8(im, 224, backend="c")
        assert np.array_equal(out[i], ref), f"image {i} size {im.shape}: max |d| = {np.abs(out[i].astype(int) - ref).max()}"


def test_resize_crop_matches_pillow_directly(pre):
    imgs = _imgs([(333, 517), (800, 600)], seed=3)
    out = pre.resize_crop_u8(imgs).cpu().numpy()
    for i, im in enumerate(imgs):
        assert np.array_equal(out[i], OP.clip_resize_crop_u8(im, 224, backend="pil"))


def test_other_model_resolution():
    from marqo_amd.engine.preprocess import ImagePreprocessor
    p = ImagePreprocessor("cuda:0", 336)
    imgs = _imgs([(500, 400), (336, 336), (100, 700)], seed=5)
    out = p.resize_crop_u8(imgs).cpu().numpy()
    for i, im in enumerate(imgs):
        assert np.array_equal(out[i], OP.clip_resize_crop_u8(im, 336, backend="c"))


def test_empty_and_exact_size(pre):
    assert pre.resize_crop_u8([]).shape == (0, 224, 224, 3)
    im = _imgs([(224, 224)], seed=9)
    assert np.array_equal(pre.resize_crop_u8(im).cpu().numpy()[0], im[0])  # Pillow returns a copy when the size matches


@pytest.mark.parametrize("hn,wn,overlap", [(3, 3, False), (3, 3, True), (2, 4, False), (1, 1, False), (5, 7, True)])
def test_chunk_grid_bit_exact(pre, hn, wn, overlap):
    imgs = _imgs([(480, 640), (240, 240), (100, 333), (900, 50)], seed=11)
    out, boxes = pre.chunk_grid_u8(imgs, hn, wn, overlap)
    out = out.cpu().numpy()
    for i, im in enumerate(imgs):
        patches, bbs = OP.chunk_image_simple(im, hn, wn, overlap, backend="c")
        assert boxes.shape[1] == len(patches)
        for k, (patch, bb) in enumerate(zip(patches, bbs)):
            ref = OP.clip_resize_crop_u8(patch, 224, backend="c")
            assert np.array_equal(out[i * len(patches) + k], ref), (i, k, patch.shape)
            assert np.allclose(boxes[i, k], np.asarray(bb, dtype=np.float32), rtol=1e-6)


def test_chunk_count_known_answers(pre):
    """reference tests/processing/test_image_chunking.py: 3x3 simple -> 1 + 9 patches."""
    out, boxes = pre.chunk_grid_u8(_imgs([(300, 400)]), 3, 3, False)
    assert out.shape[0] == 10 and boxes.shape == (1, 10, 4)
    assert np.allclose(boxes[0, 0], [0, 0, 400, 300])


def test_plain_resize_bit_exact(pre):
    imgs = _imgs([(480, 640), (100, 333), (240, 240), (900, 50)], seed=13)
    out = pre.resize_u8(imgs, 240, 240).cpu().numpy()
    for i, im in enumerate(imgs):
        assert np.array_equal(out[i], OP.resize_u8(im, 240, 240, backend="c"))
    out = pre.resize_u8(imgs[:2], 77, 123).cpu().numpy()
    for i, im in enumerate(imgs[:2]):
        assert np.array_equal(out[i], OP.resize_u8(im, 123, 77, backend="c"))


def test_to_tensor_normalize_bit_exact(pre):
    u8 = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (3, 224, 224, 3), dtype=np.uint8))
    out = pre.to_tensor_normalize(u8.cuda()).cpu().numpy()
    ref = np.stack([OP.to_tensor_normalize(u.numpy()) for u in u8])
    assert out.shape == (3, 3, 224, 224)
    # same IEEE fp32 ops in the same order: (b/255 - mean)/std
    assert np.abs(out - ref).max() <= 2.4e-7 * np.abs(ref).max()


def test_preprocess_then_tower_equals_u8_path(pre):
    """`.preprocess`-style fp32 tensors and the fused uint8 path must give the same embeddings."""
    from marqo_amd.engine import archs, towers
    from oracle import towers as O
    arch = archs.VitArch(224, 32, 128, 2, 2, 256, 64)
    cfg = O.VitConfig(224, 32, 128, 2, 2, 256, 64)
    sd = O.synthetic_vit_state_dict(cfg, seed=0)
    tower = towers.VitTower(arch, sd, "cuda:0")
    u8 = pre.resize_crop_u8(_imgs([(300, 500), (640, 480)], seed=4))
    a = tower.encode_u8(u8)
    b = tower.encode_f32(pre.to_tensor_normalize(u8))
    assert torch.allclose(a, b, atol=2e-3)


@pytest.mark.parametrize("interp,filt", [("bilinear", OP.FILTER_BILINEAR), ("bicubic", OP.FILTER_BICUBIC)])
def test_squash_resize_filters_bit_exact_vs_pillow(pre, interp, filt):
    """the 'squash' preprocessors: SigLIP (bicubic) and CLIPA (BILINEAR, open_clip _apcfg, selected by the reference at
    open_clip_model.py:87-97) — PIL.Image.resize((S, S), filter) reproduced bit for bit, up- and down-scaling, extreme aspect ratios"""
    from PIL import Image
    imgs = _imgs([(224, 224), (300, 200), (64, 64), (17, 23), (1, 500), (1201, 1600), (449, 223), (37, 1000)], seed=21)
    out = pre.resize_u8(imgs, 224, 224, interpolation=interp).cpu().numpy()
    res = {"bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC}[interp]
    for i, im in enumerate(imgs):
        ref = np.asarray(Image.fromarray(im).resize((224, 224), res))
        assert np.array_equal(out[i], ref), f"{interp} image {i} {im.shape}: max |d| = {np.abs(out[i].astype(int) - ref.astype(int)).max()}"
        assert np.array_equal(OP.resize_u8(im, 224, 224, backend="c", filt=filt), ref)       # and the C oracle agrees with Pillow
    out2 = pre.resize_u8(imgs[:3], 96, 160, interpolation=interp).cpu().numpy()               # non-square target
    for i in range(3):
        assert np.array_equal(out2[i], np.asarray(Image.fromarray(imgs[i]).resize((160, 96), res)))


def test_packed_images_rgbx_staging_and_threaded_copies(pre):
    """PIL images travel as Pillow's in-memory RGBX bytes (zero-copy Arrow view) and are repacked to RGB on the device (mq_unpack_rgbx);
    arrays / CPU tensors are copied by the pack threads: the packed device buffer must hold every image's RGB bytes at its offset, for
    all-PIL, all-array and mixed batches, odd pixel counts included."""
    from PIL import Image
    from marqo_amd.engine import preprocess as P
    sizes = [(224, 224), (1, 1), (3, 5), (333, 77), (64, 64), (17, 23), (480, 640), (2, 3)] + [(50 + i, 40 + 2 * i) for i in range(24)]
    arrs = _imgs(sizes, seed=9)
    pils = [Image.fromarray(a) for a in arrs]
    views = [P.pil_pixels(p) for p in pils]
    if not all(isinstance(v, P.Rgbx) for v in views):
        pytest.skip("this Pillow / pyarrow pair has no Arrow export")
    mixed = [views[i] if i % 3 == 0 else (arrs[i] if i % 3 == 1 else torch.from_numpy(arrs[i])) for i in range(len(arrs))]
    # containers whose view was already exported (the pyarrow route) next to fresh ones (the native stager's route, when it is built)
    half = [P.pil_pixels(p) for p in pils]
    for v in half[::2]:
        assert v.view.shape[2] == 4
    for batch in (views, arrs, mixed, [v for v in views[:3]], half):
        p = P.PackedImages(batch, torch.device("cuda:0"))
        buf = p.buffer.cpu().numpy()
        for a, off in zip(arrs, p.offsets):
            assert np.array_equal(buf[int(off):int(off) + a.size], a.reshape(-1)), a.shape
    # and through the resize: PIL route == array route == oracle
    out_pil = pre.resize_crop_u8(views).cpu().numpy()
    out_arr = pre.resize_crop_u8(arrs).cpu().numpy()
    assert np.array_equal(out_pil, out_arr)
    assert np.array_equal(out_arr[3], OP.clip_resize_crop_u8(arrs[3], 224, backend="c"))
    # an image Pillow stores in several blocks cannot be exported zero-copy: it takes the copying route inside the same pack
    big = _imgs([(2300, 2300), (30, 40)], seed=11)
    p = P.PackedImages([P.pil_pixels(Image.fromarray(a)) for a in big], torch.device("cuda:0"))
    buf = p.buffer.cpu().numpy()
    for a, off in zip(big, p.offsets):
        assert np.array_equal(buf[int(off):int(off) + a.size], a.reshape(-1)), a.shape
    # device tensors of one size are stacked, of several sizes copied one by one: same packed bytes
    same = [torch.from_numpy(a).cuda() for a in _imgs([(64, 64)] * 5, seed=2)]
    p = P.PackedImages(same, torch.device("cuda:0"))
    assert torch.equal(p.buffer.reshape(5, 64, 64, 3), torch.stack(same))


# ---- image modes: the device resize follows what Pillow does for the source image's mode -------------------------------------------
def _mode_images(seed=21):
    """PIL images of the modes Image.resize treats specially, with sizes that exercise down- / up-scaling, both crop axes and the
    no-resize case (shorter side already 224)"""
    from PIL import Image
    rng = np.random.default_rng(seed)
    out = []
    for k, (h, w) in enumerate([(300, 400), (224, 224), (100, 90), (640, 224), (511, 333), (224, 500)]):
        a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        a[: h // 3, :, 3] = 0                      # a fully transparent band (arbitrary colour bytes under it)
        a[h // 3: h // 2, :, 3] = 255              # an opaque band
        out.append(("RGBA", Image.fromarray(a, "RGBA")))
        out.append(("LA", Image.fromarray(np.ascontiguousarray(a[..., [0, 3]]), "LA")))
        p = Image.fromarray(rng.integers(0, 256, (h, w), dtype=np.uint8), "P")
        p.putpalette([int(v) for v in rng.integers(0, 256, 768)])
        if k % 2:
            p.info["transparency"] = 3
        out.append(("P", p))
        out.append(("1", Image.fromarray(rng.integers(0, 2, (h, w), dtype=np.uint8) * 255).convert("1")))
        out.append(("L", Image.fromarray(rng.integers(0, 256, (h, w), dtype=np.uint8), "L")))
        out.append(("RGB", Image.fromarray(np.ascontiguousarray(a[..., :3]), "RGB")))
    opaque = rng.integers(0, 256, (260, 300, 4), dtype=np.uint8)
    opaque[..., 3] = 255
    out.append(("RGBA-opaque", Image.fromarray(opaque, "RGBA")))
    return out


def test_image_modes_clip_transform_is_bit_exact_vs_pillow(pre):
    """translucent RGBA / LA (premultiplied round trip), palette / bilevel (forced NEAREST), L and RGB in ONE mixed batch: every image
    equals the reference's own transform run by Pillow on the image in its own mode"""
    from marqo_amd.engine import preprocess as P
    imgs = _mode_images()
    px = [P.pil_pixels(im) for _, im in imgs]
    kinds = {m: type(x).__name__ for (m, _), x in zip(imgs, px)}
    assert kinds["RGBA"] == "Rgba" and kinds["LA"] == "Rgba" and kinds["P"] == "NearestRgb" and kinds["1"] == "NearestRgb"
    assert kinds["RGBA-opaque"] in ("ndarray",) and kinds["L"] == "ndarray"
    out = pre.resize_crop_u8(px).cpu().numpy()
    for k, (mode, im) in enumerate(imgs):
        ref = OP.clip_resize_crop_pil_image(im, 224)
        assert np.array_equal(out[k], ref), f"image {k} mode {mode} size {im.size}: max |d| = {np.abs(out[k].astype(int) - ref).max()}"
    # and the flatten-first treatment this replaces really differs for translucent / palette sources (the test is not vacuous)
    flat = pre.resize_crop_u8([np.asarray(im.convert("RGB")) for _, im in imgs[:3]]).cpu().numpy()
    assert not np.array_equal(flat[0], out[0]) and not np.array_equal(flat[2], out[2])


def test_image_modes_squash_is_bit_exact_vs_pillow(pre):
    from marqo_amd.engine import preprocess as P
    imgs = _mode_images(seed=22)
    px = [P.pil_pixels(im) for _, im in imgs]
    for interpolation in ("bicubic", "bilinear"):
        out = pre.resize_u8(px, 224, 224, interpolation).cpu().numpy()
        for k, (mode, im) in enumerate(imgs):
            ref = OP.squash_pil_image(im, 224, 224, bilinear=interpolation == "bilinear")
            assert np.array_equal(out[k], ref), f"{interpolation}: image {k} mode {mode} size {im.size}"
    out = pre.resize_u8(px[:6], 240, 200).cpu().numpy()
    for k, (mode, im) in enumerate(imgs[:6]):
        assert np.array_equal(out[k], OP.squash_pil_image(im, 240, 200)), (k, mode)


def test_mode_call_agrees_with_the_rgb_entry_points(pre):
    """mq_resize_mode_u8 in MQ_IMG_RGB mode == mq_clip_resize_crop_u8 / mq_resize_filter_u8 (one planner, one pair of kernels)"""
    import ctypes as C
    from marqo_amd import _lib as L
    from marqo_amd.engine.preprocess import PackedImages
    lib = L.load()
    imgs = _imgs([(333, 517), (224, 224), (90, 700)], seed=31)
    dev = torch.device("cuda:0")
    p = PackedImages(imgs, dev)
    s = torch.cuda.current_stream().cuda_stream
    for crop in (1, 0):
        need = lib.mq_resize_mode_workspace_bytes(p.heights.ctypes.data, p.widths.ctypes.data, p.n, 224, 224, 3, crop, L.MQ_IMG_RGB)
        ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        got = torch.empty(p.n, 224, 224, 3, dtype=torch.uint8, device=dev)
        L.check(lib.mq_resize_mode_u8(p.buffer.data_ptr(), p.offsets.ctypes.data, p.heights.ctypes.data, p.widths.ctypes.data, p.n, 224, 224, 3,
                                      crop, L.MQ_IMG_RGB, got.data_ptr(), ws.data_ptr(), ws.numel(), s))
        want = pre.resize_crop_u8(imgs) if crop else pre.resize_u8(imgs, 224, 224)
        assert torch.equal(got, want)
    assert lib.mq_resize_mode_u8(p.buffer.data_ptr(), p.offsets.ctypes.data, p.heights.ctypes.data, p.widths.ctypes.data, p.n, 224, 200, 3, 1, 0,
                                 got.data_ptr(), ws.data_ptr(), ws.numel(), s) == -1      # crop needs a square
    assert lib.mq_resize_mode_u8(p.buffer.data_ptr(), p.offsets.ctypes.data, p.heights.ctypes.data, p.widths.ctypes.data, p.n, 224, 224, 3, 1, 7,
                                 got.data_ptr(), ws.data_ptr(), ws.numel(), s) == -1      # unknown mode
