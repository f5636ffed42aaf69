"""_mq_stage.gather_rgbx (marqo_amd/csrc/py_stage.cpp): a batch of Pillow RGB images -> one staging buffer in one native call.  Host code
only, so everything is checked here on the CPU: bytes identical to np.asarray(img), failures reported by index (never silently skipped),
lazily opened files, images Pillow stores in several blocks, other modes, bad arguments, concurrent callers."""
import io
import threading

import numpy as np
import pytest
from PIL import Image

from marqo_amd import _lib as L
from marqo_amd.engine import preprocess as P


@pytest.fixture(scope="module")
def stage():
    L.build_stage()
    m = L.load_stage()
    if m is None:
        pytest.skip("MARQO_AMD_NATIVE_STAGE=0")
    probe = Image.fromarray(np.zeros((2, 2, 3), dtype=np.uint8))
    if not hasattr(probe, "__arrow_c_array__"):
        pytest.skip("this Pillow has no Arrow export")
    return m


def _layout(sizes):
    npix = np.asarray([h * w for h, w in sizes], dtype=np.int64)
    nbytes = npix * 4
    padded = (nbytes + 255) // 256 * 256
    off = np.zeros(len(sizes), dtype=np.int64)
    off[1:] = np.cumsum(padded)[:-1]
    return off, nbytes, int(padded.sum())


def _check(dst, off, arr):
    h, w, _ = arr.shape
    got = dst[int(off):int(off) + h * w * 4].reshape(h, w, 4)
    assert np.array_equal(got[..., :3], arr)


@pytest.mark.parametrize("threads", [1, 4])
def test_gather_matches_asarray(stage, threads):
    rng = np.random.default_rng(3)
    sizes = [(224, 224), (1, 1), (3, 5), (333, 77), (17, 23), (480, 640), (2, 3)] + [(200 + i, 300 + 2 * i) for i in range(40)]
    arrs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    imgs = [Image.fromarray(a) for a in arrs]
    off, nbytes, total = _layout(sizes)
    dst = np.full(total + 64, 0xAB, dtype=np.uint8)
    failed = stage.gather_rgbx(imgs, dst.ctypes.data, dst.size, off, nbytes, threads)
    assert failed == []
    for a, o in zip(arrs, off):
        _check(dst, o, a)
    assert (dst[total:] == 0xAB).all()                      # nothing written past the layout
    # the images are still usable and unchanged afterwards (the Arrow arrays were released)
    assert np.array_equal(np.asarray(imgs[0]), arrs[0])


def test_lazy_files_multi_block_and_other_modes(stage):
    rng = np.random.default_rng(4)
    a0 = rng.integers(0, 256, (100, 120, 3), dtype=np.uint8)
    b = io.BytesIO()
    Image.fromarray(a0).save(b, "PNG")
    lazy = Image.open(io.BytesIO(b.getvalue()))                        # not decoded yet: the export loads it
    big_arr = rng.integers(0, 256, (2300, 2300, 3), dtype=np.uint8)    # 21 MB of RGBX: more than one of Pillow's 16 MB blocks
    big = Image.fromarray(big_arr)
    grey = Image.fromarray(rng.integers(0, 256, (10, 12), dtype=np.uint8)).convert("L").copy()
    rgba = Image.fromarray(rng.integers(0, 256, (10, 12, 4), dtype=np.uint8), "RGBA").copy()
    a4 = rng.integers(0, 256, (9, 9, 3), dtype=np.uint8)
    imgs = [lazy, big, grey, rgba, Image.fromarray(a4), "not an image"]
    sizes = [(100, 120), (2300, 2300), (10, 12), (10, 12), (9, 9), (1, 1)]
    off, nbytes, total = _layout(sizes)
    dst = np.zeros(total, dtype=np.uint8)
    failed = stage.gather_rgbx(imgs, dst.ctypes.data, dst.size, off, nbytes, 2)
    _check(dst, off[0], a0)
    _check(dst, off[4], a4)
    assert 2 in failed and 5 in failed                                  # mode L is not 4 bytes per pixel; a str has no Arrow interface
    assert set(failed) <= {1, 2, 3, 5}
    if 1 in failed:   # several blocks: the caller's slow route (Rgbx.view) must give the same bytes
        v = P.Rgbx(big).view
        assert v.shape == (2300, 2300, 4) and np.array_equal(v[..., :3], big_arr)
    else:
        _check(dst, off[1], big_arr)
    if 3 not in failed:   # an RGBA image IS 4 bytes per pixel: exported as it is (the engine never sends one: pil_pixels gates on mode RGB)
        assert np.array_equal(dst[int(off[3]):int(off[3]) + 480].reshape(10, 12, 4), np.asarray(rgba))


def test_size_mismatch_is_reported_not_copied(stage):
    img = Image.fromarray(np.full((8, 8, 3), 7, dtype=np.uint8))
    dst = np.zeros(4096, dtype=np.uint8)
    for wrong in (8 * 8 * 4 + 4, 8 * 8 * 3, 0, -4):
        assert stage.gather_rgbx([img], dst.ctypes.data, dst.size, np.zeros(1, dtype=np.int64), np.asarray([wrong], dtype=np.int64), 1) == [0]
        assert not dst.any()
    assert stage.gather_rgbx([img], dst.ctypes.data, dst.size, np.asarray([-256], dtype=np.int64), np.asarray([256], dtype=np.int64), 1) == [0]
    assert stage.gather_rgbx([], 0, 0, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), 1) == []


def test_bad_arguments_raise(stage):
    img = Image.fromarray(np.zeros((2, 2, 3), dtype=np.uint8))
    dst = np.zeros(64, dtype=np.uint8)
    z = np.zeros(1, dtype=np.int64)
    with pytest.raises(TypeError):
        stage.gather_rgbx((img,), dst.ctypes.data, dst.size, z, z, 1)              # images must be a list
    with pytest.raises(ValueError):
        stage.gather_rgbx([img], dst.ctypes.data, dst.size, np.zeros(1, dtype=np.int32), z, 1)
    with pytest.raises(ValueError):
        stage.gather_rgbx([img], dst.ctypes.data, dst.size, np.zeros(2, dtype=np.int64), z, 1)
    with pytest.raises(ValueError):
        stage.gather_rgbx([img], 0, 64, z, np.asarray([16], dtype=np.int64), 1)   # null destination
    with pytest.raises(ValueError):   # a slot that does not fit the destination is the caller's bug: loud, nothing copied
        stage.gather_rgbx([img], dst.ctypes.data, dst.size, np.asarray([56], dtype=np.int64), np.asarray([16], dtype=np.int64), 1)
    with pytest.raises(ValueError):
        stage.gather_rgbx([img], dst.ctypes.data, 8, z, np.asarray([16], dtype=np.int64), 1)
    with pytest.raises((TypeError, ValueError, BufferError)):
        stage.gather_rgbx([img], dst.ctypes.data, dst.size, [0], z, 1)


def test_concurrent_callers_share_images(stage):
    """request threads may stage the SAME PIL objects at the same time (one media download feeding several fields)"""
    rng = np.random.default_rng(5)
    arrs = [rng.integers(0, 256, (224, 224, 3), dtype=np.uint8) for _ in range(48)]
    imgs = [Image.fromarray(a) for a in arrs]
    off, nbytes, total = _layout([(224, 224)] * 48)
    errors = []

    def worker():
        try:
            for _ in range(5):
                dst = np.zeros(total, dtype=np.uint8)
                assert stage.gather_rgbx(imgs, dst.ctypes.data, dst.size, off, nbytes, 4) == []
                for a, o in zip(arrs[::7], off[::7]):
                    _check(dst, o, a)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
    ts = [threading.Thread(target=worker) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors


def test_pil_pixels_defers_the_export_when_the_stager_is_loaded(stage):
    a = np.random.default_rng(6).integers(0, 256, (30, 20, 3), dtype=np.uint8)
    r = P.pil_pixels(Image.fromarray(a))
    assert isinstance(r, P.Rgbx) and r._view is None and r.shape == (30, 20, 3)
    assert np.array_equal(r.view[..., :3], a) and r._view is not None     # everybody else still gets the array view


# ---- the whole pack (layout, job table, both copy routes) on the CPU: the device repack kernel is emulated from its job table -----------
class _FakeStream:
    cuda_stream = 0


def _emulated_unpack(staged_ptr, jobs_off, nx, max_npix, buf_ptr, stream):
    """mq_unpack_rgbx on host memory: jobs = int64 (src_off, dst_off, npix) triples at staged + jobs_off; RGBX -> RGB"""
    import ctypes as C
    jobs = np.ctypeslib.as_array(C.cast(staged_ptr + jobs_off, C.POINTER(C.c_int64)), (nx * 3,)).reshape(nx, 3)
    assert int(jobs[:, 2].max()) == max_npix
    for src, dst, npix in jobs.tolist():
        x = np.ctypeslib.as_array(C.cast(staged_ptr + src, C.POINTER(C.c_uint8)), (npix * 4,)).reshape(npix, 4)
        out = np.ctypeslib.as_array(C.cast(buf_ptr + dst, C.POINTER(C.c_uint8)), (npix * 3,))
        out[:] = x[:, :3].reshape(-1)
    return 0


@pytest.mark.parametrize("native", [True, False])
def test_packed_images_layout_on_host(stage, native, monkeypatch):
    import contextlib
    import torch
    lib = L.load()
    monkeypatch.setattr(lib, "mq_unpack_rgbx", _emulated_unpack, raising=False)
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda d=None: _FakeStream())
    if not native:
        monkeypatch.setattr(L, "load_stage", lambda: None)
    rng = np.random.default_rng(8)
    sizes = [(224, 224), (1, 1), (3, 5), (333, 77), (64, 64), (17, 23), (2, 3)] + [(50 + i, 40 + 2 * i) for i in range(20)]
    arrs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    pils = [Image.fromarray(a) for a in arrs]
    views = [P.pil_pixels(p) for p in pils]
    assert all(isinstance(v, P.Rgbx) and (v._view is None) == native for v in views)
    half = [P.pil_pixels(p) for p in pils]
    for v in half[::2]:
        v.view
    mixed = [P.pil_pixels(pils[i]) if i % 3 == 0 else (arrs[i] if i % 3 == 1 else torch.from_numpy(arrs[i])) for i in range(len(arrs))]
    for batch in (views, arrs, mixed, half, views[:1]):
        p = P.PackedImages(batch, torch.device("cpu"))
        buf = p.buffer.numpy()
        assert p.n == len(batch)
        for a, off in zip(arrs, p.offsets):
            assert np.array_equal(buf[int(off):int(off) + a.size], a.reshape(-1)), a.shape


def test_rgb_sizes_vouches_for_whole_batches_only(stage):
    """the batch fast path of PackedImages (round 4): ONE native scan says whether every item is a loaded Pillow RGB image and fills heights / widths;
    anything else in the list — another mode, a file that is not decoded yet, an array — sends the whole batch down the per-image route"""
    rng = np.random.default_rng(3)
    imgs = [Image.fromarray(rng.integers(0, 256, (20 + i, 30 + 2 * i, 3), dtype=np.uint8)) for i in range(9)]
    h, w = np.zeros(9, np.int32), np.zeros(9, np.int32)
    assert stage.rgb_sizes(imgs, h, w) is True
    assert h.tolist() == [20 + i for i in range(9)] and w.tolist() == [30 + 2 * i for i in range(9)]
    assert P.pil_rgb_sizes(imgs)[0].tolist() == h.tolist()
    buf = io.BytesIO()
    imgs[0].save(buf, format="PNG")
    buf.seek(0)
    lazy = Image.open(buf)                                  # not decoded yet: its core does not exist
    for odd in (imgs[1].convert("L"), imgs[2].convert("RGBA"), lazy, np.zeros((4, 4, 3), np.uint8), "http://x/y.png", None):
        assert stage.rgb_sizes(imgs + [odd], np.zeros(10, np.int32), np.zeros(10, np.int32)) is False
        assert P.pil_rgb_sizes(imgs + [odd]) is None
    assert P.pil_rgb_sizes([]) is None and P.pil_rgb_sizes(tuple(imgs)) is None
    with pytest.raises((BufferError, TypeError, ValueError)):
        stage.rgb_sizes(imgs, np.zeros(9, np.int32).tobytes(), w)      # read-only buffer
    assert stage.rgb_sizes(imgs, np.zeros(3, np.int32), w) is False    # wrong length: refused, nothing written out of bounds


def test_copy_pool_survives_fork_and_many_callers(stage):
    """the copy threads are kept between calls (round 4); a fork()ed child must not wait for workers it does not have"""
    import os
    rng = np.random.default_rng(4)
    arrs = [rng.integers(0, 256, (224, 224, 3), dtype=np.uint8) for _ in range(48)]     # 9.6 MB: above the single-thread cut-off
    imgs = [Image.fromarray(a) for a in arrs]
    off, nbytes, total = _layout([(224, 224)] * 48)
    dst = np.zeros(total, dtype=np.uint8)
    assert stage.gather_rgbx(imgs, dst.ctypes.data, dst.nbytes, off, nbytes, 4) == []
    pid = os.fork()
    if pid == 0:
        d2 = np.zeros(total, dtype=np.uint8)
        ok = stage.gather_rgbx(imgs, d2.ctypes.data, d2.nbytes, off, nbytes, 4) == [] and np.array_equal(d2[int(off[7]):int(off[7]) + 224 * 224 * 4].reshape(224, 224, 4)[..., :3], arrs[7])
        os._exit(0 if ok else 3)
    _, status = os.waitpid(pid, 0)
    assert os.waitstatus_to_exitcode(status) == 0
    errs = []

    def worker():
        d = np.zeros(total, dtype=np.uint8)
        for _ in range(10):
            if stage.gather_rgbx(imgs, d.ctypes.data, d.nbytes, off, nbytes, 4) != [] or not np.array_equal(
                    d[int(off[5]):int(off[5]) + 224 * 224 * 4].reshape(224, 224, 4)[..., :3], arrs[5]):
                errs.append(1)
    ts = [threading.Thread(target=worker) for _ in range(5)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not errs
