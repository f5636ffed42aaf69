#!/usr/bin/env python
"""Where does a 256-image vectorise_ndarray() call spend its host time?  cProfile of the product API on the three input kinds of
bench.py's e2e_vectorise (PIL images, uint8 arrays, preprocessed device tensors), next to the wall time per call."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
import numpy as np
import torch
from PIL import Image

from marqo_amd.s2_inference import s2_inference as s2
from marqo_amd.s2_inference.enums import Modality


def main():
    dev, name, n = "cuda:0", "open_clip/ViT-B-32/laion2b_s34b_b79k", 256
    rng = np.random.default_rng(0)
    arrs = [rng.integers(0, 256, (224, 224, 3), dtype=np.uint8) for _ in range(n)]
    pil = [Image.fromarray(a) for a in arrs]
    props = s2.get_model_properties_from_registry(name)
    model, pre = s2.load_multimodal_model_and_get_preprocessors(name, props, dev)
    dev_tensors = [pre["image"](p) for p in pil]
    kw = dict(device=dev, modality=Modality.IMAGE, model_properties=props)
    for label, content in (("PIL", pil), ("u8 arrays", arrs), ("device tensors", dev_tensors)):
        for _ in range(3):
            s2.vectorise_ndarray(name, content, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            s2.vectorise_ndarray(name, content, **kw)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(10):
            s2.vectorise_ndarray(name, content, **kw)
        torch.cuda.synchronize()
        pr.disable()
        sio = io.StringIO()
        pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(14)
        print(f"==== {label}: {ms:.3f} ms per {n}-image call = {n / ms * 1e3:.0f} embeddings/s (un-profiled)")
        print("\n".join(l[:150] for l in sio.getvalue().splitlines()[6:26]))


def phases(threads: int):
    """wall time per phase of a call, per thread, with `threads` concurrent callers (wrappers around the phases' entry points)"""
    import collections
    import threading
    from marqo_amd.engine import preprocess as P
    from marqo_amd.engine import towers as T
    acc = collections.defaultdict(float)
    lock = threading.Lock()

    def wrap(obj, name, label):
        fn = getattr(obj, name)

        def inner(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                with lock:
                    acc[label] += time.perf_counter() - t0
        setattr(obj, name, inner)
    wrap(P.PackedImages, "__init__", "pack+H2D enqueue")
    wrap(P.ImagePreprocessor, "resize_crop_u8", "resize (incl. pack)")
    wrap(T.VitTower, "encode_u8", "tower enqueue (u8)")
    wrap(T.VitTower, "encode_f32", "tower enqueue (f32)")
    from marqo_amd.s2_inference import open_clip_model as M
    wrap(M.OPEN_CLIP, "_convert_output", "D2H wait")
    wrap(M.OPEN_CLIP, "encode_image", "encode_image total")
    dev, name, n = "cuda:0", "open_clip/ViT-B-32/laion2b_s34b_b79k", 256
    rng = np.random.default_rng(0)
    arrs = [rng.integers(0, 256, (224, 224, 3), dtype=np.uint8) for _ in range(n)]
    pil = [Image.fromarray(a) for a in arrs]
    props = s2.get_model_properties_from_registry(name)
    kw = dict(device=dev, modality=Modality.IMAGE, model_properties=props)
    for label, content in (("u8 arrays", arrs), ("PIL", pil)):
        s2.vectorise_ndarray(name, content, **kw)
        reps = 8

        def worker():
            for _ in range(reps):
                s2.vectorise_ndarray(name, content, **kw)
        for k in range(2):   # the second pass is the measured one (pinned blocks cached, graphs captured)
            acc.clear()
            ts = [threading.Thread(target=worker) for _ in range(threads)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        calls = reps * threads
        print(f"==== {label}, {threads} callers: {n * calls / wall:.0f} embeddings/s; wall {wall / reps * 1e3:.2f} ms per round of {threads} calls")
        for kname, v in sorted(acc.items(), key=lambda kv: -kv[1]):
            print(f"   {kname:24s} {v / calls * 1e3:8.3f} ms per call")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--threads":
        phases(int(sys.argv[2]))
    else:
        main()
