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


if __name__ == "__main__":
    main()
