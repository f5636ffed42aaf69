"""16 request threads, each calling vectorise_ndarray() with a few images at a time (PER_DOCUMENT add_documents: one image field per call) — PIL images and
the device tensors `.preprocess` returns; MARQO_AMD_COALESCE_US=0 for the un-merged rows.   python tools/small_image_calls.py [--threads 16] [--items 1,4]"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
import numpy as np
import torch
from PIL import Image

from marqo_amd.s2_inference import coalesce
from marqo_amd.s2_inference import s2_inference as s2
from marqo_amd.s2_inference.enums import Modality


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--items", default="1,4")
    ap.add_argument("--calls", type=int, default=40)
    a = ap.parse_args()
    dev, name = "cuda:0", "open_clip/ViT-B-32/laion2b_s34b_b79k"
    props = s2.get_model_properties_from_registry(name)
    kw = dict(device=dev, modality=Modality.IMAGE, model_properties=props)
    rng = np.random.default_rng(0)
    pool = [Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(64)]
    model, pre = s2.load_multimodal_model_and_get_preprocessors(name, props, dev)
    s2.vectorise_ndarray(name, pool[:2], **kw)
    for k in (1, 4):      # the single-item graphs are captured here, before any thread runs beside the capture
        s2.vectorise_ndarray(name, pool[:k], **kw)
        s2.vectorise_ndarray(name, [pre["image"](p) for p in pool[:k]], **kw)
    print(f"# MARQO_AMD_COALESCE_US={os.environ.get('MARQO_AMD_COALESCE_US', '(default)')}; {a.threads} threads x {a.calls} calls", flush=True)
    for form in ("pil", "device_tensors"):
        for items in [int(v) for v in a.items.split(",")]:
            lat, errs = [], []
            start = threading.Barrier(a.threads + 1)

            def worker(t):
                try:
                    mine = [[pool[(t * 7 + c * 3 + i) % len(pool)] for i in range(items)] for c in range(a.calls)]
                    if form == "device_tensors":
                        mine = [[pre["image"](p) for p in batch] for batch in mine]
                    s2.vectorise_ndarray(name, mine[0], **kw)     # (synchronous: host rows come back; NO device-wide synchronise here — on ROCm one
                    start.wait()                                   # issued while another thread captures a hipGraph invalidates that capture)
                    for batch in mine:
                        t0 = time.perf_counter()
                        s2.vectorise_ndarray(name, batch, **kw)
                        lat.append(time.perf_counter() - t0)
                except BaseException as e:  # noqa: BLE001
                    errs.append(e)
                    start.abort()
            ts = [threading.Thread(target=worker, args=(t,)) for t in range(a.threads)]
            for t in ts:
                t.start()
            before = dict(coalesce.get_coalescer().stats)
            start.wait()
            t0 = time.perf_counter()
            for t in ts:
                t.join()
            dt = time.perf_counter() - t0
            if errs:
                raise errs[0]
            st = coalesce.get_coalescer().stats
            lat.sort()
            n = a.threads * a.calls * items
            one = []
            batch = [pool[i] for i in range(items)] if form == "pil" else [pre["image"](pool[i]) for i in range(items)]
            for _ in range(3):
                s2.vectorise_ndarray(name, batch, **kw)
            t0 = time.perf_counter()
            for _ in range(a.calls):
                s2.vectorise_ndarray(name, batch, **kw)
            alone = (time.perf_counter() - t0) / a.calls
            print(f"{form:14s} {items} per call: {n / dt:8.0f} embeddings/s ({n / items / dt:7.0f} requests/s), latency p50 {lat[len(lat) // 2] * 1e3:.2f} p95 {lat[int(len(lat) * .95)] * 1e3:.2f} ms; "
                  f"coalescer {st['engine_calls'] - before['engine_calls']} engine calls for {st['calls'] - before['calls']} calls; one thread alone {alone * 1e3:.2f} ms per call", flush=True)


if __name__ == "__main__":
    main()
