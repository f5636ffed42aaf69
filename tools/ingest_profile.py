#!/usr/bin/env python
"""Wall time per phase of a bulk-ingest step (bench.py --workload add_documents_mixed: 128 strings + 128 PIL images through BulkVectoriser),
measured with perf_counter wrappers around the phases' entry points (no cProfile: its per-call overhead distorts the per-image loops).
usage: python tools/ingest_profile.py [docs]"""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
import numpy as np
import torch
from PIL import Image

from marqo_amd import _lib as L
from marqo_amd.engine import preprocess as P
from marqo_amd.ingest import BulkVectoriser
from marqo_amd.s2_inference import open_clip_model as ocm
from marqo_amd.s2_inference import s2_inference as s2
from marqo_amd.s2_inference.enums import Modality

T = defaultdict(float)


def timed(label, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            T[label] += time.perf_counter() - t0
    return w


def main():
    docs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev, name = "cuda:0", "open_clip/ViT-B-32/laion2b_s34b_b79k"
    rng = np.random.default_rng(7)
    imgs = [Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(docs)]
    words = ["alpha", "beta", "gamma", "delta", "marqo", "tensor", "search", "image", "text", "vector"]
    texts = [" ".join(words[int(j) % 10] for j in rng.integers(0, 10, int(rng.integers(3, 60)))) + f" {i}" for i in range(docs)]
    bv = BulkVectoriser(name, dev)

    def step():
        for i in range(docs):
            bv.add((i, "t"), texts[i])
            bv.add((i, "i"), imgs[i], Modality.IMAGE)
        return bv.flush()

    for _ in range(3):
        step()
    lib = L.load()
    # wrappers (installed after the warm-up so that one-time work is not counted)
    P.PackedImages.__init__ = timed("  PackedImages.__init__ (pack + H2D enqueue + unpack launch)", P.PackedImages.__init__)
    gather, unpack = lib.mq_host_gather, lib.mq_unpack_rgbx
    lib.mq_host_gather = timed("    mq_host_gather (memcpy threads into the pinned buffer)", gather)
    lib.mq_unpack_rgbx = timed("    mq_unpack_rgbx (launch)", unpack)
    ocm.pil_to_pixels = timed("  pil_to_pixels x n (Pillow export)", ocm.pil_to_pixels)
    ocm.OPEN_CLIP._preprocess_images = timed(" _preprocess_images", ocm.OPEN_CLIP._preprocess_images)
    ocm.OPEN_CLIP.encode_image = timed("encode_image (total)", ocm.OPEN_CLIP.encode_image)
    ocm.OPEN_CLIP.encode_text = timed("encode_text (total)", ocm.OPEN_CLIP.encode_text)
    bv._run_pending = timed("BulkVectoriser._run_pending", bv._run_pending)
    from marqo_amd.engine import towers as TW
    TW._pack = timed("  towers._pack (host ids -> packed ids + cu_seqlens)", TW._pack)
    TW._TextTowerBase._call_text = timed("  _call_text (op enqueue)", TW._TextTowerBase._call_text)
    TW._TowerBase._workspace = timed("  _workspace", TW._TowerBase._workspace)
    torch.Tensor.cpu = timed("Tensor.cpu (D2H + wait for the GPU)", torch.Tensor.cpu)
    torch.zeros = timed("torch.zeros (all)", torch.zeros)
    torch.empty = timed("torch.empty (all, incl. pinned staging)", torch.empty)
    tok = s2._available_models[next(iter(s2._available_models))]["model"].tokenizer
    type(tok).__call__ = timed("  tokenizer", type(tok).__call__)
    if os.environ.get("INGEST_TORCH_THREADS"):
        torch.set_num_threads(int(os.environ["INGEST_TORCH_THREADS"]))
    for threads in (P.PACK_THREADS, 1, P.PACK_THREADS):
        P.PACK_THREADS = threads
        T.clear()
        reps = 20
        torch.cuda.synchronize()
        c0 = time.process_time()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        cpu = (time.process_time() - c0) / reps * 1e3
        print(f"==== {docs} documents (text + image), PACK_THREADS={threads}, torch threads {torch.get_num_threads()}: {wall:.2f} ms per step = "
              f"{2 * docs / wall * 1e3:.0f} embeddings/s; process CPU time {cpu:.1f} ms per step")
        for k, v in T.items():
            print(f"{v / reps * 1e3:9.3f} ms  {k}")
    print("host: os.cpu_count() =", os.cpu_count(), " sched_getaffinity =", len(os.sched_getaffinity(0)), " torch threads =", torch.get_num_threads())
    try:
        print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
    except OSError as e:
        print("cgroup cpu.max:", e)


if __name__ == "__main__":
    main()
