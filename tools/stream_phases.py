"""where the host time of one add_documents_stream request goes (perf_counter wrappers, no cProfile), pipelined form.  usage: python tools/stream_phases.py"""
import os, sys, time
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
import numpy as np
import torch
from PIL import Image
from marqo_amd import ingest as ING
from marqo_amd.engine import preprocess as P
from marqo_amd.s2_inference import open_clip_model as ocm
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


docs, dev, name = 128, "cuda:0", "open_clip/ViT-B-32/laion2b_s34b_b79k"
rng = np.random.default_rng(100)
words = ["alpha", "beta", "gamma", "delta", "marqo", "tensor", "search", "image", "text", "vector"]
pool = []
for r in range(4):
    imgs = [Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(docs)]
    texts = [" ".join(words[int(j) % 10] for j in rng.integers(0, 10, int(rng.integers(3, 60)))) + f" {r} {i}" for i in range(docs)]
    pool.append((texts, imgs))
ing = ING.RequestShardedIngest(name, dev)
state = {"i": 0}


def step():
    i = state["i"]
    state["i"] += 1
    texts, imgs = pool[i % len(pool)]
    t0 = time.perf_counter()
    items = [((i, d, "t"), texts[d], Modality.TEXT) for d in range(docs)] + [((i, d, "i"), imgs[d], Modality.IMAGE) for d in range(docs)]
    T["build the request's item list (caller)"] += time.perf_counter() - t0
    ing.submit(i, items)


for _ in range(6):
    step()
ing.collect()
ING.RequestShardedIngest.submit = timed("submit (total)", ING.RequestShardedIngest.submit)
ING.BulkVectoriser.add = timed("  BulkVectoriser.add x 256", ING.BulkVectoriser.add)
ING.BulkVectoriser.flush_async = timed("  flush_async (tokenise / pack / enqueue)", ING.BulkVectoriser.flush_async)
ocm.OPEN_CLIP.encode_text = timed("    encode_text", ocm.OPEN_CLIP.encode_text)
ocm.OPEN_CLIP.encode_image = timed("    encode_image", ocm.OPEN_CLIP.encode_image)
ocm.OPEN_CLIP._preprocess_images = timed("      _preprocess_images", ocm.OPEN_CLIP._preprocess_images)
P.PackedImages.__init__ = timed("        PackedImages.__init__", P.PackedImages.__init__)
ING.RequestShardedIngest._resolve = timed("  _resolve (previous request: D2H + filing rows)", ING.RequestShardedIngest._resolve)
torch.Tensor.cpu = timed("    Tensor.cpu", torch.Tensor.cpu)
from marqo_amd.engine import towers as TW
TW._TextTowerBase._call_text = timed("      _call_text (op enqueue)", TW._TextTowerBase._call_text)
from marqo_amd.s2_inference import s2_inference as s2
tok = s2._available_models[next(iter(s2._available_models))]["model"].tokenizer
type(tok).__call__ = timed("      tokenizer", type(tok).__call__)
for depth in (1, 0):
    ing.pipeline_depth = depth
    T.clear()
    n = 40
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    ing.collect()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"==== pipeline_depth={depth}: {(t1 - t0) / n * 1e3:.3f} ms per request in the submit loop, collect {(t2 - t1) * 1e3:.2f} ms for {n} requests")
    for k, v in T.items():
        print(f"{k:60s} {v / n * 1e3:7.3f} ms per request")
