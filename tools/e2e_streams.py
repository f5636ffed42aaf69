"""one synchronous 256-image vectorise_ndarray() call from PIL images: the call as one batch, in pipelined stages on the request stream, and with
the stages alternating between HIP streams (open_clip_model.PIPELINE_CHUNK / PIPELINE_STREAMS), interleaved in one process; checks that every
form returns the same bits"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
import numpy as np
import torch
from PIL import Image
from marqo_amd.s2_inference import open_clip_model as ocm
from marqo_amd.s2_inference import s2_inference as s2
from marqo_amd.s2_inference.enums import Modality

dev, name, n = "cuda:0", "open_clip/ViT-B-32/laion2b_s34b_b79k", int(os.environ.get("N_IMAGES", "256"))
rng = np.random.default_rng(0)
pil = [Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(n)]
props = s2.get_model_properties_from_registry(name)
kw = dict(device=dev, modality=Modality.IMAGE, model_properties=props)
def stages_of(split):
    def f(m):
        out, a = [], 0
        for k in split:
            out.append((a, min(a + k, m)))
            a += k
        assert a == m, (split, m)
        return out
    return f


# (stage sizes, streams, towers enqueued by the helper thread)
if n <= 384:
    h = n // 2
    configs = [((n,), 1, True), ((h, n - h), 2, True), ((h, n - h), 2, False), ((h, n - h), 1, True), ((h, n - h), 1, False), ((64, n - 64), 2, True)]
else:
    q = n // 4
    configs = [((n,), 1, True), ((n // 2, n // 2), 2, True), ((n // 2, n // 2), 2, False), ((q, q, q, q), 2, True), ((q, q, q, q), 2, False), ((q, n - q), 2, True)]
ref, res = None, {c: [] for c in configs}
ocm.PIPELINE_CHUNK = 1      # every list call takes the staged path
for rep in range(int(os.environ.get('REPS', '3'))):
    for c in configs:
        ocm._pipeline_stages, ocm.PIPELINE_STREAMS, ocm.PIPELINE_THREAD = stages_of(c[0]), c[1], c[2]
        for _ in range(3):
            out = s2.vectorise_ndarray(name, pil, **kw)
        if ref is None:
            ref = out.copy()
        if rep == 0:
            print(f"{c}: max |diff| vs the one-batch call {np.abs(out - ref).max():.2e}")
        ts = []
        for _ in range(11):
            t0 = time.perf_counter()
            s2.vectorise_ndarray(name, pil, **kw)
            ts.append(time.perf_counter() - t0)
        res[c].append(sorted(ts)[5])
for c in configs:
    m = sorted(res[c])[len(res[c]) // 2]
    print(f"stages={c[0]} streams={c[1]} helper_thread={c[2]}: median of medians {m * 1e3:.3f} ms = {n / m:.0f} emb/s   ({', '.join(f'{t * 1e3:.3f}' for t in res[c])})")
