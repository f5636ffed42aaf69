"""median wall time of one synchronous 256-image vectorise_ndarray() call from PIL images (the bench's headline_e2e form), quick A/B helper"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
import numpy as np
import torch
from PIL import Image
from marqo_amd.s2_inference import s2_inference as s2
from marqo_amd.s2_inference.enums import Modality

dev, name, n = "cuda:0", "open_clip/ViT-B-32/laion2b_s34b_b79k", 256
rng = np.random.default_rng(0)
pil = [Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(n)]
props = s2.get_model_properties_from_registry(name)
kw = dict(device=dev, modality=Modality.IMAGE, model_properties=props)
for _ in range(4):
    s2.vectorise_ndarray(name, pil, **kw)
ts = []
for _ in range(15):
    t0 = time.perf_counter()
    s2.vectorise_ndarray(name, pil, **kw)
    ts.append(time.perf_counter() - t0)
ts.sort()
print(f"PACK_SLICE={os.environ.get('MARQO_AMD_PACK_SLICE', 'default')} GEMM_PL={os.environ.get('MQ_GEMM_PL', '0')}: median {ts[7] * 1e3:.3f} ms = {n / ts[7]:.0f} emb/s, best {ts[0] * 1e3:.3f} ms")
