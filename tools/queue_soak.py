"""soak: text (CLIP + BERT) and image small calls at once, 3 x 12 threads, ~40 s; every row checked against the lone call"""
import os, sys, threading, time
sys.path.insert(0, os.getcwd())
os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
import numpy as np, torch
from PIL import Image
from marqo_amd.s2_inference import s2_inference as s2
from marqo_amd.s2_inference.enums import Modality, AvailableModelsKey
dev = "cuda:0"
clip, e5 = "open_clip/ViT-B-32/laion2b_s34b_b79k", "hf/e5-base-v2"
pc, pe = s2.get_model_properties_from_registry(clip), s2.get_model_properties_from_registry(e5)
texts = [f"document number {i} about topic {i % 17} with some more words {'x ' * (i % 23)}" for i in range(200)]
kc, ke = dict(device=dev, modality=Modality.TEXT, model_properties=pc), dict(device=dev, modality=Modality.TEXT, model_properties=pe)
ki = dict(device=dev, modality=Modality.IMAGE, model_properties=pc)
lone_c = np.concatenate([s2.vectorise_ndarray(clip, [t], **kc) for t in texts])
lone_e = np.concatenate([s2.vectorise_ndarray(e5, [t], **ke) for t in texts])
model, pre = s2.load_multimodal_model_and_get_preprocessors(clip, pc, dev)
rng = np.random.default_rng(0)
views = [pre["image"](Image.fromarray(rng.integers(0, 256, (200 + i, 230 - i, 3), dtype=np.uint8))) for i in range(48)]
lone_i = np.concatenate([s2.vectorise_ndarray(clip, [v], **ki) for v in views])
def cos(a, b):
    return float((1 - (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))).max())
stop, errs, counts = time.time() + float(os.environ.get("SOAK_S", "40")), [], [0, 0, 0]
def worker(kind, t):
    r = np.random.default_rng(1000 * kind + t)
    try:
        while time.time() < stop:
            n = int(r.integers(1, 7))
            if kind == 0:
                pick = r.integers(0, len(texts), n); out = s2.vectorise_ndarray(clip, [texts[i] for i in pick], **kc); ref = lone_c[pick]
            elif kind == 1:
                pick = r.integers(0, len(texts), n); out = s2.vectorise_ndarray(e5, [texts[i] for i in pick], **ke); ref = lone_e[pick]
            else:
                pick = r.integers(0, len(views), n); out = s2.vectorise_ndarray(clip, [views[i] for i in pick], **ki); ref = lone_i[pick]
            if out.shape != ref.shape or cos(out, ref) > 2e-4:
                errs.append((kind, t, pick.tolist(), cos(out, ref)))
            counts[kind] += 1
    except BaseException as e:
        errs.append((kind, t, repr(e)))
ts = [threading.Thread(target=worker, args=(k, t)) for k in range(3) for t in range(12)]
t0 = time.time()
for t in ts: t.start()
for t in ts: t.join(300)
dt = time.time() - t0
m = s2.get_available_models()
enc = m[s2._create_model_cache_key(clip, dev, pc)][AvailableModelsKey.model]
print(f"soak {dt:.1f} s: requests clip-text {counts[0]}, e5 {counts[1]}, images {counts[2]}; errors {len(errs)} {errs[:3]}; alive {sum(t.is_alive() for t in ts)}")
print("clip text queue", enc.text.queue_stats().get(True)); print("image queue", enc.vision.queue_stats().get(True))
