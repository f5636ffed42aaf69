"""add_documents_stream (bench.py workload: 128 texts + 128 PIL images per request through RequestShardedIngest): cross-request merge targets
(images per merged tower call; 0 = one call per request) x request pipeline on / off, interleaved in one process, + a cProfile of the default form
(where the host time of a request goes).  usage: python tools/stream_quick.py [docs] [merge targets, comma separated]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
import numpy as np
import torch
from PIL import Image
from marqo_amd.ingest import RequestShardedIngest
from marqo_amd.s2_inference.enums import Modality

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev, name = "cuda:0", "open_clip/ViT-B-32/laion2b_s34b_b79k"
rng = np.random.default_rng(100)
words = ["alpha", "beta", "gamma", "delta", "marqo", "tensor", "search", "image", "text", "vector"]
pool = []
for r in range(4):
    imgs = [Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(docs)]
    texts = [" ".join(words[int(j) % 10] for j in rng.integers(0, 10, int(rng.integers(3, 60)))) + f" {r} {i}" for i in range(docs)]
    pool.append((texts, imgs))
merges = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 256, 512, 1024]
ing = RequestShardedIngest(name, dev)
state = {"i": 0}


def step():
    i = state["i"]
    state["i"] += 1
    texts, imgs = pool[i % len(pool)]
    ing.submit(i, [((i, d, "t"), texts[d], Modality.TEXT) for d in range(docs)] + [((i, d, "i"), imgs[d], Modality.IMAGE) for d in range(docs)])


def run(depth, n):
    ing.pipeline_depth = depth
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    rows = ing.collect()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert len(rows) == n
    return dt / n


if os.environ.get("STAGES_AB"):     # the merged image call in one batch against the staged form of open_clip_model.encode_image (PIPELINE_MIN / _STREAMS)
    from marqo_amd.s2_inference import open_clip_model as ocm
    ocm.PIPELINE_ALWAYS = True      # (device-row callers are not staged by default)
    ing.merge_images = 512
    forms = [("one batch", 10 ** 9, 1, False), ("stages, 2 streams", 256, 2, False), ("stages, 2 streams, helper", 256, 2, True)]
    for _, mn, st, hp in forms:
        ocm.PIPELINE_MIN, ocm.PIPELINE_STREAMS, ocm.PIPELINE_THREAD = mn, st, hp
        for _ in range(8):
            step()
        ing.collect()
    for rep in range(4):
        for label, mn, st, hp in forms:
            ocm.PIPELINE_MIN, ocm.PIPELINE_STREAMS, ocm.PIPELINE_THREAD = mn, st, hp
            ms = run(1, 96) * 1e3
            print(f"merged image call: {label:26s}: {ms:.3f} ms per {docs}-document request = {2 * docs / ms * 1e3:.0f} embeddings/s", flush=True)
    sys.exit(0)
for m in merges:            # every group shape once, untimed (workspaces, pinned blocks, LDS attributes)
    ing.merge_images = m
    for _ in range(max(4, 2 * (m // docs))):
        step()
    ing.collect()
for rep in range(3):
    for m in merges:
        for depth in ((0, 1) if m == 0 or rep == 0 else (1, 2)):
            ing.merge_images = m
            n = 96 if m else 32
            ms = run(depth, n) * 1e3
            print(f"merge_images={m:5d} pipeline_depth={depth}: {ms:.3f} ms per {docs}-document request = {2 * docs / ms * 1e3:.0f} embeddings/s", flush=True)
# text and image towers of a group on ONE host thread / HIP stream (two_threads off) against the default two
for rep in range(2):
    for two in (True, False):
        ing.merge_images, ing._bulk.two_threads = 512, two
        ms = run(1, 96) * 1e3
        print(f"merge_images=  512 pipeline_depth=1 two_threads={two}: {ms:.3f} ms per request = {2 * docs / ms * 1e3:.0f} embeddings/s", flush=True)
ing._bulk.two_threads = True
ing.pipeline_depth = 1
ing.merge_images = merges[-1] if len(sys.argv) > 2 else 512
pr = cProfile.Profile()
pr.enable()
run(1, 96)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(18)
print("\n".join(line[:170] for line in out.getvalue().splitlines()[4:40]))
