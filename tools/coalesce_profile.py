"""Where does a MERGED engine call spend its time?  16 request threads x 4 items through the coalescer at depth 1 (one merged call at a time, so
one cProfile instance can follow whichever thread leads): per-call wall time, items per call, the gap between consecutive engine calls, and
the cumulative profile of the leaders' run(merged).   python tools/coalesce_profile.py [--model open_clip/ViT-B-32/laion2b_s34b_b79k]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
import numpy as np
import torch

from marqo_amd.s2_inference import coalesce
from marqo_amd.s2_inference import s2_inference as s2
from marqo_amd.s2_inference.enums import Modality


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="open_clip/ViT-B-32/laion2b_s34b_b79k")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--items", type=int, default=4)
    ap.add_argument("--calls", type=int, default=60)
    ap.add_argument("--window", default="1000")
    ap.add_argument("--no-profile", action="store_true", help="time stamps only (cProfile itself slows the leader down)")
    args = ap.parse_args()
    os.environ["MARQO_AMD_COALESCE_DEPTH"] = "1"
    os.environ["MARQO_AMD_COALESCE_US"] = args.window
    dev = "cuda:0"
    words = ["alpha", "beta", "gamma", "delta", "marqo", "tensor", "search", "image", "text", "vector"]
    rng = np.random.default_rng(0)
    content = {(t, c): [" ".join(words[int(j)] for j in rng.integers(0, 10, 12)) + f" {t} {c} {i}" for i in range(args.items)]
               for t in range(args.threads) for c in range(args.calls)}
    props = s2.get_model_properties_from_registry(args.model)
    kw = dict(modality=Modality.TEXT)
    for _ in range(3):
        s2.vectorise_ndarray(args.model, content[(0, 0)], model_properties=props, device=dev, **kw)
    pr = cProfile.Profile()
    spans = []
    orig = coalesce.Coalescer.submit

    def submit(self, key, part, run, window, limit):
        def prun(merged):
            t0 = time.perf_counter()
            if not args.no_profile:
                pr.enable()
            try:
                return run(merged)
            finally:
                if not args.no_profile:
                    pr.disable()
                spans.append((t0, time.perf_counter(), len(merged)))
        return orig(self, key, part, prun, window, limit)
    coalesce.Coalescer.submit = submit
    start = threading.Barrier(args.threads + 1)
    lat = []

    def worker(t):
        start.wait()
        for c in range(args.calls):
            t0 = time.perf_counter()
            s2.vectorise_ndarray(args.model, content[(t, c)], model_properties=props, device=dev, **kw)
            lat.append(time.perf_counter() - t0)
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(args.threads)]
    for t in ts:
        t.start()
    torch.cuda.synchronize()
    start.wait()
    t0 = time.perf_counter()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    spans.sort()
    n = args.threads * args.calls * args.items
    durs = np.array([e - s for s, e, _ in spans])
    items = np.array([k for _, _, k in spans])
    gaps = np.array([spans[i + 1][0] - spans[i][1] for i in range(len(spans) - 1)])
    print(f"{args.model}: {n / dt:.0f} embeddings/s; {len(spans)} engine calls, {items.mean():.1f} items per call; run(merged) mean {durs.mean() * 1e3:.3f} ms "
          f"(p50 {np.median(durs) * 1e3:.3f}); gap between calls mean {gaps.mean() * 1e3:.3f} ms (p50 {np.median(gaps) * 1e3:.3f}); "
          f"sum of runs {durs.sum() * 1e3:.1f} ms + gaps {gaps.sum() * 1e3:.1f} ms of {dt * 1e3:.1f} ms wall; caller latency p50 {np.median(lat) * 1e3:.2f} ms")
    # the same merged sizes from ONE thread (no other thread competes for the interpreter)
    coalesce.Coalescer.submit = orig
    os.environ["MARQO_AMD_COALESCE_US"] = "0"
    for k in (args.items, int(items.mean()), args.threads * args.items):
        batch = [x for t in range(args.threads) for x in content[(t, 0)]][:k]
        s2.vectorise_ndarray(args.model, batch, model_properties=props, device=dev, **kw)
        t0 = time.perf_counter()
        for _ in range(50):
            s2.vectorise_ndarray(args.model, batch, model_properties=props, device=dev, **kw)
        print(f"   one thread, {k:3d} items per call: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per call")
    if not args.no_profile:
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(45)
        print(buf.getvalue())


if __name__ == "__main__":
    main()
