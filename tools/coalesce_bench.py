"""Unmodified-Marqo traffic: many request threads, each calling vectorise() with a handful of items (PER_DOCUMENT: one call per document and
field, add_documents_handler.py:264-290; up to 8 indexing + 8 search threads, api/configs.py:27-28) — with and without the opt-in
cross-request coalescer (MARQO_AMD_COALESCE_US).   python tools/coalesce_bench.py [--threads 16] [--items 4] [--calls 60]"""
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

from marqo_amd.s2_inference import coalesce
from marqo_amd.s2_inference import s2_inference as s2
from marqo_amd.s2_inference.enums import Modality


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--items", type=int, default=4)
    ap.add_argument("--calls", type=int, default=60)
    ap.add_argument("--depth", default="2")
    ap.add_argument("--windows", default="0,200", help="MARQO_AMD_COALESCE_US values to time")
    ap.add_argument("--switch-interval", type=float, default=0.0, help="sys.setswitchinterval (seconds; 0 = leave the interpreter's 5 ms)")
    ap.add_argument("--only", default="", help="substring of the model names to run")
    args = ap.parse_args()
    if args.switch_interval > 0:
        sys.setswitchinterval(args.switch_interval)
    dev = "cuda:0"
    os.environ["MARQO_AMD_COALESCE_DEPTH"] = args.depth
    words = ["alpha", "beta", "gamma", "delta", "marqo", "tensor", "search", "image", "text", "vector"]
    rng = np.random.default_rng(0)

    def texts(t, c):
        return [" ".join(words[int(j)] for j in rng.integers(0, 10, 12)) + f" {t} {c} {i}" for i in range(args.items)]
    for name, kw in (("open_clip/ViT-B-32/laion2b_s34b_b79k", dict(modality=Modality.TEXT)), ("hf/e5-base-v2", dict(modality=Modality.TEXT))):
        if args.only and args.only not in name:
            continue
        props = s2.get_model_properties_from_registry(name)
        content = {(t, c): texts(t, c) for t in range(args.threads) for c in range(args.calls)}
        s2.vectorise_ndarray(name, content[(0, 0)], model_properties=props, device=dev, **kw)      # load
        ref = {k: s2.vectorise_ndarray(name, v, model_properties=props, device=dev, **kw) for k, v in list(content.items())[:8]}
        for window in args.windows.split(","):
            os.environ["MARQO_AMD_COALESCE_US"] = window
            before = dict(coalesce.get_coalescer().stats)
            out, lat = {}, []
            start = threading.Barrier(args.threads + 1)

            def worker(t):
                start.wait()
                for c in range(args.calls):
                    t0 = time.perf_counter()
                    out[(t, c)] = s2.vectorise_ndarray(name, content[(t, c)], model_properties=props, device=dev, **kw)
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
            st = coalesce.get_coalescer().stats
            n = args.threads * args.calls * args.items
            worst = max(float((1 - (out[k] * ref[k]).sum(-1) / (np.linalg.norm(out[k], axis=-1) * np.linalg.norm(ref[k], axis=-1))).max()) for k in ref)
            lat.sort()
            print(f"{name} {args.threads} threads x {args.calls} calls x {args.items} items, MARQO_AMD_COALESCE_US={window} depth {args.depth}: {n / dt:9.0f} embeddings/s, "
                  f"call latency p50 {lat[len(lat) // 2] * 1e3:.2f} ms p95 {lat[int(len(lat) * 0.95)] * 1e3:.2f} ms; engine calls "
                  f"{st['engine_calls'] - before['engine_calls']} for {st['calls'] - before['calls']} coalesced-path calls; max 1-cos vs the lone call {worst:.1e}", flush=True)
        # the serial rate: one thread, the same calls
        os.environ["MARQO_AMD_COALESCE_US"] = "0"
        t0 = time.perf_counter()
        for c in range(args.calls):
            s2.vectorise_ndarray(name, content[(0, c)], model_properties=props, device=dev, **kw)
        dt = time.perf_counter() - t0
        print(f"{name} ONE thread x {args.calls} calls x {args.items} items (serial): {args.calls * args.items / dt:9.0f} embeddings/s", flush=True)


if __name__ == "__main__":
    main()
