"""The native request queue (mq_queue_*, csrc/queue.hip) under the reference's serving load: T request threads, each handing over a few texts at a time.
Three measurements per model, in one process:
  raw     the threads call TextQueue.encode() with ready token ids (the C ABI's own ceiling: everything between hand-over and wake-up is native);
  direct  the same requests, every thread calling the tower itself (tower.encode_ids on its own stream, no merging);
  product vectorise_ndarray() from the threads — tokeniser, validation, cache key and output conversion included — with the queue
          (MARQO_AMD_NATIVE_QUEUE=1, default) or, in a second run of this tool with MARQO_AMD_NATIVE_QUEUE=0, with the Python coalescer.
python tools/queue_bench.py [--threads 16] [--items 1,4] [--calls 80] [--only ViT-B-32]"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
os.environ.pop("MARQO_AMD_COALESCE_US", None)
import numpy as np
import torch

from marqo_amd.engine import native_queue as NQ
from marqo_amd.engine.towers import request_stream
from marqo_amd.s2_inference import coalesce
from marqo_amd.s2_inference import s2_inference as s2
from marqo_amd.s2_inference.enums import AvailableModelsKey, Modality


def run_threads(n_threads, calls, fn):
    lat, errs = [], []
    start = threading.Barrier(n_threads + 1)

    def worker(t):
        try:
            start.wait()
            for c in range(calls):
                t0 = time.perf_counter()
                fn(t, c)
                lat.append(time.perf_counter() - t0)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for t in ts:
        t.start()
    torch.cuda.synchronize()
    start.wait()
    t0 = time.perf_counter()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    if errs:
        raise errs[0]
    lat.sort()
    return dt, lat[len(lat) // 2] * 1e3, lat[int(len(lat) * 0.95)] * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--items", default="1,4")
    ap.add_argument("--calls", type=int, default=80)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = "cuda:0"
    words = ["alpha", "beta", "gamma", "delta", "marqo", "tensor", "search", "image", "text", "vector"]
    rng = np.random.default_rng(0)
    print(f"# MARQO_AMD_NATIVE_QUEUE={'1' if NQ.ENABLED else '0'} seqs<={NQ.MAX_SEQS} depth={NQ.DEPTH} window={NQ.WINDOW_US}us; {args.threads} threads x {args.calls} calls", flush=True)
    for name in ("open_clip/ViT-B-32/laion2b_s34b_b79k", "hf/e5-base-v2"):
        if args.only and args.only not in name:
            continue
        props = s2.get_model_properties_from_registry(name)
        kw = dict(model_properties=props, device=dev, modality=Modality.TEXT)
        s2.vectorise_ndarray(name, ["load"], **kw)
        model = s2.get_available_models()[s2._create_model_cache_key(name, dev, props)][AvailableModelsKey.model]
        clip = hasattr(model, "text")
        tower = model.text if clip else model._model
        for items in [int(v) for v in args.items.split(",")]:
            content = {(t, c): [" ".join(words[int(j)] for j in rng.integers(0, 10, 12)) + f" {t} {c} {i}" for i in range(items)]
                       for t in range(args.threads) for c in range(args.calls)}
            n = args.threads * args.calls * items
            # ---- token ids of every request, as the loaders make them
            ids = {}
            for k, v in content.items():
                if clip:
                    a = np.asarray(model.tokenizer(v)).astype(np.int64)
                    ln = a.argmax(axis=1) + 1
                else:
                    tok = model._tokenizer(v, max_length=model.model_properties.tokens)
                    a, ln = tok["input_ids"].astype(np.int64), tok["attention_mask"].sum(axis=1)
                keep = np.arange(a.shape[1])[None, :] < ln[:, None]
                ids[k] = (a, ln, np.ascontiguousarray(a[keep], dtype=np.int32), np.ascontiguousarray(ln, dtype=np.int32))
            if NQ.ENABLED:
                q = tower._queue(True, clip)
                before = q.stats()
                dt, p50, p95 = run_threads(args.threads, args.calls, lambda t, c: q.encode(ids[(t, c)][2], ids[(t, c)][3]))
                st = q.stats()
                calls = st["calls"] - before["calls"]
                print(f"{name} {items} items/call raw queue    : {n / dt:9.0f} embeddings/s ({n / items / dt:7.0f} requests/s), latency p50 {p50:.2f} p95 {p95:.2f} ms; "
                      f"{calls} tower calls for {st['requests'] - before['requests']} requests ({(st['sequences'] - before['sequences']) / max(calls, 1):.1f} sequences each, "
                      f"largest {st['max_call_sequences']})", flush=True)

            def direct(t, c):
                a, ln, _, _ = ids[(t, c)]
                with request_stream(tower.device, device_output=True):      # (device rows: the tower launches eagerly on this thread's stream)
                    if clip:
                        out = tower.encode_ids(torch.from_numpy(a), normalize=True)
                    else:
                        out = tower.encode_ids(torch.from_numpy(a), torch.from_numpy((np.arange(a.shape[1])[None, :] < ln[:, None]).astype(np.int64)), normalize=True)
                return out.cpu()
            dt, p50, p95 = run_threads(args.threads, args.calls, direct)
            print(f"{name} {items} items/call direct tower : {n / dt:9.0f} embeddings/s ({n / items / dt:7.0f} requests/s), latency p50 {p50:.2f} p95 {p95:.2f} ms", flush=True)
            cb = dict(coalesce.get_coalescer().stats)
            qb = tower.queue_stats().get(True, {"calls": 0, "requests": 0})
            dt, p50, p95 = run_threads(args.threads, args.calls, lambda t, c: s2.vectorise_ndarray(name, content[(t, c)], **kw))
            ca = coalesce.get_coalescer().stats
            qa = tower.queue_stats().get(True, {"calls": 0, "requests": 0})
            print(f"{name} {items} items/call vectorise()  : {n / dt:9.0f} embeddings/s ({n / items / dt:7.0f} requests/s), latency p50 {p50:.2f} p95 {p95:.2f} ms; "
                  f"queue: {qa['calls'] - qb['calls']} tower calls for {qa['requests'] - qb['requests']} requests; coalescer: {ca['engine_calls'] - cb['engine_calls']} engine calls "
                  f"for {ca['calls'] - cb['calls']} calls", flush=True)
            t0 = time.perf_counter()
            for c in range(args.calls):
                s2.vectorise_ndarray(name, content[(0, c)], **kw)
            dt = time.perf_counter() - t0
            print(f"{name} {items} items/call ONE thread   : {args.calls * items / dt:9.0f} embeddings/s ({args.calls / dt:7.0f} requests/s), {dt / args.calls * 1e3:.2f} ms per call", flush=True)


if __name__ == "__main__":
    main()
