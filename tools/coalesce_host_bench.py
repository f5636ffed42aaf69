"""Interpreter-side ceiling of the cross-request coalescer, measurable WITHOUT a GPU: 16 request threads x 4 items call vectorise_ndarray() on a stub
model whose engine call sleeps (the GIL is released, like a real engine call) for a time that grows with the merged batch.  What is timed is
everything the product's Python does per request — validation, cache key, coalescer hand-offs, result slicing — under contention.
python tools/coalesce_host_bench.py [--threads 16] [--items 4] [--calls 300] [--window 1000]"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from marqo_amd.s2_inference import coalesce
from marqo_amd.s2_inference import s2_inference as s2
from marqo_amd.s2_inference.enums import Modality


class StubModel:
    """engine stand-in: `encode` takes base + per_item microseconds with the GIL released"""
    supports_dynamic_batching = True

    def __init__(self, base_us, per_item_us, dim=512):
        self.base, self.per, self.dim = base_us * 1e-6, per_item_us * 1e-6, dim

    def encode(self, content, normalize=True, modality=None, **kw):
        n = 1 if isinstance(content, str) else len(content)
        time.sleep(self.base + self.per * n)
        return np.zeros((n, self.dim), dtype=np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--items", type=int, default=4)
    ap.add_argument("--calls", type=int, default=300)
    ap.add_argument("--window", default="1000")
    ap.add_argument("--base-us", type=float, default=600.0, help="engine time of a lone small call (measured: 0.66 ms for 4 texts)")
    ap.add_argument("--per-item-us", type=float, default=6.0, help="marginal engine time per item (measured: 0.92 ms for 35 texts)")
    args = ap.parse_args()
    name, dev = "stub/model", "cpu"
    props = {"name": name, "dimensions": 512, "type": "random"}          # (any registered loader type: the cache entry is planted below)
    key = s2._create_model_cache_key(name, dev, props)
    import datetime
    s2._available_models[key] = {s2.AvailableModelsKey.model: StubModel(args.base_us, args.per_item_us),
                                 s2.AvailableModelsKey.most_recently_used_time: datetime.datetime.now(), s2.AvailableModelsKey.model_size: 0.1}
    content = [f"text number {i}" for i in range(args.items)]
    kw = dict(model_properties=props, device=dev, modality=Modality.TEXT)
    for window in ("0", args.window):
        os.environ["MARQO_AMD_COALESCE_US"] = window
        before = dict(coalesce.get_coalescer().stats)
        start = threading.Barrier(args.threads + 1)

        def worker():
            start.wait()
            for _ in range(args.calls):
                s2.vectorise_ndarray(name, content, **kw)
        ts = [threading.Thread(target=worker) for _ in range(args.threads)]
        for t in ts:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        st = coalesce.get_coalescer().stats
        n = args.threads * args.calls * args.items
        calls = st["engine_calls"] - before["engine_calls"]
        print(f"{args.threads} threads x {args.calls} calls x {args.items} items, MARQO_AMD_COALESCE_US={window}: {n / dt:9.0f} items/s "
              f"({args.threads * args.calls / dt:7.0f} requests/s); engine calls {calls}" + (f" ({n / calls:.1f} items each)" if calls else ""), flush=True)
    os.environ["MARQO_AMD_COALESCE_US"] = "0"
    t0 = time.perf_counter()
    for _ in range(args.calls):
        s2.vectorise_ndarray(name, content, **kw)
    dt = time.perf_counter() - t0
    print(f"ONE thread x {args.calls} calls x {args.items} items (serial): {args.calls * args.items / dt:9.0f} items/s")


if __name__ == "__main__":
    main()
