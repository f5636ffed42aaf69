"""Single-request latency of the towers (the search path: one query text / one image per call), p50 / p95 over 300 calls.
python tools/latency_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from marqo_amd.engine import archs, synthetic, towers


def timeit(fn, n=300, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        out.cpu()  # what vectorise() does: the embedding goes back to the host
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[int(len(ts) * 0.95)]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="substring of the model names to run")
    args = ap.parse_args()
    dev = "cuda"
    rows = []
    for name in ("ViT-B-32", "ViT-L-14", "ViT-B-16-SigLIP"):
        if args.only and args.only not in name:
            continue
        v, t = archs.resolve_open_clip(name)
        sd = synthetic.random_open_clip_state_dict(vision=v, text=t, seed=0)
        vt, tt = towers.VitTower(v, sd, dev), towers.ClipTextTower(t, sd, dev)
        img = torch.randint(0, 256, (1, v.image_size, v.image_size, 3), dtype=torch.uint8)
        img_d = img.to(dev)
        if t.causal:
            ids = torch.zeros(1, 77, dtype=torch.int64); ids[0, 0] = 49406; ids[0, 1:9] = torch.randint(1, 49406, (8,)); ids[0, 9] = 49407
        else:
            ids = torch.ones(1, t.ctx, dtype=torch.int64); ids[0, :8] = torch.randint(2, t.vocab, (8,))
        rows.append((name + " text, 1 query (10 tokens)" if t.causal else name + " text, 1 query (64 positions)", *timeit(lambda: tt.encode_ids(ids))))
        rows.append((name + " image, 1 x u8 on host", *timeit(lambda: vt.encode_u8(img))))
        rows.append((name + " image, 1 x u8 on device", *timeit(lambda: vt.encode_u8(img_d))))
        rows.append((name + " text, 16 queries", *timeit(lambda: tt.encode_ids(ids.repeat(16, 1)))))
        img4 = torch.randint(0, 256, (4, v.image_size, v.image_size, 3), dtype=torch.uint8).to(dev)
        rows.append((name + " image, 4 x u8 on device", *timeit(lambda: vt.encode_u8(img4))))
        del vt, tt
    if not args.only or args.only in "e5-base-v2":
        b = archs.HF_BERT_ARCHS["intfloat/e5-base-v2"]
        bt = towers.BertTower(b, synthetic.random_bert_state_dict(b, seed=0), dev)
        ids = torch.randint(1000, b.vocab, (1, 12)); mask = torch.ones(1, 12, dtype=torch.int64)
        rows.append(("e5-base-v2 text, 1 query (12 tokens)", *timeit(lambda: bt.encode_ids(ids, mask))))
        rows.append(("e5-base-v2 text, 16 queries (12 tokens)", *timeit(lambda: bt.encode_ids(ids.repeat(16, 1), mask.repeat(16, 1)))))
    for r in rows:
        print(f"{r[0]:44s} p50 {r[1]:7.3f} ms   p95 {r[2]:7.3f} ms")


if __name__ == "__main__":
    main()
