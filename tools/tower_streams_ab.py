"""the headline step (256 uint8 images resident in HBM -> ViT-B/32 tower) as ONE call on one stream against the same call split into k sub-batches on k
HIP streams (`VitTower.n_streams`, MARQO_AMD_STREAMS): interleaved in one process, bit-identity of the embeddings checked.
python tools/tower_streams_ab.py [--workload vit_b32_image] [--batch 256] [--ks 1,2,3,4] [--reps 7] [--steps 20]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vit_b32_image")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--ks", default="1,2,3,4")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda:0"
    torch.cuda.set_device(0)
    w = bench.Workload(a.workload, "bf16", a.batch, dev, 1234)
    tower = w.towers[0]
    ks = [int(k) for k in a.ks.split(",")]
    ref = None
    res = {k: [] for k in ks}
    for k in ks:
        tower.n_streams = k
        out = w.run()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        print(f"k={k}: max |diff| vs k={ks[0]}: {(out - ref).abs().max().item():.3e}", flush=True)
    for rep in range(a.reps):
        for k in ks:
            tower.n_streams = k
            for _ in range(5):
                w.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                w.run()
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / a.steps)
    for k in ks:
        ts = sorted(res[k])
        m = ts[len(ts) // 2]
        print(f"{a.workload} batch {w.batch} streams={k}: median {m * 1e3:.4f} ms/step = {w.batch / m:.0f} emb/s   ({', '.join(f'{t * 1e3:.3f}' for t in res[k])})")


if __name__ == "__main__":
    main()
