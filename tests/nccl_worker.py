"""Worker for tests/test_nccl_gpu.py, launched with `python -m torch.distributed.run`: runs the sharded bulk-ingest flush
(marqo_amd.ingest.BulkVectoriser -> vectorise_device -> marqo_amd.parallel.gather_embeddings) over a REAL RCCL process group
(backend "nccl") and checks every rank's result against the single-process path.  Prints one JSON line per rank."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
    os.environ["MARQO_MAX_CUDA_MODEL_MEMORY"] = "64"
    res = {"rank": rank, "world": world, "device": dev, "ok": False}
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        # a bare collective first: a failure here is RCCL refusing the topology (e.g. two ranks on one GPU), not our code
        probe = torch.full((4, 8), float(rank), device=dev)
        out = torch.empty(world * 4, 8, device=dev)
        dist.all_gather_into_tensor(out, probe)
        torch.cuda.synchronize()
        assert all(float(out[r * 4, 0]) == r for r in range(world))
        from marqo_amd.ingest import BulkVectoriser
        from marqo_amd.s2_inference import s2_inference as s2
        from marqo_amd.s2_inference.enums import Modality
        from PIL import Image
        name = "open_clip/ViT-B-32/laion2b_s34b_b79k"
        rng = np.random.default_rng(0)
        texts = [("word " * (1 + (7 * i) % 40)).strip() + f" {i}" for i in range(37)]
        imgs = [Image.fromarray(rng.integers(0, 256, (64 + 8 * (i % 5), 96, 3), dtype=np.uint8)) for i in range(9)]
        ref_t = s2.vectorise_ndarray(name, texts, device=dev)
        ref_i = s2.vectorise_ndarray(name, imgs, device=dev, modality=Modality.IMAGE)
        bv = BulkVectoriser(name, dev)
        bv.force_collective = True
        for i, t in enumerate(texts):
            bv.add(("t", i), t)
        for i, im in enumerate(imgs):
            bv.add(("i", i), im, Modality.IMAGE)
        got = bv.flush()
        t = np.stack([got[("t", i)] for i in range(len(texts))])
        im = np.stack([got[("i", i)] for i in range(len(imgs))])
        # the gathered matrix is in request order and equals the un-sharded result (same kernels; batch composition differs, so
        # compare to bf16-level tolerance, not bitwise)
        cos = lambda a, b: float((1 - (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))).max())
        res.update(ok=bool(cos(t, ref_t) < 1e-5 and cos(im, ref_i) < 1e-5), cos_text=cos(t, ref_t), cos_image=cos(im, ref_i),
                   backend=dist.get_backend(), nccl_version=str(torch.cuda.nccl.version()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 - reported to the parent test
        res["error"] = f"{type(e).__name__}: {e}"[:600]
    print("NCCL_WORKER " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
