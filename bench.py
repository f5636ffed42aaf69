#!/usr/bin/env python
"""bench.py — embeddings/sec of the vectorise() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Headline workload (BASELINE.json configs[1], the config the metric is quoted on): open_clip ViT-B/32 image tower, synthetic uint8
224x224 RGB images ALREADY RESIDENT IN HBM (tower-only: `value` excludes host packing, H2D, D2H and list conversion — those are
reported beside it under `e2e_vectorise`), 256 images per GPU per step, random-init weights of that architecture (no network for
checkpoints).  One "step" = one pass of the hot path over one batch: normalise + patchify -> patch-embed GEMM -> 12 pre-LN blocks ->
ln_post -> projection -> L2, i.e. exactly mq_encode_image_u8; for N > 1 the [256, 512] fp32 shards are then all-gathered over RCCL
(the only collective on the path).  Work per GPU is fixed -> weak scaling.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline        — the dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs / HIP-event time of its launches
  cpu_baseline    — the CPU fp32 oracle (oracle/towers.py) run with the reference's 16-item batch loop (s2_inference.py:135-146) on a
                    bounded sample; `cores` = the torch thread count of the fastest run of a thread ladder, `host_hw_threads` / `cpu_quota` = what the box offers
  cos_err_vs_cpu  — max (1 - cosine) of the GPU embeddings vs that CPU path on the same sample
  e2e_vectorise   — (N = 1) the same 256 images through the product's `vectorise_ndarray()` / `vectorise()` from HOST memory, as
                    SURVEY.md §8(d) defines end-to-end: pack + H2D + K10 + tower + D2H (+ `.tolist()`), for each hand-over form
                    (PIL list / uint8 ndarray list / `.preprocess`ed device tensors), single caller and 4 concurrent callers
  also            — (N = 1) the text half of the "images + text" metric and BASELINE configs[2] on the same box: CLIP text B/32,
                    e5-base-v2 @ 77 tokens and the ViT-L/14 mixed batch, each tower-only with its own cpu_baseline
`--workload add_documents_mixed` is BASELINE configs[3] in miniature: mixed text + image documents in 128-document requests through
marqo_amd.ingest.BulkVectoriser, sharded over the ranks, gathered in request order.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
FP8_DENSE_PEAK_TFLOPS = 5000.0   # MX-scaled fp8 MFMA (K = 128), ~5 PF dense
WORKLOADS = {
    # the headline workload (BASELINE.json configs[1]) is the default; the others are reported in DESIGN.md §6
    "vit_b32_image": dict(kind="image", arch="ViT-B-32", desc="open_clip ViT-B/32 image tower, uint8 224x224 resident in HBM (tower-only), batch 256/GPU", batch=256),
    "vit_l14_image": dict(kind="image", arch="ViT-L-14", desc="open_clip ViT-L/14 image tower, uint8 224x224, batch 64/GPU", batch=64),
    "vit_h14_image": dict(kind="image", arch="ViT-H-14", desc="open_clip ViT-H/14 image tower (80-wide heads run as 96), uint8 224x224, batch 64/GPU", batch=64),
    "vit_bigg14_image": dict(kind="image", arch="ViT-bigG-14", desc="open_clip ViT-bigG/14 image tower (104-wide heads run as 112), uint8 224x224, batch 32/GPU", batch=32),
    "siglip_b16_image": dict(kind="image", arch="ViT-B-16-SigLIP", desc="open_clip ViT-B-16-SigLIP image tower (196 tokens, attention-pool head), uint8 224x224, batch 128/GPU", batch=128),
    "siglip_l16_384_image": dict(kind="image", arch="ViT-L-16-SigLIP-384", desc="open_clip ViT-L-16-SigLIP-384 image tower (576 tokens), uint8 384x384, batch 32/GPU", batch=32),
    "eva02_b16_image": dict(kind="image", arch="EVA02-B-16", desc="open_clip EVA02-B-16 image tower (timm Eva: 197 tokens, 2-D rotary positions, sub-LayerNorms, SwiGLU), uint8 224x224, batch 128/GPU", batch=128),
    "eva02_l14_image": dict(kind="image", arch="EVA02-L-14", desc="open_clip EVA02-L-14 image tower (257 tokens, SwiGLU hidden 2 730 -> 2 752), uint8 224x224, batch 64/GPU", batch=64),
    "siglip_b16_text": dict(kind="clip_text", arch="ViT-B-16-SigLIP", desc="open_clip ViT-B-16-SigLIP text tower (unmasked, 64 positions), batch 1024/GPU", batch=1024),
    "clip_text_b32": dict(kind="clip_text", arch="ViT-B-32", desc="open_clip ViT-B/32 text tower, 77-token ids, batch 1024/GPU", batch=1024),
    "clip_text_l14": dict(kind="clip_text", arch="ViT-L-14", desc="open_clip ViT-L/14 text tower, 77-token ids, batch 1024/GPU", batch=1024),
    "bert_base_77": dict(kind="bert", arch="intfloat/e5-base-v2", desc="e5-base-v2 (BERT-base) + mean-pool + L2, 77-token ids, batch 1024/GPU", batch=1024),
    "vit_l14_mixed": dict(kind="mixed", arch="ViT-L-14", desc="open_clip ViT-L/14 dual encoder, 128 images + 128 texts (5..75 tokens) per GPU (BASELINE configs[2])", batch=256),
    "vit_l14_chunked_fp8": dict(kind="chunked", arch="ViT-L-14", desc="BASELINE configs[4]: 480x640 uint8 images resident in HBM -> K11 'simple' 3x3 grid chunking on the GPU "
                                "(resize to 240x240, whole image + 9 cells = 10 crops per image, each through the CLIP transform) -> ViT-L/14 image tower under the "
                                "load-time fp8 block-split policy; 24 images = 240 crop embeddings per GPU per step", batch=24),
    "vit_l14_chunked_fp8_trained": dict(kind="chunked", arch="ViT-L-14", weights="trained_like", desc="BASELINE configs[4] on a REPRESENTATIVE tower: the same step (480x640 uint8 sources in HBM "
                                        "-> K11 3x3 grid -> 10 crops per image -> ViT-L/14 under the load-time fp8 policy) with trained-like weight statistics (LayerNorm gain spread, outlier "
                                        "channels, peaky attention, class-token massive activation: engine/synthetic.py) and source images of natural-image statistics (1/f spectrum, "
                                        "correlated channels) — what the policy picks on a real checkpoint rather than on N(0, s) weights; 24 images = 240 crop embeddings per GPU per step", batch=24),
    "add_documents_stream": dict(kind="stream", arch="ViT-B-32", desc="BASELINE configs[3] as a stream: mixed {text, 224x224 PIL image} documents in 128-document requests; every "
                                 "rank owns whole requests (request i -> rank i % N, nothing is sharded inside a request), runs them through the single-GPU "
                                 "BulkVectoriser path, consecutive owned requests MERGED into chip-filling groups (one tower call per modality and group, text and images on two host threads / HIP streams) with ONE group in flight (group g + 1 is tokenised / packed / enqueued while group g runs; rows are copied to the host behind their tower) and ONE gather onto rank 0 closes the stream; ViT-B/32", batch=128),
    "stub": dict(kind="stub", arch="-", desc="launcher self-test: no GPU work, one gloo all_gather per step (tests/test_bench_launcher.py)", batch=4),
    "add_documents_mixed": dict(kind="ingest", arch="ViT-B-32", desc="add_documents bulk ingest in miniature (BASELINE configs[3]): documents {text, 224x224 image} in "
                                "128-document requests through BulkVectoriser (host PIL images + strings -> vectorise -> gather in order), ViT-B/32", batch=128),
}
# the default line's `also` rows: the text half of the metric, BASELINE configs[2], configs[4] (fp8 policy + on-GPU chunker) and configs[3] (ingest stream)
ALSO_DEFAULT = ("clip_text_b32", "bert_base_77", "vit_l14_mixed", "vit_l14_chunked_fp8", "vit_l14_chunked_fp8_trained", "add_documents_stream")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="vit_b32_image", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="items per GPU per step (default: the workload's)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp8"],
                    help="GEMM operand type of the encoder blocks (fp8 = e4m3, BASELINE config 5; not the headline metric)")
    ap.add_argument("--merge-images", type=int, default=None, help="add_documents_stream: images per merged tower call (0 = no cross-request merging; default: marqo_amd.ingest.MERGE_IMAGES)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the headline baseline sample")
    ap.add_argument("--no-extras", action="store_true", help="skip e2e_vectorise and the `also` workloads (profiling runs)")
    ap.add_argument("--no-also", action="store_true", help="keep e2e_vectorise but skip the `also` workloads (host-side A/B runs)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py spawns its own ranks (0 = pick a free one)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: re-exec this command under it (one rank per GPU, 127.0.0.1 rendezvous) and
    pass rank 0's JSON line through.  The driver's own form (`python -m torch.distributed.run ... bench.py --gpus N`) sets WORLD_SIZE and
    never gets here."""
    import socket
    import subprocess
    port = args.master_port
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def run_stub(args, rank, world):
    """launcher self-test (no GPU): every rank contributes [batch, 8] rows, gloo all_gather, rank 0 prints the contract line"""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = args.batch or WORKLOADS["stub"]["batch"]
    rows = torch.full((batch, 8), float(rank))

    def step():
        if world == 1:
            return rows
        outs = [torch.empty_like(rows) for _ in range(world)]
        dist.all_gather(outs, rows)
        return torch.cat(outs)

    def fence():
        if world > 1:
            dist.barrier()
    elapsed, out = timed(step, args.steps, args.warmup, fence)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out.shape[0] == batch * world and sorted(set(out[:, 0].tolist())) == [float(r) for r in range(world)]
    if rank == 0:
        print(json.dumps({"metric": "embeddings/sec", "value": round(batch * world * args.steps / elapsed, 1), "unit": "embeddings/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": WORKLOADS["stub"]["desc"], "global_batch": batch * world, "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ---- CPU baselines (oracle/ is the checker; it is only ever timed here, never on the product path) ---------------------------------
def _best_threads(run16, target_seconds):
    cores = os.cpu_count() or 1
    t_start = time.perf_counter()
    best_threads, best_dt = None, None
    for th in [t for t in (16, 32, 64, 128) if t < cores] + [cores]:
        torch.set_num_threads(th)
        run16()  # warm-up at this thread count
        t0 = time.perf_counter()
        run16()
        dt = time.perf_counter() - t0
        if best_dt is None or dt < best_dt:
            best_threads, best_dt = th, dt
        elif dt > 1.3 * best_dt:
            break
        if time.perf_counter() - t_start > 0.5 * target_seconds:
            break
    torch.set_num_threads(best_threads)
    return best_threads, best_dt, cores, max(target_seconds - (time.perf_counter() - t_start), best_dt)


def cpu_baseline_run(run_items, n_total, target_seconds, min_items=16):
    """run_items(lo, hi) -> fp32 embeddings of items [lo, hi) through the reference-equivalent CPU path (fp32 PyTorch eager, the
    reference's 16-item batch loop).  The thread count is the best of a short ladder (all host threads is often NOT the fastest on
    a 2-socket SMT box).  -> (emb/s, n, embeddings, threads used, host cores)."""
    def run(lo, hi):
        outs = [run_items(i, min(i + 16, hi)) for i in range(lo, hi, 16)]  # MARQO_MAX_VECTORISE_BATCH_SIZE default (api/configs.py:38)
        return torch.cat(outs, dim=0)
    threads, dt16, cores, budget = _best_threads(lambda: run(0, min(16, n_total)), target_seconds)
    n = int(min(n_total, max(min_items, (budget / max(dt16, 1e-3)) * 16) // 16 * 16))
    t0 = time.perf_counter()
    emb = run(0, n)
    dt = time.perf_counter() - t0
    return n / dt, n, emb, threads, cores


def _cos_err(gpu, cpu):
    g, c = gpu.double(), cpu.double()
    return float((1 - (g * c).sum(-1) / (g.norm(dim=-1) * c.norm(dim=-1))).max())


def _baseline_dict(rate, n, threads, cores, what):
    try:   # the container may grant far less CPU time than the host has hardware threads (the GPU boxes: 16 CPUs' worth of 256)
        from marqo_amd._lib import cpu_quota
        quota = cpu_quota()
    except Exception:  # noqa: BLE001 - informational
        quota = None
    # `cores` = the threads the winning run of the ladder actually used (the bench contract's meaning); the host's hardware threads and the
    # container's CPU quota are reported beside it
    return {"value": round(rate, 2), "unit": "embeddings/s", "cores": threads, "host_hw_threads": cores, "cpu_quota": quota, "kind": "port",
            "sample": f"{n} {what}, fp32 PyTorch eager, 16-item batches (reference loop s2_inference.py:135-146), best of a 16/32/64/128/all thread ladder"}


# ---- workloads ------------------------------------------------------------------------------------------------------------------
class Workload:
    """towers + HBM-resident synthetic inputs of one bench workload; `run()` enqueues one step and returns [batch, D] on device"""

    def __init__(self, name, precision, batch, dev, seed):
        from marqo_amd.engine import archs, synthetic, towers
        wl = WORKLOADS[name]
        self.name, self.wl, self.kind, self.batch, self.dev = name, wl, wl["kind"], batch or wl["batch"], dev
        batch = self.batch
        g = torch.Generator().manual_seed(seed)
        self.varch = self.tarch = self.barch = None
        if self.kind in ("image", "clip_text", "mixed", "chunked"):
            self.varch, self.tarch = archs.resolve_open_clip(wl["arch"])

        def clip_ids(n, lo, hi):
            ids = torch.zeros(n, 77, dtype=torch.int64)
            lens = torch.randint(lo, hi + 1, (n,), generator=g)
            for i in range(n):
                li = int(lens[i])
                ids[i, 0] = 49406
                ids[i, 1:1 + li] = torch.randint(1, 49406, (li,), generator=g)
                ids[i, 1 + li] = 49407
            return ids

        self.images_cpu = self.ids_cpu = None
        if self.kind == "image":
            self.sd = synthetic.random_open_clip_state_dict(vision=self.varch, seed=0)
            tower = towers.VitTower(self.varch, self.sd, dev, precision=precision)
            self.images_cpu = torch.randint(0, 256, (batch, self.varch.image_size, self.varch.image_size, 3), generator=g, dtype=torch.uint8)
            images = self.images_cpu.to(dev)
            self.towers = [tower]
            self.gflop_per_emb = self.varch.gflop_per_image
            self.run = lambda: tower.encode_u8(images)
        elif self.kind == "clip_text":
            tarch = self.tarch
            self.sd = synthetic.random_open_clip_state_dict(text=tarch, seed=0)
            tower = towers.ClipTextTower(tarch, self.sd, dev, precision=precision)
            self.towers = [tower]
            self.gflop_per_emb = tarch.gflop_per_text(tarch.ctx)
            # ids resident in HBM like the images (what the device tokeniser hands over); only the n lengths live on the host
            if tarch.causal:
                ids = clip_ids(batch, 75, 75)
                d_ids, lens = ids.to(torch.int32).to(dev), ids.argmax(1) + 1
            else:  # SigLIP: every text is ctx positions (pieces ... </s> then </s> padding), all of them run
                ids = torch.ones(batch, tarch.ctx, dtype=torch.int64)
                ids[:, :20] = torch.randint(2, tarch.vocab, (batch, 20), generator=g)
                d_ids, lens = ids.to(torch.int32).to(dev), torch.full((batch,), tarch.ctx, dtype=torch.int64)
            self.ids_cpu = ids
            self.run = lambda: tower.encode_device(d_ids, lens)
        elif self.kind == "bert":
            self.barch = archs.HF_BERT_ARCHS[wl["arch"]]
            self.sd = synthetic.random_bert_state_dict(self.barch, seed=0)
            tower = towers.BertTower(self.barch, self.sd, dev, precision=precision)
            ids = torch.randint(1000, self.barch.vocab, (batch, 77), generator=g)
            ids[:, 0], ids[:, -1] = 101, 102
            self.ids_cpu = ids
            self.towers = [tower]
            self.gflop_per_emb = self.barch.gflop_per_text(77)
            d_ids, lens = ids.to(torch.int32).to(dev), torch.full((batch,), 77, dtype=torch.int64)
            self.run = lambda: tower.encode_device(d_ids, lens)
        elif self.kind == "chunked":   # device-resident source images -> grid chunker (K11) -> tower; embeddings = crops
            from marqo_amd.engine.preprocess import ImagePreprocessor
            if wl.get("weights") == "trained_like":
                self.sd = synthetic.trained_like_open_clip_state_dict(self.varch, seed=2)
                self.src_cpu = list(synthetic.natural_images_u8(batch, 480, 640, seed=seed))
            else:
                self.sd = synthetic.random_open_clip_state_dict(vision=self.varch, seed=0)
                self.src_cpu = [torch.randint(0, 256, (480, 640, 3), generator=g, dtype=torch.uint8) for _ in range(batch)]
            tower = towers.VitTower(self.varch, self.sd, dev, precision=precision)
            src = [t.to(dev) for t in self.src_cpu]
            pre = ImagePreprocessor(dev, self.varch.image_size)
            self.crops_per_image = 10
            self.towers = [tower]
            self.gflop_per_emb = self.varch.gflop_per_image
            self.pre = pre

            def run_chunked():
                crops, _ = pre.chunk_grid_u8(src, 3, 3, False)       # uint8 [n * 10, S, S, 3] on the device
                return tower.encode_u8(crops)
            self.run = run_chunked
            self.batch = batch * self.crops_per_image                # `value` counts embeddings = crops
        elif self.kind == "mixed":  # half images, half texts of ragged length through the two towers of one model
            self.sd = synthetic.random_open_clip_state_dict(vision=self.varch, text=self.tarch, seed=0)
            vt = towers.VitTower(self.varch, self.sd, dev, precision=precision)
            tt = towers.ClipTextTower(self.tarch, self.sd, dev, precision=precision)
            n_img = batch // 2
            self.n_img = n_img
            self.images_cpu = torch.randint(0, 256, (n_img, self.varch.image_size, self.varch.image_size, 3), generator=g, dtype=torch.uint8)
            images = self.images_cpu.to(dev)
            ids = clip_ids(batch - n_img, 5, 75)
            self.ids_cpu = ids
            self.towers = [vt, tt]
            d_ids, lens = ids.to(torch.int32).to(dev), ids.argmax(1) + 1
            mean_tokens = float((ids.argmax(1) + 1).float().mean())
            self.gflop_per_emb = (n_img * self.varch.gflop_per_image + (batch - n_img) * self.tarch.gflop_per_text(int(round(mean_tokens)))) / batch
            self.run = lambda: torch.cat([vt.encode_u8(images), tt.encode_device(d_ids, lens)], dim=0)
        else:
            raise ValueError(self.kind)
        self.fp8_policy = None
        if precision == "fp8":
            # load-time policy of the product (engine/towers.py::tune_fp8): static scales + how many trailing blocks run on e4m3 inside
            # MARQO_AMD_FP8_BUDGET (default 5e-4 vs the bf16 tower; set it to 1 to force every block onto fp8), outside the timed region
            self.fp8_policy = [{"layers": t.cfg.enc.layers, "fp8_first_layer": t.tune_fp8_default(), "fp8_mlp_extra": t.fp8_mlp_extra, "policy_trace": t.fp8_policy_trace,
                                "residual_stream": t.residual_stream, "calibration_err_vs_bf16": t.fp8_calibration_error,
                                "all_blocks_err_vs_bf16": t.fp8_all_blocks_error} for t in self.towers]

    def cpu_baseline(self, gpu_out, target_seconds):
        """reference-equivalent CPU path on a bounded sample of this workload's own inputs + cosine error of the GPU result"""
        from oracle import towers as O
        if self.kind == "image":
            a = self.varch
            if a.pool == "map":
                cfg, fwd = O.SiglipVitConfig(a.image_size, a.patch_size, a.width, a.layers, a.heads, a.mlp_dim), O.siglip_vit_forward
            elif a.eva:
                cfg, fwd = O.EvaVitConfig(a.image_size, a.patch_size, a.width, a.layers, a.heads, a.mlp_dim, a.out_dim, ref_grid=a.rope_ref_grid), O.eva_vit_forward
            else:
                cfg, fwd = O.VitConfig(a.image_size, a.patch_size, a.width, a.layers, a.heads, a.mlp_dim, a.out_dim, a.quick_gelu), O.vit_forward
            rate, n, emb, th, cores = cpu_baseline_run(lambda lo, hi: fwd(self.sd, cfg, O.preprocess_u8_exact_size(self.images_cpu[lo:hi])),
                                                       self.batch, target_seconds)
            what = f"of the step's {self.batch} images"
        elif self.kind == "chunked":
            # reference flow on the host: chunk_image 'simple' (PIL resize to 240x240 + crops, processing/image.py:46-151) -> CLIP transform per
            # crop (clip_utils.py:48-67) -> ViT forward in fp32, 16-item batches
            from oracle import preprocess as OP
            a = self.varch
            cfg = O.VitConfig(a.image_size, a.patch_size, a.width, a.layers, a.heads, a.mlp_dim, a.out_dim, a.quick_gelu)
            n_src = len(self.src_cpu)

            def crops_of(lo, hi):   # item index = crop index; image = index // 10
                out = []
                for im in range(lo // 10, (hi + 9) // 10):
                    patches, _ = OP.chunk_image_simple(self.src_cpu[im].numpy(), 3, 3, False)
                    out.extend(OP.clip_transform(p_, a.image_size) for p_ in patches)
                off = lo - (lo // 10) * 10
                return torch.from_numpy(np.stack(out[off:off + (hi - lo)]))
            # (>= 64 crops whatever the time budget: the fp8 policy's error is the number this workload exists to show)
            rate, n, emb, th, cores = cpu_baseline_run(lambda lo, hi: O.vit_forward(self.sd, cfg, crops_of(lo, hi)),
                                                       n_src * 10, target_seconds, min_items=64)
            what = f"crops of the step's {n_src} source images (PIL-exact 240x240 resize + 3x3 grid + CLIP transform on the host, then the fp32 tower)"
        elif self.kind == "clip_text" and self.tarch.causal:
            t = self.tarch
            cfg = O.ClipTextConfig(t.vocab, t.ctx, t.width, t.layers, t.heads, t.mlp_dim, t.out_dim, t.quick_gelu)
            rate, n, emb, th, cores = cpu_baseline_run(lambda lo, hi: O.clip_text_forward(self.sd, cfg, self.ids_cpu[lo:hi]), self.batch, target_seconds)
            what = f"of the step's {self.batch} texts (all 77 positions, as the reference runs them)"
        elif self.kind == "bert":
            b = self.barch
            cfg = O.BertConfig(b.vocab, b.max_pos, b.width, b.layers, b.heads, b.mlp_dim, b.ln_eps, "mean", b.pos_offset)
            rate, n, emb, th, cores = cpu_baseline_run(lambda lo, hi: O.hf_encode(self.sd, cfg, self.ids_cpu[lo:hi], torch.ones_like(self.ids_cpu[lo:hi])),
                                                       self.batch, target_seconds)
            what = f"of the step's {self.batch} texts of 77 tokens"
        elif self.kind == "mixed":
            a, t = self.varch, self.tarch
            vcfg = O.VitConfig(a.image_size, a.patch_size, a.width, a.layers, a.heads, a.mlp_dim, a.out_dim, a.quick_gelu)
            tcfg = O.ClipTextConfig(t.vocab, t.ctx, t.width, t.layers, t.heads, t.mlp_dim, t.out_dim, t.quick_gelu)
            # interleave so that a bounded sample keeps the 50/50 mix: item 2i = image i, item 2i+1 = text i
            def run_items(lo, hi):
                i0, i1 = lo // 2, hi // 2
                im = O.vit_forward(self.sd, vcfg, O.preprocess_u8_exact_size(self.images_cpu[i0:i1]))
                tx = O.clip_text_forward(self.sd, tcfg, self.ids_cpu[i0:i1])
                return torch.stack([im, tx], dim=1).reshape(-1, im.shape[1])
            rate, n, emb, th, cores = cpu_baseline_run(run_items, 2 * min(self.n_img, self.batch - self.n_img), target_seconds)
            k = n // 2
            gpu_out = torch.stack([gpu_out[:k], gpu_out[self.n_img:self.n_img + k]], dim=1).reshape(-1, gpu_out.shape[1])
            what = f"items ({k} images + {k} texts) of the step's {self.batch}"
        else:
            return None, None
        return _baseline_dict(rate, n, th, cores, what), _cos_err(gpu_out[:n].float().cpu(), emb)


def timed(run, steps, warmup, fence):
    for _ in range(warmup):
        run()
    fence()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = run()
    fence()
    return time.perf_counter() - t0, out


_SUSTAINED = {}


def measure_sustained_peak(lib, L, target_ms=60.0):
    """mq_probe_mfma_peak once per process (median of 3 burns of ~60 ms); falls back to the round-1 constant if the probe fails"""
    if not _SUSTAINED:
        try:
            scratch = torch.empty(1 << 16, dtype=torch.uint8, device="cuda")
            tf, mhz = C.c_double(0.0), C.c_double(0.0)
            runs = []
            for _ in range(3):
                L.check(lib.mq_probe_mfma_peak(target_ms, scratch.data_ptr(), scratch.numel(), C.byref(tf), C.byref(mhz),
                                               torch.cuda.current_stream().cuda_stream), "mq_probe_mfma_peak")
                runs.append((tf.value, mhz.value))
            runs.sort()
            _SUSTAINED.update(tflops=runs[1][0], shader_mhz=runs[1][1], source="mq_probe_mfma_peak in this run (median of 3 x 60 ms burns)")
        except Exception as e:  # noqa: BLE001 - never lose the line to the probe
            _SUSTAINED.update(tflops=2114.0, shader_mhz=2036.0, source=f"constant from profiles/r01f_mfma_sustained_peak.txt (probe failed: {type(e).__name__})")
    return _SUSTAINED


def gemm_roofline(lib, L, run_local, prof_steps, precision, workload_name):
    """HIP-event timing of the GEMM family on the launch stream (separate, instrumented steps)"""
    lib.mq_profile_enable(1)
    for _ in range(prof_steps):
        run_local()
    ms = (C.c_double * L.MQ_PROF_FAMILIES)()
    cnt = (C.c_int64 * L.MQ_PROF_FAMILIES)()
    flops = C.c_double(0.0)
    L.check(lib.mq_profile_collect(ms, cnt, C.byref(flops)), "mq_profile_collect")
    lib.mq_profile_enable(0)
    gemm_ms, gemm_launches = ms[0], cnt[0]
    achieved = flops.value / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    peak = FP8_DENSE_PEAK_TFLOPS if precision == "fp8" else BF16_DENSE_PEAK_TFLOPS
    families = {L.PROF_FAMILY_NAMES[i]: {"ms_per_step": ms[i] / prof_steps, "launches_per_step": cnt[i] // prof_steps}
                for i in range(L.MQ_PROF_FAMILIES) if cnt[i]}
    roofline = {
        "kernel": ("gemm_fp8_kernel (e4m3 MX MFMA 16x16x128, (32*MT)x128x128 tiles, software-pipelined k-loop, fused epilogues; bf16 gemm_nt_kernel in the blocks the policy keeps on bf16)" if precision == "fp8"
                   else "gemm_nt_kernel (bf16 MFMA 16x16x32, (32*MT)x128x64 tiles, software-pipelined k-loop, fused epilogues incl. the folded LayerNorm)"),
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": None,
        "flops_per_launch": flops.value / max(gemm_launches, 1),
        "avg_launch_us": gemm_ms * 1e3 / max(gemm_launches, 1),
        "launches_per_step": gemm_launches // max(prof_steps, 1),
        "per_family": families,
    }
    # (ViT-B/32-shaped blocks at chip-filling batches run attention + out-projection + residual + statistics as ONE launch, csrc/attn_proj.hip: that launch's
    # time and count sit in the `attention` family, its out-projection FLOPs — 15.1 GF per block at the headline — in neither `achieved` nor its divisor)
    roofline["note"] = ("`achieved` = the tiled GEMM launches' own FLOPs / their own time; where a block takes mq_attention_proj (attention + out-projection + residual + "
                        "LayerNorm statistics in one launch) its out-projection is timed in per_family.attention")
    if precision == "bf16":
        # what a register-resident v_mfma_f32_16x16x32_bf16 burn sustains on THIS chip, in THIS run, at the clock it holds under full MFMA load
        # (csrc/probe.hip; boxes of the pool hold 1.86-2.03 GHz): `peak` / `frac` stay priced against the guide's 2.4 GHz dense figure,
        # `frac_of_sustained` is the number that is comparable across boxes.
        sustained = measure_sustained_peak(lib, L)
        roofline["peak_sustained_measured"] = round(sustained["tflops"], 1)
        roofline["peak_sustained_source"] = sustained["source"]
        roofline["shader_mhz_under_mfma_load"] = round(sustained["shader_mhz"], 0)
        roofline["frac_of_sustained"] = round(achieved / sustained["tflops"], 4)
    # how `achieved` was timed: HIP events around every launch of the family add the event records' own dispatch gaps to the kernel time:
    # against rocprofv3's kernel trace of the same command the GEMM family read 2.648 vs 2.553 ms per step (round 4) — `achieved` / `frac`
    # UNDER-state the kernels by about 4 %
    roofline["timing"] = "HIP events around each launch on the launch stream; overstates kernel time by ~4 % vs rocprofv3 --kernel-trace (profiles/)"
    # HBM-side traffic of the dominant kernel from the committed PMC passes of this same command (tools/gpu_evidence.sh:
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH doubled per MI355X_MICROARCH.md; PMC cannot be sampled
    # from inside this process).  Bytes per launch, averaged over the GEMM launches like `achieved`.
    for rnd in ("r08f", "r06", "r05", "r04", "r03", "r02", "r01"):
        tpath = os.path.join(ROOT, "profiles", f"{rnd}_traffic_{workload_name}_{precision}.json")
        if not os.path.isfile(tpath):
            continue
        try:
            with open(tpath) as f:
                tj = json.load(f)
            fam = "gemm_fp8_kernel" if precision == "fp8" else "gemm_nt_kernel"
            rows = [v for k, v in tj.items() if k.startswith(fam)]
            n_l = sum(v["launches"] for v in rows)
            if n_l:
                roofline["traffic"] = round(sum(v["launches"] * (v["read_bytes"] + v["write_bytes"]) for v in rows) / n_l, 1)
                roofline["traffic_unit"] = "bytes/launch (fabric-side reads incl. Infinity-Cache hits + writes)"
                roofline["traffic_source"] = os.path.relpath(tpath, ROOT)
                roofline["traffic_measured"] = "from_file"     # PMC passes cannot run inside this process: a committed measurement of the same command
                break
        except (OSError, ValueError, KeyError):
            pass
    return roofline


# ---- end-to-end vectorise() from host memory (wall time includes preprocessing + H2D + towers + D2H) -------------------
def e2e_vectorise(dev, images_cpu_u8, tower_only_rate, reps=21):
    """the SAME images as the headline step, but handed over from HOST memory through the product API (random-init weights via
    MARQO_AMD_SYNTHETIC_WEIGHTS, the registry name of the headline model)"""
    from PIL import Image
    os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
    os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
    from marqo_amd.s2_inference import s2_inference as s2
    from marqo_amd.s2_inference.enums import Modality
    name = "open_clip/ViT-B-32/laion2b_s34b_b79k"
    n = images_cpu_u8.shape[0]
    arrs = [images_cpu_u8[i].numpy() for i in range(n)]
    pil = [Image.fromarray(a) for a in arrs]
    props = s2.get_model_properties_from_registry(name)
    model, pre = s2.load_multimodal_model_and_get_preprocessors(name, props, dev)
    dev_tensors = [pre["image"](p) for p in pil]        # what add_documents' download threads hand over (add_docs.py:130-134)
    torch.cuda.synchronize()

    def rate(fn, reps=reps, warm=5):
        """items per second of the MEDIAN call (every call is synchronous: it returns host rows).  Three warm-up calls: the first ones of a form
        allocate its pinned staging blocks and workspaces (tens of ms each), and a mean over six calls is hostage to one host hiccup —
        profiles/r03final_bench_default.json reported 20.3 k for a form whose sibling measured 33.3 k seconds later"""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return n / ts[len(ts) // 2]

    kw = dict(device=dev, modality=Modality.IMAGE, model_properties=props)
    out = {"n_images": n, "model": name, "unit": "embeddings/s",
           "note": "one synchronous vectorise_ndarray() call per 256 images: host pack -> pinned H2D -> K10 resize -> tower -> D2H"}
    out["ndarray_from_pil"] = round(rate(lambda: s2.vectorise_ndarray(name, pil, **kw)), 1)
    out["ndarray_from_u8_arrays"] = round(rate(lambda: s2.vectorise_ndarray(name, arrs, **kw)), 1)
    out["ndarray_from_device_tensors"] = round(rate(lambda: s2.vectorise_ndarray(name, dev_tensors, **kw)), 1)
    out["list_from_pil"] = round(rate(lambda: s2.vectorise(name, pil, **kw)), 1)
    emb = s2.vectorise_ndarray(name, pil, **kw)
    t0 = time.perf_counter()
    for _ in range(20):
        s2._convert_vectorized_output(emb)
    out["tolist_ms_per_call"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    # the reference serves up to 8 indexing threads (api/configs.py:27): 4 concurrent callers, each its own 256-image requests
    def concurrent(content, threads=4, reps=6, warm=2):
        """`threads` long-lived request threads (the reference's API workers are pooled, not created per request): each warms its own
        stream / workspace / pinned staging blocks with `warm` calls, then all start the timed `reps` calls together"""
        start, stop = threading.Barrier(threads + 1), threading.Barrier(threads + 1)
        errors = []

        def worker():
            try:
                for _ in range(warm):
                    s2.vectorise_ndarray(name, content, **kw)
                torch.cuda.synchronize()
                start.wait()
                for _ in range(reps):
                    s2.vectorise_ndarray(name, content, **kw)
                stop.wait()
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
                start.abort(), stop.abort()
        ts = [threading.Thread(target=worker) for _ in range(threads)]
        for t in ts:
            t.start()
        try:
            start.wait()
            t0 = time.perf_counter()
            stop.wait()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        except threading.BrokenBarrierError:
            elapsed = float("nan")
        for t in ts:
            t.join()
        if errors:
            raise errors[0]
        return n * reps * threads / elapsed
    out["ndarray_from_u8_arrays_4_callers"] = round(concurrent(arrs), 1)
    out["ndarray_from_device_tensors_4_callers"] = round(concurrent(dev_tensors), 1)
    out["ndarray_from_pil_4_callers"] = round(concurrent(pil), 1)
    out["tower_only"] = round(tower_only_rate, 1)
    best = max(out["ndarray_from_u8_arrays_4_callers"], out["ndarray_from_device_tensors_4_callers"], out["ndarray_from_u8_arrays"],
               out["ndarray_from_device_tensors"], out["ndarray_from_pil_4_callers"], out["ndarray_from_pil"])
    out["best_e2e_over_tower_only"] = round(best / tower_only_rate, 3)
    # THE end-to-end figure: one synchronous caller handing over PIL images, as add_documents / search hand them to vectorise()
    out["headline_e2e"] = {"form": "ndarray_from_pil", "value": out["ndarray_from_pil"], "over_tower_only": round(out["ndarray_from_pil"] / tower_only_rate, 3),
                           "note": "a single synchronous caller: the call runs in two 128-image stages (host pack of 2 x 25 MB of Pillow RGBX, H2D of 64-image slices under it; "
                                   "the stages' towers on two HIP streams, enqueued by a helper thread) + D2H (profiles/r05w_e2e_stages.txt; host-side: +-5 % between runs)"}
    try:
        out["small_text_calls"] = small_text_calls(s2, name, props, dev)
    except Exception as e:  # noqa: BLE001
        out["small_text_calls"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    try:
        out["small_image_calls"] = small_image_calls(s2, name, props, dev, dev_tensors[:64])
    except Exception as e:  # noqa: BLE001
        out["small_image_calls"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    s2.clear_loaded_models()
    return out


def _small_calls(s2, name, kw, make_content, threads, calls):
    """`threads` request threads x `calls` synchronous vectorise_ndarray() calls of 1 and of 4 items: requests/s, embeddings/s, the median / p95 latency a
    thread sees, and one thread calling alone.  No device-wide synchronise anywhere near the threads (on ROCm one issued while another thread captures a
    hipGraph invalidates the capture)"""
    res = {}
    for items in (1, 4):
        content = {(t, c): make_content(t, c, items) for t in range(threads) for c in range(calls)}
        lat, errors = [], []
        start = threading.Barrier(threads + 1)

        def worker(t):
            try:
                s2.vectorise_ndarray(name, content[(t, 0)], **kw)
                start.wait()
                for c in range(calls):
                    t0 = time.perf_counter()
                    s2.vectorise_ndarray(name, content[(t, c)], **kw)
                    lat.append(time.perf_counter() - t0)
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
                start.abort()
        ts = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
        for t in ts:
            t.start()
        try:
            start.wait()
        except threading.BrokenBarrierError:
            pass
        t0 = time.perf_counter()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        if errors:
            raise errors[0]
        lat.sort()
        t0 = time.perf_counter()
        for c in range(calls):
            s2.vectorise_ndarray(name, content[(0, c)], **kw)
        alone = (time.perf_counter() - t0) / calls
        res[f"{items}_per_call"] = {"requests_per_s": round(threads * calls / dt, 1), "embeddings_per_s": round(threads * calls * items / dt, 1),
                                    "latency_p50_ms": round(lat[len(lat) // 2] * 1e3, 3), "latency_p95_ms": round(lat[int(len(lat) * 0.95)] * 1e3, 3),
                                    "one_thread_ms_per_call": round(alone * 1e3, 3), "one_thread_embeddings_per_s": round(items / alone, 1)}
    return res


def _queue_summary(tower):
    st = getattr(tower, "queue_stats", lambda: {})().get(True)
    if not st:
        return None
    return {"requests": st["requests"], "tower_calls": st["calls"], "sequences_per_call": round(st["sequences"] / max(st["calls"], 1), 2),
            "largest_call": st["max_call_sequences"], "graph_replays": st.get("graph_replays", 0), "failed_calls": st["failed_calls"]}


def small_text_calls(s2, name, props, dev, threads=16, calls=60):
    """the reference's serving load on the text side: 8 indexing + 8 search request threads (api/configs.py:27-28), each calling vectorise() with ONE
    query (search) or 4 texts (the chunks of a document field) at a time — merged across threads by the tower's native request queue (mq_queue_*,
    csrc/queue.hip)"""
    from marqo_amd.engine import native_queue as NQ
    from marqo_amd.s2_inference.enums import AvailableModelsKey, Modality
    kw = dict(device=dev, modality=Modality.TEXT, model_properties=props)
    words = ["alpha", "beta", "gamma", "delta", "marqo", "tensor", "search", "image", "text", "vector"]
    rng = np.random.default_rng(0)
    s2.vectorise_ndarray(name, ["warm"], **kw)
    res = {"model": name, "threads": threads, "calls_per_thread": calls, "native_queue": bool(NQ.ENABLED), "queue_depth": NQ.DEPTH, "queue_max_seqs": NQ.MAX_SEQS}
    res.update(_small_calls(s2, name, kw, lambda t, c, items: [" ".join(words[int(j)] for j in rng.integers(0, 10, 12)) + f" {t} {c} {i}" for i in range(items)],
                            threads, calls))
    model = s2.get_available_models()[s2._create_model_cache_key(name, dev, props)][AvailableModelsKey.model]
    q = _queue_summary(model.text)
    if q:
        res["queue"] = q
    return res


def small_image_calls(s2, name, props, dev, views, threads=16, calls=40):
    """the indexing threads' image side: vectorise() per document field with the one to four tensors `.preprocess` returned (add_docs.py:129-141) — their
    device addresses go to the image tower's queue (MQ_QUEUE_IMAGE_F32)"""
    from marqo_amd.engine import native_queue as NQ
    from marqo_amd.s2_inference.enums import AvailableModelsKey, Modality
    kw = dict(device=dev, modality=Modality.IMAGE, model_properties=props)
    for k in (1, 4):
        s2.vectorise_ndarray(name, views[:k], **kw)
    res = {"model": name, "threads": threads, "calls_per_thread": calls, "content": "fp32 [3, 224, 224] device tensors from .preprocess",
           "native_queue": bool(NQ.ENABLED and NQ.IMAGE_REQUEST_MAX > 0)}
    res.update(_small_calls(s2, name, kw, lambda t, c, items: [views[(7 * t + 3 * c + i) % len(views)] for i in range(items)], threads, calls))
    model = s2.get_available_models()[s2._create_model_cache_key(name, dev, props)][AvailableModelsKey.model]
    q = _queue_summary(model.vision)
    if q:
        res["queue"] = q
    return res


# ---- BASELINE configs[3] in miniature ------------------------------------------------------------------------------------------------
def run_ingest(args, dev, rank, world, dist, lib, L):
    from PIL import Image
    os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
    os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
    from marqo_amd.engine import archs
    from marqo_amd.ingest import BulkVectoriser
    from marqo_amd.s2_inference import s2_inference as s2
    from marqo_amd.s2_inference.enums import Modality
    wl = WORKLOADS["add_documents_mixed"]
    docs = args.batch or wl["batch"]
    name = "open_clip/ViT-B-32/laion2b_s34b_b79k"
    varch, tarch = archs.resolve_open_clip("ViT-B-32")
    rng = np.random.default_rng(7)     # every rank builds the same request (in production: the same request broadcast by the API layer)
    imgs = [Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(docs)]
    words = ["alpha", "beta", "gamma", "delta", "marqo", "tensor", "search", "image", "text", "vector"]
    texts = [" ".join(words[int(j) % 10] for j in rng.integers(0, 10, int(rng.integers(3, 60)))) + f" {i}" for i in range(docs)]
    bv = BulkVectoriser(name, dev)

    def step():
        for i in range(docs):
            bv.add((i, "t"), texts[i])
            bv.add((i, "i"), imgs[i], Modality.IMAGE)
        return bv.flush()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, out = timed(step, args.steps, args.warmup, fence)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert len(out) == 2 * docs
    n_emb = 2 * docs
    value = n_emb * args.steps / elapsed
    gf = (varch.gflop_per_image + tarch.gflop_per_text(30)) / 2
    result = {
        "metric": "embeddings/sec", "value": round(value, 1), "unit": "embeddings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": wl["desc"], "global_batch": n_emb, "docs_per_request": docs,
                   "parallelism": f"dp{world} (replicated weights; texts sharded by token estimate, images contiguously; ONE RCCL all_gather per modality from HBM; "
                                  f"every rank returns the request's embeddings in order)",
                   "weights": "random-init (seed 0) ViT-B-32", "gflop_per_embedding": round(gf, 3)},
        "e2e_tflops": round(value * gf / 1e3, 1),
        # the dominant kernel family over a few more requests (HIP events on the launch streams, as in the headline)
        "roofline": gemm_roofline(lib, L, step, min(args.steps, 6), args.precision, "add_documents_mixed"),
        "cpu_baseline": (mixed_request_cpu_baseline(varch, tarch, imgs, docs, args.cpu_seconds)
                         if rank == 0 and world == 1 and not args.no_cpu_baseline else None),
        "note": "end-to-end through the Python boundary (host PIL -> uint8 pack -> H2D -> K10 -> towers -> gather -> D2H -> per-key rows); "
                "the request (total work) is fixed, ranks split it: strong scaling",
    }
    if rank == 0:
        print(json.dumps(result), flush=True)


# ---- BASELINE configs[3] as a stream: ranks own whole requests, one gather onto rank 0 ----------------------------------------------
def run_stream(args, dev, rank, world, dist, lib, L):
    from PIL import Image
    os.environ["MARQO_AMD_SYNTHETIC_WEIGHTS"] = "1"
    os.environ.setdefault("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
    from marqo_amd.engine import archs
    from marqo_amd.ingest import RequestShardedIngest
    from marqo_amd.s2_inference import s2_inference as s2
    from marqo_amd.s2_inference.enums import Modality
    wl = WORKLOADS["add_documents_stream"]
    docs = args.batch or wl["batch"]
    if args.steps <= 0:             # the stream at BASELINE configs[3]'s stated size: 100 000 documents over the ranks
        args.steps = -(-100000 // (docs * world))
    name = "open_clip/ViT-B-32/laion2b_s34b_b79k"
    varch, tarch = archs.resolve_open_clip("ViT-B-32")
    # a pool of 4 distinct synthetic requests per rank (4 x 128 images = 77 MB of pixels), cycled: the stream's requests are independent
    rng = np.random.default_rng(100 + rank)
    words = ["alpha", "beta", "gamma", "delta", "marqo", "tensor", "search", "image", "text", "vector"]
    pool = []
    for r in range(4):
        imgs = [Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)) for _ in range(docs)]
        texts = [" ".join(words[int(j) % 10] for j in rng.integers(0, 10, int(rng.integers(3, 60)))) + f" {r} {i}" for i in range(docs)]
        pool.append((texts, imgs))
    ing = RequestShardedIngest(name, dev)
    if getattr(args, "merge_images", None) is not None:
        ing.merge_images = args.merge_images
    group = max(1, -(-ing.merge_images // docs)) if ing.merge_images > 0 else 1       # requests per merged tower call
    state = {"next": rank}          # this rank's next request index (rank, rank + world, ...)

    def step():                     # ONE owned request of `docs` documents = 2 * docs embeddings
        i = state["next"]
        state["next"] += world
        texts, imgs = pool[(i // world) % len(pool)]
        items = [((i, d, "t"), texts[d], Modality.TEXT) for d in range(docs)] + [((i, d, "i"), imgs[d], Modality.IMAGE) for d in range(docs)]
        ing.submit(i, items)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2 * group)):
        step()
    ing.collect()
    fence()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    rows = ing.collect()            # the stream's ONE data-path collective: gather onto rank 0
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_emb = 2 * docs
    if rank == 0:
        assert len(rows) == args.steps * world and all(len(v) == n_emb for v in rows.values()), (len(rows), args.steps, world)
    value = n_emb * world * args.steps / elapsed
    memory = {"device_peak_allocated_mb": round(torch.cuda.max_memory_allocated() / 2**20, 1),
              "device_peak_reserved_mb": round(torch.cuda.max_memory_reserved() / 2**20, 1)}
    try:
        hs = torch.cuda.host_memory_stats()
        memory["pinned_peak_allocated_mb"] = round(hs.get("allocated_bytes.peak", 0) / 2**20, 1)
        memory["pinned_reserved_mb"] = round(hs.get("reserved_bytes.current", hs.get("reserved_bytes.peak", 0)) / 2**20, 1)
    except Exception:  # noqa: BLE001 - statistics of the pinned-memory allocator are not in every build
        pass
    # roofline of the dominant kernel family over a few more GROUPS (HIP events on the launch streams, as in the headline)
    # (groups one at a time, both modalities on the caller's thread: with a group in flight the two towers' kernels share the GPU and every
    # family's HIP-event time would count the other tower's kernels too)
    ing.collect()
    ing.pipeline_depth, ing._bulk.two_threads = 0, False

    def group_step():
        for _ in range(group):
            step()
        ing.drain()
    prof_groups = max(2, min(args.steps // group, 6))
    roofline = gemm_roofline(lib, L, group_step, prof_groups, args.precision, "add_documents_stream")
    ing.collect()
    ing.pipeline_depth, ing._bulk.two_threads = 1, True
    ing.close()
    roofline["requests_per_profiled_step"] = group
    gf = (varch.gflop_per_image + tarch.gflop_per_text(30)) / 2
    result = {
        "metric": "embeddings/sec", "value": round(value, 1), "unit": "embeddings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": wl["desc"], "global_batch": n_emb * world, "docs_per_request": docs, "documents_in_stream": docs * world * args.steps,
                   "micro_batching": (f"requests merged across the stream until >= {ing.merge_images} images or >= {ing.merge_text_tokens} estimated text tokens "
                                      f"({group} requests per tower call here) or a {ing.merge_deadline_ms} ms deadline" if ing.merge_images > 0 else "off: one tower call per request and modality"),
                   "inputs": "host PIL images + strings (the boundary add_documents hands over); host packing, H2D and the final D2H are INSIDE the timed "
                             "region — this workload is end-to-end by definition, the tower-only figure is the headline workload's",
                   "parallelism": f"dp{world}: replicated weights; request i belongs to rank i % {world}; no collective inside a request; ONE gather "
                                  f"(dist.gather, not all_gather) of the ranks' [n, 512] rows onto rank 0 at the end of the stream",
                   "weights": "random-init (seed 0) ViT-B-32", "gflop_per_embedding": round(gf, 3)},
        "e2e_tflops": round(value * gf / 1e3, 1), "roofline": roofline,
    }
    busy = sum(f["ms_per_step"] for f in roofline.get("per_family", {}).values()) / group
    result["gpu_busy_ms_per_step"] = round(busy, 4)               # sum of the kernel families' HIP-event time per request (groups run one at a time)
    result["memory"] = memory
    result["gpu_idle_share"] = round(max(0.0, 1.0 - busy / (elapsed / args.steps * 1e3)), 4)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the CPU path runs the SAME request (its images, its texts through the loaded model's tokeniser), so the error below compares like with like
        model, _ = s2.load_multimodal_model_and_get_preprocessors(name, s2.get_model_properties_from_registry(name), dev)
        ids = torch.from_numpy(np.asarray(model.tokenizer(pool[0][0])).astype(np.int64))
        base, emb_cpu = mixed_request_cpu_baseline(varch, tarch, pool[0][1], docs, args.cpu_seconds, ids=ids, return_rows=True)
        result["cpu_baseline"] = base
        # a timed request that carried pool[0] — the documents the CPU path just ran
        i0 = next((i for i in sorted(rows or {}) if (i // world) % len(pool) == 0), None)
        req0 = rows.get(i0) if i0 is not None else None
        if req0 is not None:
            k = emb_cpu.shape[0] // 2      # CPU rows: image 0, text 0, image 1, text 1, ...
            gpu = np.stack([req0[(i0, d, m)] for d in range(k) for m in ("i", "t")])
            result["cos_err_vs_cpu"] = _cos_err(torch.from_numpy(gpu).float(), emb_cpu)
    elif rank == 0:
        result["cpu_baseline"] = None
    return result if rank == 0 else None


def mixed_request_cpu_baseline(varch, tarch, imgs, docs, cpu_seconds, ids=None, return_rows=False):
    """reference-equivalent CPU path on a bounded sample of one {text, image} request: fp32 towers, 16-item batches (tokeniser time not included)"""
    from oracle import towers as O
    from marqo_amd.engine import synthetic
    sd = synthetic.random_open_clip_state_dict(vision=varch, text=tarch, seed=0)
    vcfg = O.VitConfig(varch.image_size, varch.patch_size, varch.width, varch.layers, varch.heads, varch.mlp_dim, varch.out_dim, varch.quick_gelu)
    tcfg = O.ClipTextConfig(tarch.vocab, tarch.ctx, tarch.width, tarch.layers, tarch.heads, tarch.mlp_dim, tarch.out_dim, tarch.quick_gelu)
    px = torch.from_numpy(np.stack([np.asarray(im) for im in imgs]))
    if ids is None:
        gt = torch.Generator().manual_seed(3)
        ids = torch.zeros(docs, 77, dtype=torch.int64)
        for i in range(docs):
            li = int(torch.randint(5, 62, (1,), generator=gt))
            ids[i, 0], ids[i, 1:1 + li], ids[i, 1 + li] = 49406, torch.randint(1, 49406, (li,), generator=gt), 49407

    def run_items(lo, hi):      # item 2i = image i, item 2i + 1 = text i: a bounded sample keeps the 50 / 50 mix
        i0, i1 = lo // 2, hi // 2
        im = O.vit_forward(sd, vcfg, O.preprocess_u8_exact_size(px[i0:i1]))
        tx = O.clip_text_forward(sd, tcfg, ids[i0:i1])
        return torch.stack([im, tx], dim=1).reshape(-1, im.shape[1])
    rate, n, emb, th, cores = cpu_baseline_run(run_items, 2 * docs, cpu_seconds)
    base = _baseline_dict(rate, n, th, cores, f"items ({n // 2} images + {n // 2} texts of 5..61 tokens) of one request")
    return (base, emb) if return_rows else base


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if args.workload == "stub":
        return run_stub(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    from marqo_amd import _lib as L
    from marqo_amd.parallel import gather_embeddings
    lib = L.load()

    if WORKLOADS[args.workload]["kind"] == "stream":
        result = run_stream(args, dev, rank, world, dist, lib, L)
        if rank == 0:
            print(json.dumps(result), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if WORKLOADS[args.workload]["kind"] == "ingest":
        run_ingest(args, dev, rank, world, dist, lib, L)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.workload.startswith("vit_l14_chunked_fp8"):
        args.precision = "fp8"     # the workload IS the fp8 configuration; its bf16 twin is timed beside it (`bf16_twin`)
    w = Workload(args.workload, args.precision, args.batch, dev, 1234 + rank)
    batch, kind, wl = w.batch, w.kind, w.wl

    def step():
        emb = w.run()                          # [batch, D] fp32 on device
        if world > 1:
            emb = gather_embeddings(emb)       # RCCL all_gather of the shards (final concat)
        return emb

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, out = timed(step, args.steps, args.warmup, fence)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = batch * world * args.steps / elapsed

    roofline = gemm_roofline(lib, L, w.run, min(args.steps, 10), args.precision, args.workload)
    peak = roofline["peak"]
    e2e_tflops = value * w.gflop_per_emb / 1e3
    result = {
        "metric": "embeddings/sec", "value": round(value, 1), "unit": "embeddings/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": wl["desc"], "global_batch": batch * world,
                   "inputs": "resident in HBM when the timed region starts (tower-only); host-resident hand-over is reported under e2e_vectorise",
                   "gflop_per_embedding": round(w.gflop_per_emb, 3),
                   "parallelism": f"dp{world} (replicated weights, sharded items, RCCL all_gather of embeddings)",
                   "weights": ("trained-like statistics (engine/synthetic.py, seed 2) " if wl.get("weights") == "trained_like" else "random-init (seed 0) ") + wl["arch"],
                   # transparency: the towers run the out-projection / MLP of the LAST block only on the pooled rows (class token /
                   # EOT): dead-row elimination with bit-identical embeddings (tests/test_towers_gpu.py::test_row_selected_*), 5.8 % of
                   # ViT-B/32's GEMM FLOPs.  e2e_tflops counts the full algorithmic FLOPs per embedding;
                   # roofline.achieved counts only the FLOPs of the GEMMs actually launched.  MQ_ROW_SELECT=0 runs every row.
                   "last_block_rows": "pooled" if os.environ.get("MQ_ROW_SELECT", "1") != "0" and kind != "bert" else "all",
                   **({"fp8_policy": w.fp8_policy} if w.fp8_policy else {})},
        "e2e_tflops": round(e2e_tflops, 1), "e2e_frac_of_peak": round(e2e_tflops / (peak * world), 4),
        "roofline": roofline,
    }

    solo = rank == 0 and world == 1
    if kind == "chunked":
        result["config"]["crops_per_image"] = w.crops_per_image
        result["config"]["inputs"] = "source images resident in HBM when the timed region starts; the chunker (K11) and the tower are inside it"
        if solo:   # the same step on the bf16 tower: what the fp8 policy buys
            twin = Workload(args.workload, "bf16", args.batch, dev, 1234 + rank)
            el, o16 = timed(twin.run, max(3, args.steps // 2), 2, torch.cuda.synchronize)
            v16 = twin.batch * max(3, args.steps // 2) / el
            result["bf16_twin"] = {"value": round(v16, 1), "unit": "embeddings/s", "fp8_over_bf16": round(value / v16, 3),
                                   "cos_err_fp8_vs_bf16": _cos_err(out.float().cpu(), o16.float().cpu())}
            del twin
    if solo and not args.no_cpu_baseline:
        base, cos = w.cpu_baseline(out, args.cpu_seconds)
        if base is not None:
            result["cpu_baseline"] = base
            result["cos_err_vs_cpu"] = cos
    if solo and not args.no_extras and args.workload == "vit_b32_image" and args.precision == "bf16":
        try:
            result["e2e_vectorise"] = e2e_vectorise(dev, w.images_cpu, value)
        except Exception as e:  # noqa: BLE001 - never lose the headline line to an extras failure
            result["e2e_vectorise"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        also = []
        del w
        torch.cuda.empty_cache()
        for name in (() if args.no_also else ALSO_DEFAULT):
            try:
                if WORKLOADS[name]["kind"] == "stream":      # BASELINE configs[3]: end-to-end by definition (host PIL + strings in, host rows out)
                    sa = argparse.Namespace(**{**vars(args), "steps": 0, "warmup": 8, "batch": 0, "cpu_seconds": 8.0, "workload": name})   # steps 0 = the full 100 000 documents
                    r = run_stream(sa, dev, 0, 1, None, lib, L)
                    also.append({"workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"], "documents_in_stream": r["config"]["documents_in_stream"],
                                 "micro_batching": r["config"]["micro_batching"], "memory": r.get("memory"),
                                 "gflop_per_embedding": r["config"]["gflop_per_embedding"], "e2e_tflops": r["e2e_tflops"], "gemm_tflops": r["roofline"]["achieved"],
                                 "gemm_frac": r["roofline"]["frac"], "roofline": r["roofline"], "gpu_busy_ms_per_step": r["gpu_busy_ms_per_step"],
                                 "gpu_idle_share": r["gpu_idle_share"], "cpu_baseline": r.get("cpu_baseline"), "cos_err_vs_cpu": r.get("cos_err_vs_cpu")})
                    from marqo_amd.s2_inference import s2_inference as _s2
                    _s2.clear_loaded_models()
                    torch.cuda.empty_cache()
                    continue
                prec = "fp8" if name.startswith("vit_l14_chunked_fp8") else args.precision     # configs[4] IS the fp8 configuration
                x = Workload(name, prec, 0, dev, 1234)
                el, o = timed(x.run, 10, 3, torch.cuda.synchronize)
                v = x.batch * 10 / el
                rf = gemm_roofline(lib, L, x.run, 5, prec, name)
                row = {"workload": WORKLOADS[name]["desc"], "value": round(v, 1), "unit": "embeddings/s", "ms_per_step": round(el / 10 * 1e3, 4),
                       "steps": 10, "warmup": 3, "dtype": prec, "gflop_per_embedding": round(x.gflop_per_emb, 3),
                       "e2e_tflops": round(v * x.gflop_per_emb / 1e3, 1), "gemm_tflops": rf["achieved"], "gemm_frac": rf["frac"]}
                if x.kind == "chunked":
                    row["roofline"] = rf
                    row["fp8_policy"] = x.fp8_policy
                    twin = Workload(name, "bf16", 0, dev, 1234)        # the same step on the bf16 tower: what the policy buys, and at what error
                    el16, o16 = timed(twin.run, 5, 2, torch.cuda.synchronize)
                    v16 = twin.batch * 5 / el16
                    row["bf16_twin"] = {"value": round(v16, 1), "unit": "embeddings/s", "fp8_over_bf16": round(v / v16, 3),
                                        "cos_err_fp8_vs_bf16": _cos_err(o.float().cpu(), o16.float().cpu())}
                    del twin
                if not args.no_cpu_baseline:
                    base, cos = x.cpu_baseline(o, 8.0)
                    row["cpu_baseline"], row["cos_err_vs_cpu"] = base, cos
                also.append(row)
                del x
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                also.append({"workload": name, "error": f"{type(e).__name__}: {e}"[:300]})
        result["also"] = also
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
