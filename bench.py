#!/usr/bin/env python
"""bench.py — embeddings/sec of the vectorise() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], the config the metric is quoted on): open_clip ViT-B/32 image
tower, synthetic uint8 224x224 RGB images already resident in HBM, 256 images per GPU per step,
random-init weights of that architecture (no network for checkpoints).  One "step" = one pass of the
hot path over one batch: normalise + patchify -> patch-embed GEMM -> 12 pre-LN blocks -> ln_post ->
projection -> L2, i.e. exactly mq_encode_image_u8; for N > 1 the [256, 512] fp32 shards are then
all-gathered over RCCL (the only collective on the path).  Work per GPU is fixed -> weak scaling.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline     — the dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs / HIP-event time of its launches
  cpu_baseline — the CPU fp32 oracle (oracle/towers.py) run with the reference's 16-item batch loop
                 (s2_inference.py:135-146) on a bounded sample, all host cores
  cos_err_vs_cpu — max (1 - cosine) of the GPU embeddings vs that CPU path on the same sample
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
FP8_DENSE_PEAK_TFLOPS = 5000.0   # MX-scaled fp8 MFMA (K = 128), ~5 PF dense
WORKLOADS = {
    # the headline workload (BASELINE.json configs[1]) is the default; the others are reported in DESIGN.md §6
    "vit_b32_image": dict(kind="image", arch="ViT-B-32", desc="open_clip ViT-B/32 image tower, uint8 224x224, batch 256/GPU", batch=256),
    "vit_l14_image": dict(kind="image", arch="ViT-L-14", desc="open_clip ViT-L/14 image tower, uint8 224x224, batch 64/GPU", batch=64),
    "vit_h14_image": dict(kind="image", arch="ViT-H-14", desc="open_clip ViT-H/14 image tower (80-wide heads run as 96), uint8 224x224, batch 64/GPU", batch=64),
    "vit_bigg14_image": dict(kind="image", arch="ViT-bigG-14", desc="open_clip ViT-bigG/14 image tower (104-wide heads run as 112), uint8 224x224, batch 32/GPU", batch=32),
    "siglip_b16_image": dict(kind="image", arch="ViT-B-16-SigLIP", desc="open_clip ViT-B-16-SigLIP image tower (196 tokens, attention-pool head), uint8 224x224, batch 128/GPU", batch=128),
    "siglip_l16_384_image": dict(kind="image", arch="ViT-L-16-SigLIP-384", desc="open_clip ViT-L-16-SigLIP-384 image tower (576 tokens), uint8 384x384, batch 32/GPU", batch=32),
    "siglip_b16_text": dict(kind="clip_text", arch="ViT-B-16-SigLIP", desc="open_clip ViT-B-16-SigLIP text tower (unmasked, 64 positions), batch 1024/GPU", batch=1024),
    "clip_text_b32": dict(kind="clip_text", arch="ViT-B-32", desc="open_clip ViT-B/32 text tower, 77-token ids, batch 1024/GPU", batch=1024),
    "clip_text_l14": dict(kind="clip_text", arch="ViT-L-14", desc="open_clip ViT-L/14 text tower, 77-token ids, batch 1024/GPU", batch=1024),
    "bert_base_77": dict(kind="bert", arch="intfloat/e5-base-v2", desc="e5-base-v2 (BERT-base) + mean-pool + L2, 77-token ids, batch 1024/GPU", batch=1024),
    "vit_l14_mixed": dict(kind="mixed", arch="ViT-L-14", desc="open_clip ViT-L/14 dual encoder, 128 images + 128 texts (5..75 tokens) per GPU (BASELINE configs[2])", batch=256),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="vit_b32_image", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: the workload's)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp8"],
                    help="GEMM operand type of the encoder blocks (fp8 = e4m3, BASELINE config 5; not the headline metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def cpu_baseline(sd, arch, images_u8_cpu, target_seconds):
    """Reference-equivalent CPU path (fp32 PyTorch eager, the reference's 16-item batch loop) on a bounded sample.
    The thread count is the best of a short ladder (all host threads is often NOT the fastest on a 2-socket SMT box);
    returns (emb/s, n, embeddings, threads used, total host threads)."""
    import numpy as np
    from oracle import towers as O
    cores = os.cpu_count() or 1
    if arch.pool == "map":
        cfg, fwd = O.SiglipVitConfig(arch.image_size, arch.patch_size, arch.width, arch.layers, arch.heads, arch.mlp_dim), O.siglip_vit_forward
    else:
        cfg, fwd = O.VitConfig(arch.image_size, arch.patch_size, arch.width, arch.layers, arch.heads, arch.mlp_dim,
                               arch.out_dim, arch.quick_gelu), O.vit_forward

    def run(imgs):
        outs = []
        for i in range(0, imgs.shape[0], 16):  # MARQO_MAX_VECTORISE_BATCH_SIZE default (api/configs.py:38)
            outs.append(fwd(sd, cfg, O.preprocess_u8_exact_size(imgs[i:i + 16])).numpy())
        return np.concatenate(outs, axis=0)

    t_start = time.perf_counter()
    best_threads, best_dt = None, None
    for th in [t for t in (16, 32, 64, 128) if t < cores] + [cores]:
        torch.set_num_threads(th)
        run(images_u8_cpu[:16])  # warm-up at this thread count
        t0 = time.perf_counter()
        run(images_u8_cpu[:16])
        dt = time.perf_counter() - t0
        if best_dt is None or dt < best_dt:
            best_threads, best_dt = th, dt
        elif dt > 1.3 * best_dt:
            break
        if time.perf_counter() - t_start > 0.5 * target_seconds:
            break
    torch.set_num_threads(best_threads)
    budget = max(target_seconds - (time.perf_counter() - t_start), best_dt)
    n = int(min(images_u8_cpu.shape[0], max(16, (budget / max(best_dt, 1e-3)) * 16) // 16 * 16))
    t0 = time.perf_counter()
    emb = run(images_u8_cpu[:n])
    dt = time.perf_counter() - t0
    return n / dt, n, torch.from_numpy(emb), best_threads, cores


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
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
    from marqo_amd.engine import archs, synthetic, towers
    from marqo_amd.parallel import gather_embeddings

    wl = WORKLOADS[args.workload]
    batch = args.batch or wl["batch"]
    kind = wl["kind"]
    lib = L.load()
    g = torch.Generator().manual_seed(1234 + rank)
    varch = tarch = barch = None
    if kind in ("image", "clip_text", "mixed"):
        varch, tarch = archs.resolve_open_clip(wl["arch"])

    def clip_ids(n, lo, hi):
        ids = torch.zeros(n, 77, dtype=torch.int64)
        lens = torch.randint(lo, hi + 1, (n,), generator=g)
        for i in range(n):
            li = int(lens[i])
            ids[i, 0] = 49406
            ids[i, 1:1 + li] = torch.randint(1, 49406, (li,), generator=g)
            ids[i, 1 + li] = 49407
        return ids

    towers_used, images, images_cpu, sd = [], None, None, None
    if kind == "image":
        sd = synthetic.random_open_clip_state_dict(vision=varch, seed=0)
        tower = towers.VitTower(varch, sd, dev, precision=args.precision)
        images_cpu = torch.randint(0, 256, (batch, varch.image_size, varch.image_size, 3), generator=g, dtype=torch.uint8)
        images = images_cpu.to(dev)
        towers_used = [tower]
        gflop_per_emb = varch.gflop_per_image
        run_local = lambda: tower.encode_u8(images)
    elif kind == "clip_text":
        sd = synthetic.random_open_clip_state_dict(text=tarch, seed=0)
        tower = towers.ClipTextTower(tarch, sd, dev, precision=args.precision)
        towers_used = [tower]
        gflop_per_emb = tarch.gflop_per_text(tarch.ctx)
        # ids resident in HBM like the images (what the device tokeniser hands over); only the n lengths live on the host
        if tarch.causal:
            ids = clip_ids(batch, 75, 75)
            d_ids, lens = ids.to(torch.int32).to(dev), ids.argmax(1) + 1
        else:  # SigLIP: every text is ctx positions (pieces ... </s> then </s> padding), all of them run
            ids = torch.ones(batch, tarch.ctx, dtype=torch.int64)
            ids[:, :20] = torch.randint(2, tarch.vocab, (batch, 20), generator=g)
            d_ids, lens = ids.to(torch.int32).to(dev), torch.full((batch,), tarch.ctx, dtype=torch.int64)
        run_local = lambda: tower.encode_device(d_ids, lens)
    elif kind == "bert":
        barch = archs.HF_BERT_ARCHS[wl["arch"]]
        sd = synthetic.random_bert_state_dict(barch, seed=0)
        tower = towers.BertTower(barch, sd, dev, precision=args.precision)
        ids = torch.randint(1000, barch.vocab, (batch, 77), generator=g)
        ids[:, 0], ids[:, -1] = 101, 102
        towers_used = [tower]
        gflop_per_emb = barch.gflop_per_text(77)
        d_ids, lens = ids.to(torch.int32).to(dev), torch.full((batch,), 77, dtype=torch.int64)
        run_local = lambda: tower.encode_device(d_ids, lens)
    else:  # mixed: half images, half texts of ragged length through the two towers of one model
        sd = synthetic.random_open_clip_state_dict(vision=varch, text=tarch, seed=0)
        vt = towers.VitTower(varch, sd, dev, precision=args.precision)
        tt = towers.ClipTextTower(tarch, sd, dev, precision=args.precision)
        n_img = batch // 2
        images_cpu = torch.randint(0, 256, (n_img, varch.image_size, varch.image_size, 3), generator=g, dtype=torch.uint8)
        images = images_cpu.to(dev)
        ids = clip_ids(batch - n_img, 5, 75)
        towers_used = [vt, tt]
        d_ids, lens = ids.to(torch.int32).to(dev), ids.argmax(1) + 1
        mean_tokens = float((ids.argmax(1) + 1).float().mean())
        gflop_per_emb = (n_img * varch.gflop_per_image + (batch - n_img) * tarch.gflop_per_text(int(round(mean_tokens)))) / batch
        run_local = lambda: torch.cat([vt.encode_u8(images), tt.encode_device(d_ids, lens)], dim=0)
    tower = towers_used[0]

    if args.precision == "fp8":
        for t in towers_used:
            t.calibrate_fp8(run_local)  # static activation scales, outside the timed region

    def step():
        emb = run_local()                      # [batch, D] fp32 on device
        if world > 1:
            emb = gather_embeddings(emb)       # RCCL all_gather of the shards (final concat)
        return emb

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = batch * world * args.steps / elapsed

    # ---- roofline of the dominant kernel, HIP events on the launch stream (separate, instrumented steps) ----
    lib.mq_profile_enable(1)
    prof_steps = min(args.steps, 10)
    for _ in range(prof_steps):
        run_local()
    ms = (C.c_double * L.MQ_PROF_FAMILIES)()
    cnt = (C.c_int64 * L.MQ_PROF_FAMILIES)()
    flops = C.c_double(0.0)
    L.check(lib.mq_profile_collect(ms, cnt, C.byref(flops)), "mq_profile_collect")
    lib.mq_profile_enable(0)
    gemm_ms, gemm_launches = ms[0], cnt[0]
    achieved = flops.value / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    peak = FP8_DENSE_PEAK_TFLOPS if args.precision == "fp8" else BF16_DENSE_PEAK_TFLOPS
    families = {L.PROF_FAMILY_NAMES[i]: {"ms_per_step": ms[i] / prof_steps, "launches_per_step": cnt[i] // prof_steps}
                for i in range(L.MQ_PROF_FAMILIES) if cnt[i]}
    roofline = {
        "kernel": ("gemm_fp8_kernel (e4m3 MFMA 16x16x128 unit-scale MX, (32*MT)x128x128 tiles, fused epilogues)" if args.precision == "fp8"
                   else "gemm_nt_kernel (bf16 MFMA 16x16x32, (32*MT)x128x64 tiles, fused epilogues)"),
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": None,
        "flops_per_launch": flops.value / max(gemm_launches, 1),
        "avg_launch_us": gemm_ms * 1e3 / max(gemm_launches, 1),
        "launches_per_step": gemm_launches // prof_steps,
        "per_family": families,
    }
    if args.precision == "bf16":
        # informational: what a register-resident v_mfma_f32_16x16x32_bf16 burn sustains on this chip with random operands at
        # the clock it then holds (2.04 GHz) — tools/probes/mfma_peak.hip, profiles/r01f_mfma_sustained_peak.txt.  `peak` / `frac`
        # above stay priced against the guide's 2.4 GHz figure.
        roofline["peak_sustained_measured"] = 2114.0
        roofline["frac_of_sustained"] = round(achieved / 2114.0, 4)
    # HBM-side traffic of the dominant kernel from the committed PMC passes of this same command (tools/gpu_evidence.sh:
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH doubled per MI355X_MICROARCH.md; PMC cannot be sampled
    # from inside this process).  Bytes per launch, averaged over the GEMM launches like `achieved`.
    tpath = os.path.join(ROOT, "profiles", f"r01_traffic_{args.workload}_{args.precision}.json")
    if os.path.isfile(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            fam = "gemm_fp8_kernel" if args.precision == "fp8" else "gemm_nt_kernel"
            rows = [v for k, v in tj.items() if k.startswith(fam) or k.startswith("gemm_big_kernel")]
            n_l = sum(v["launches"] for v in rows)
            if n_l:
                roofline["traffic"] = round(sum(v["launches"] * (v["read_bytes"] + v["write_bytes"]) for v in rows) / n_l, 1)
                roofline["traffic_unit"] = "bytes/launch (fabric-side reads incl. Infinity-Cache hits + writes)"
                roofline["traffic_source"] = os.path.relpath(tpath, ROOT)
        except (OSError, ValueError, KeyError):
            pass
    e2e_tflops = value * gflop_per_emb / 1e3
    result = {
        "metric": "embeddings/sec", "value": round(value, 1), "unit": "embeddings/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": wl["desc"], "global_batch": batch * world,
                   "gflop_per_embedding": round(gflop_per_emb, 3),
                   "parallelism": f"dp{world} (replicated weights, sharded items, RCCL all_gather of embeddings)",
                   "weights": "random-init (seed 0) " + wl["arch"],
                   # transparency: the towers run the out-projection / MLP of the LAST block only on the pooled rows (class token /
                   # EOT): dead-row elimination with bit-identical embeddings (tests/test_towers_gpu.py::test_row_selected_*), 5.8 % of
                   # ViT-B/32's GEMM FLOPs.  e2e_tflops counts the full algorithmic FLOPs per embedding;
                   # roofline.achieved counts only the FLOPs of the GEMMs actually launched.  MQ_ROW_SELECT=0 runs every row.
                   "last_block_rows": "pooled" if os.environ.get("MQ_ROW_SELECT", "1") != "0" and kind != "bert" else "all"},
        "e2e_tflops": round(e2e_tflops, 1), "e2e_frac_of_peak": round(e2e_tflops / (peak * world), 4),
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline and kind == "image":
        cpu_rate, n_cpu, cpu_emb, cores, host_threads = cpu_baseline(sd, varch, images_cpu, args.cpu_seconds)
        gpu_emb = out[:n_cpu].float().cpu()
        cos = (gpu_emb.double() * cpu_emb.double()).sum(-1) / (gpu_emb.double().norm(dim=-1) * cpu_emb.double().norm(dim=-1))
        result["cpu_baseline"] = {"value": round(cpu_rate, 2), "unit": "embeddings/s", "cores": cores, "kind": "port",
                                  "sample": f"{n_cpu} of the step's {batch} images, fp32 PyTorch eager, 16-image batches "
                                            f"(reference loop s2_inference.py:135-146), {cores} of {host_threads} host threads "
                                            f"(best of a 16/32/64/128/all ladder)"}
        result["cos_err_vs_cpu"] = float((1 - cos).max())
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
