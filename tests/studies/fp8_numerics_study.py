"""Numerics study behind the fp8 policy (engine/towers.py: tune_fp8) — CPU simulation with the oracle towers (test infrastructure: lives
under tests/ because it imports oracle/).  Re-creates profiles/r02_fp8_numerics_sim.txt:

    python tests/studies/fp8_numerics_study.py > profiles/r02_fp8_numerics_sim.txt

ViT-L/14 (24 blocks) on synthetic images, plain and realistic-statistics weights (oracle/towers.py), 1 - cos against the fp32 oracle:
  * every GEMM operand on e4m3 under per-tensor / per-row / MX (32-element block exponent) scaling: the error is set by the e4m3 mantissa
    (3 bits), not by the scaling granularity;
  * which GEMMs / which blocks carry the error: early blocks cost 2-7x more than late ones -> the shipped policy keeps the first blocks on
    bf16 and moves the LAST blocks to e4m3 until a cos-error budget is reached;
  * int8 (per-row / per-128-block symmetric) for comparison, and the bf16 residual stream (mq_tune("residual_bf16", 1)).
"""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fp8_sim as S  # noqa: E402
from oracle import towers as O  # noqa: E402

torch.set_num_threads(int(os.environ.get("STUDY_THREADS", "8")))
N_IMAGES = int(os.environ.get("STUDY_IMAGES", "4"))


def _int8(x, mode):
    if mode == "int8row":
        s = (x.abs().amax(dim=-1, keepdim=True) / 127).clamp_min(1e-30)
        return torch.round(x / s).clamp(-127, 127) * s
    *lead, K = x.shape
    xb = x.reshape(*lead, K // 128, 128)
    s = (xb.abs().amax(dim=-1, keepdim=True) / 127).clamp_min(1e-30)
    return (torch.round(xb / s).clamp(-127, 127) * s).reshape(x.shape)


_orig_quantize = S.quantize


def _quantize(x, mode, static_scale=None):
    return _int8(x, mode) if mode.startswith("int8") else _orig_quantize(x, mode, static_scale)


S.quantize = _quantize
bf = lambda t: t.to(torch.bfloat16).float()


def vit_bf16(sd, cfg, px, res_bf16):
    """bf16 GEMM operands, fp32 (or bf16) residual stream — the arithmetic of the shipped bf16 path"""
    W, heads = cfg.width, cfg.heads
    lin = lambda a, w, b: F.linear(bf(a), bf(w), b)
    x = F.conv2d(bf(px), bf(sd["visual.conv1.weight"]), None, stride=cfg.patch_size)
    B = x.shape[0]
    x = x.reshape(B, W, -1).permute(0, 2, 1)
    x = torch.cat([sd["visual.class_embedding"].expand(B, 1, W), x], 1) + sd["visual.positional_embedding"]
    x = F.layer_norm(x, (W,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], cfg.ln_eps)
    x = bf(x) if res_bf16 else x
    T, hd = x.shape[1], W // heads
    for i in range(cfg.layers):
        p = f"visual.transformer.resblocks.{i}."
        h = F.layer_norm(x, (W,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], cfg.ln_eps)
        q, k, v = bf(lin(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"])).split(W, dim=-1)
        q, k, v = (t.view(B, T, heads, hd).transpose(1, 2) for t in (q, k, v))
        o = (torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), -1) @ v).transpose(1, 2).reshape(B, T, W)
        x = x + lin(o, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
        x = bf(x) if res_bf16 else x
        h = F.layer_norm(x, (W,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], cfg.ln_eps)
        h = O._act(lin(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]), cfg.quick_gelu)
        x = x + lin(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
        x = bf(x) if res_bf16 else x
    pooled = F.layer_norm(x[:, 0], (W,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], cfg.ln_eps)
    return O.l2_normalize_clip(bf(pooled) @ bf(sd["visual.proj"]))


def main():
    cfg = O.VitConfig(224, 14, 1024, 24, 16, 4096, 768)
    px = O.preprocess_u8_exact_size(O.synthetic_images_u8(N_IMAGES, 224, 0))
    run = lambda sd, pol: S.cos_err(S.vit_forward_fp8(sd, cfg, px, pol), ref)
    print(f"# ViT-L/14, 24 blocks, {N_IMAGES} synthetic images; max over images of 1 - cos against the fp32 oracle")
    for name, sd in (("plain", O.synthetic_vit_state_dict(cfg, 0)), ("realistic", O.synthetic_vit_state_dict_realistic(cfg, 0))):
        ref = O.vit_forward(sd, cfg, px)
        print(f"== {name} weights")
        print("bf16 operands, fp32 residual stream            %.2e" % S.cos_err(vit_bf16(sd, cfg, px, False), ref), flush=True)
        print("bf16 operands, bf16 residual stream            %.2e" % S.cos_err(vit_bf16(sd, cfg, px, True), ref), flush=True)
        for act in ("tensor", "row", "mx"):
            for wt in ("row", "mx"):
                print("all GEMMs e4m3, act=%-6s weight=%-3s          %.2e" % (act, wt, run(sd, S.Fp8Policy(act=act, weight=wt))), flush=True)
        for g in (("qkv", "out"), ("fc1", "fc2"), ("qkv",), ("out",), ("fc1",), ("fc2",)):
            print("e4m3 (act mx, weight row) on %-16s  %.2e" % (",".join(g), run(sd, S.Fp8Policy(gemms=g, act="mx", weight="row"))), flush=True)
        for rng, lbl in ((range(0, 6), "0..5"), (range(0, 12), "0..11"), (range(12, 24), "12..23"), (range(18, 24), "18..23")):
            print("e4m3 (act mx, weight row) blocks %-8s       %.2e" % (lbl, run(sd, S.Fp8Policy(act="mx", weight="row", layers=list(rng)))), flush=True)
        for act, wt in (("int8row", "int8row"), ("int8blk", "int8row")):
            print("all GEMMs int8, act=%-8s weight=%-8s    %.2e" % (act, wt, run(sd, S.Fp8Policy(act=act, weight=wt))), flush=True)


if __name__ == "__main__":
    main()
