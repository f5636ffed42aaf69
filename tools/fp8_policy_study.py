"""The fp8 policy space of a pre-LN tower on the GPU: error vs the tower's own bf16 output (calibration batch) and embeddings/s for
(first fully-e4m3 block, number of MLP-only e4m3 blocks in front of it), then what the load-time search picks.
python tools/fp8_policy_study.py [--arch ViT-L-14] [--weights random|realistic]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marqo_amd import _lib as L
from marqo_amd.engine import archs, synthetic, towers
from oracle import towers as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="ViT-L-14")
    ap.add_argument("--weights", default="random")
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    varch, _ = archs.resolve_open_clip(args.arch)
    if args.weights == "realistic":
        cfg = O.VitConfig(varch.image_size, varch.patch_size, varch.width, varch.layers, varch.heads, varch.mlp_dim, varch.out_dim)
        sd = O.synthetic_vit_state_dict_realistic(cfg, 0)
    else:
        sd = synthetic.random_open_clip_state_dict(vision=varch, seed=0)
    t8 = towers.VitTower(varch, sd, "cuda:0", precision="fp8")
    t16 = towers.VitTower(varch, sd, "cuda:0", precision="bf16")
    cal = t8.calibration_images()
    u8 = torch.randint(0, 256, (args.batch, varch.image_size, varch.image_size, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to("cuda:0")
    t8.cfg.enc.fp8_first_layer, t8.cfg.enc.fp8_mlp_extra = 0, 0
    t8.calibrate_fp8(lambda: t8.encode_u8(cal), passes=2, margin=t8.FP8_SCALE_MARGIN)
    enc, layers = t8.cfg.enc, varch.layers
    enc.precision = L.MQ_PREC_BF16
    ref = t8.encode_u8(cal).double()
    enc.precision = L.MQ_PREC_FP8

    def rate(tower):
        for _ in range(3):
            tower.encode_u8(u8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            tower.encode_u8(u8)
        torch.cuda.synchronize()
        return args.batch * 8 / (time.perf_counter() - t0)
    r16 = rate(t16)
    print(f"{args.arch} {args.weights} weights: bf16 tower ({t16.residual_stream} residual stream) {r16:.0f} embeddings/s", flush=True)
    print("first  extra  e4m3 share   1-cos vs bf16   emb/s    x bf16")
    for first in sorted({0, layers // 6, layers // 3, layers // 2, (13 * layers) // 24, (2 * layers) // 3, (5 * layers) // 6, layers}):
        for extra in sorted({0, first // 2, first}):
            enc.fp8_first_layer, enc.fp8_mlp_extra = first, extra
            out = t8.encode_u8(cal).double()
            e = float((1 - (out * ref).sum(-1) / (out.norm(dim=-1) * ref.norm(dim=-1))).max())
            r = rate(t8)
            print(f"{first:5d}  {extra:5d}  {((layers - first) + 2 / 3 * extra) / layers:9.2f}   {e:12.2e}   {r:7.0f}   {r / r16:5.2f}", flush=True)
    first = t8.tune_fp8_default()
    r = rate(t8)
    print(f"policy (budget {t8.FP8_BUDGET:.1e}): blocks [{first}, {layers}) e4m3 + MLP halves of [{first - t8.fp8_mlp_extra}, {first}); calibration error "
          f"{t8.fp8_calibration_error:.2e}; {r:.0f} embeddings/s = x{r / r16:.2f} bf16; search trace (split, extra, error): {t8.fp8_policy_trace}")


if __name__ == "__main__":
    main()
