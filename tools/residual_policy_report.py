"""What the load-time residual-stream policy measures on the fixtures: bf16-vs-fp32-stream error on the calibration batch, and each form's
error against the fp32 CPU oracle on held-out inputs.  python tools/residual_policy_report.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marqo_amd.engine import archs, towers
from oracle import towers as O


def cos_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((1 - (a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))).max())


def main():
    os.environ["MARQO_AMD_RESIDUAL_STREAM"] = "auto"
    for arch_name, cfg in (("ViT-B-32", O.VitConfig(224, 32, 768, 12, 12, 3072, 512)), ("ViT-L-14", O.VitConfig(224, 14, 1024, 24, 16, 4096, 768))):
        varch, _ = archs.resolve_open_clip(arch_name)
        for weights in ("plain", "realistic"):
            sd = O.synthetic_vit_state_dict(cfg, 0) if weights == "plain" else O.synthetic_vit_state_dict_realistic(cfg, 0)
            u8 = O.synthetic_images_u8(4, 224, seed=9)
            ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8))
            t = towers.VitTower(varch, sd, "cuda:0")
            row = f"{arch_name} image {weights}: policy {t.residual_stream} calibration bf16-vs-fp32 stream {t.residual_stream_error:.2e}"
            for mode in (2, 1):
                t.cfg.enc.residual_stream = mode
                row += f" | {'fp32' if mode == 2 else 'bf16'} stream vs oracle {cos_err(t.encode_u8(u8.to('cuda:0')), ref):.2e}"
            print(row, flush=True)
    for name, tcfg, an in (("CLIP text L/14", O.ClipTextConfig(49408, 77, 768, 12, 12, 3072, 768), "ViT-L-14"), ("CLIP text B/32", O.ClipTextConfig(49408, 77, 512, 12, 8, 2048, 512), "ViT-B-32")):
        _, tarch = archs.resolve_open_clip(an)
        for weights in ("plain", "realistic"):
            sd = O.synthetic_clip_text_state_dict(tcfg, 0) if weights == "plain" else O.synthetic_clip_text_state_dict_realistic(tcfg, 0)
            ids = O.synthetic_clip_ids(12, seed=6)
            ref = O.clip_text_forward(sd, tcfg, ids)
            t = towers.ClipTextTower(tarch, sd, "cuda:0")
            row = f"{name} {weights}: policy {t.residual_stream} calibration bf16-vs-fp32 stream {t.residual_stream_error:.2e}"
            for mode in (2, 1):
                t.cfg.enc.residual_stream = mode
                row += f" | {'fp32' if mode == 2 else 'bf16'} stream vs oracle {cos_err(t.encode_ids(ids), ref):.2e}"
            print(row, flush=True)


if __name__ == "__main__":
    main()
