"""Generate the golden fixtures that PIN oracle/towers.py to an independent implementation.

Run in the build container (no GPU needed):  python tests/golden/make_golden.py

For each tower a SMALL model of the real architecture is instantiated from ``transformers``
(5.15 here; the reference pins 4.41.2 — same module arithmetic) with seeded random weights,
run on seeded inputs in fp32 on CPU, and the (weights, inputs, outputs) triple is stored as
``tests/golden/<name>.npz`` with the weights renamed to the checkpoint naming the loaders consume
(open_clip names for CLIP, HuggingFace names for BERT).

* ``bert_small``      transformers.BertModel  == what HuggingFaceModel loads through AutoModel
                      (hugging_face_model.py:125-130); pooled / normalised exactly as
                      hugging_face_model.py:187-214.
* ``clip_vit_small``  transformers.CLIPVisionModelWithProjection (gelu and quick_gelu variants)
* ``clip_text_small`` transformers.CLIPTextModelWithProjection (argmax-EOT pooling)
* ``siglip_small``    transformers.SiglipVisionModel (attention-pool head) + SiglipTextModel (no mask, last-token pooling,
                      biased projection), hidden_act='gelu' like timm's SigLIP ViTs; weights renamed to the open_clip / timm
                      checkpoint naming (visual.trunk.*, text.*)

Head dim is 64 in the fixtures except ``bert_small_h32`` (32-wide heads, zero-padded to 64 at load); BASELINE.json's four towers
all have 64-wide heads.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def _np(sd):
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in sd.items()}


def _jitter(model, seed):
    """HF init leaves LayerNorm at (1, 0) and biases at 0; perturb them so that a dropped bias or a
    swapped gamma/beta cannot pass the parity tests."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.mul_(3.0)  # HF init std 0.02 would make every block a near no-op


def make_bert(heads=2, name="bert_small.npz"):
    from transformers import BertConfig, BertModel
    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=heads,
                     intermediate_size=256, max_position_embeddings=64, layer_norm_eps=1e-12,
                     hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = BertModel(cfg).eval()
    _jitter(m, 1)
    g = torch.Generator().manual_seed(2)
    lens = [5, 17, 1 + 1, 33, 64, 9]
    S = max(lens)
    ids = torch.zeros(len(lens), S, dtype=torch.int64)
    mask = torch.zeros(len(lens), S, dtype=torch.int64)
    for i, L in enumerate(lens):
        ids[i, :L] = torch.randint(3, 300, (L,), generator=g)
        mask[i, :L] = 1
    with torch.no_grad():
        out = m(input_ids=ids, attention_mask=mask)
    last = out.last_hidden_state
    # hugging_face_model.py:205-209, 194-195
    lh = last.masked_fill(~mask[..., None].bool(), 0.0)
    mean = lh.sum(1) / mask.sum(1)[..., None]
    mean_n = torch.nn.functional.normalize(mean, p=2, dim=1)
    cls = last[:, 0]
    cls_n = torch.nn.functional.normalize(cls, p=2, dim=1)
    sd = {k: v for k, v in m.state_dict().items() if not k.startswith("pooler.")}
    np.savez_compressed(os.path.join(HERE, name),
                        ids=ids.numpy(), mask=mask.numpy(), last_hidden=last.numpy(),
                        mean=mean.numpy(), mean_norm=mean_n.numpy(), cls=cls.numpy(), cls_norm=cls_n.numpy(),
                        cfg=np.array([300, 64, 128, 2, heads, 256], dtype=np.int64),  # vocab max_pos W layers heads F
                        **{"w:" + k: v for k, v in _np(sd).items()})


def make_mpnet():
    """transformers.MPNetModel == what HuggingFaceModel loads through AutoModel for hf/all-mpnet-base-v1 / -v2 and the flax
    all_datasets_v{3,4}_mpnet-base entries: post-LN encoder, no token types, position ids offset by 2, one relative-position bias
    table shared by all layers.  Sequences up to 200 tokens so that the log-spaced buckets beyond the exact range (|d| >= 8) and the
    clamp at max_distance = 128 are all exercised."""
    from transformers import MPNetConfig, MPNetModel
    torch.manual_seed(0)
    cfg = MPNetConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                      max_position_embeddings=258, layer_norm_eps=1e-5, hidden_act="gelu", hidden_dropout_prob=0.0,
                      attention_probs_dropout_prob=0.0, relative_attention_num_buckets=32, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    m = MPNetModel(cfg).eval()
    _jitter(m, 5)
    with torch.no_grad():
        m.encoder.relative_attention_bias.weight.mul_(10.0)   # (a bias the softmax can feel: dropping it must fail the parity tests)
    g = torch.Generator().manual_seed(6)
    lens = [5, 17, 2, 33, 200, 9, 130]
    S = max(lens)
    ids = torch.ones(len(lens), S, dtype=torch.int64)          # pad id 1
    mask = torch.zeros(len(lens), S, dtype=torch.int64)
    for i, L in enumerate(lens):
        ids[i, :L] = torch.randint(4, 300, (L,), generator=g)
        ids[i, 0], ids[i, L - 1] = 0, 2
        mask[i, :L] = 1
    with torch.no_grad():
        last = m(input_ids=ids, attention_mask=mask).last_hidden_state
    lh = last.masked_fill(~mask[..., None].bool(), 0.0)
    mean = lh.sum(1) / mask.sum(1)[..., None]
    sd = {k: v for k, v in m.state_dict().items() if not k.startswith("pooler.") and not k.endswith("position_ids")}
    np.savez_compressed(os.path.join(HERE, "mpnet_small.npz"), ids=ids.numpy(), mask=mask.numpy(), last_hidden=last.numpy(),
                        mean=mean.numpy(), mean_norm=torch.nn.functional.normalize(mean, p=2, dim=1).numpy(),
                        cfg=np.array([300, 256, 128, 2, 2, 256], dtype=np.int64),   # vocab, usable positions, W, layers, heads, F
                        **{"w:" + k: v for k, v in _np(sd).items()})


def make_xlmr():
    """transformers.XLMRobertaModel == what HuggingFaceModel loads through AutoModel for the multilingual-e5 family:
    BERT encoder, position ids offset by padding_idx + 1 = 2, one token type."""
    from transformers import XLMRobertaConfig, XLMRobertaModel
    torch.manual_seed(0)
    cfg = XLMRobertaConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                           max_position_embeddings=66, type_vocab_size=1, layer_norm_eps=1e-5, hidden_act="gelu",
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    m = XLMRobertaModel(cfg, add_pooling_layer=False).eval()
    _jitter(m, 5)
    g = torch.Generator().manual_seed(6)
    lens = [5, 17, 2, 33, 64, 9]
    S = max(lens)
    ids = torch.full((len(lens), S), 1, dtype=torch.int64)  # <pad> = 1
    mask = torch.zeros(len(lens), S, dtype=torch.int64)
    for i, L in enumerate(lens):
        ids[i, :L] = torch.randint(4, 300, (L,), generator=g)
        ids[i, 0], ids[i, L - 1] = 0, 2  # <s> ... </s>
        mask[i, :L] = 1
    with torch.no_grad():
        last = m(input_ids=ids, attention_mask=mask).last_hidden_state
    lh = last.masked_fill(~mask[..., None].bool(), 0.0)
    mean = lh.sum(1) / mask.sum(1)[..., None]
    mean_n = torch.nn.functional.normalize(mean, p=2, dim=1)
    sd = dict(m.state_dict())
    np.savez_compressed(os.path.join(HERE, "xlmr_small.npz"), ids=ids.numpy(), mask=mask.numpy(), last_hidden=last.numpy(),
                        mean=mean.numpy(), mean_norm=mean_n.numpy(),
                        cfg=np.array([300, 66, 128, 2, 2, 256], dtype=np.int64),  # vocab max_pos W layers heads F
                        **{"w:" + k: v for k, v in _np(sd).items()})


def _clip_blocks_to_open_clip(hf_sd, hf_prefix, oc_prefix, layers):
    out = {}
    for i in range(layers):
        s = f"{hf_prefix}encoder.layers.{i}."
        d = f"{oc_prefix}resblocks.{i}."
        out[d + "ln_1.weight"] = hf_sd[s + "layer_norm1.weight"]
        out[d + "ln_1.bias"] = hf_sd[s + "layer_norm1.bias"]
        out[d + "attn.in_proj_weight"] = torch.cat([hf_sd[s + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        out[d + "attn.in_proj_bias"] = torch.cat([hf_sd[s + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        out[d + "attn.out_proj.weight"] = hf_sd[s + "self_attn.out_proj.weight"]
        out[d + "attn.out_proj.bias"] = hf_sd[s + "self_attn.out_proj.bias"]
        out[d + "ln_2.weight"] = hf_sd[s + "layer_norm2.weight"]
        out[d + "ln_2.bias"] = hf_sd[s + "layer_norm2.bias"]
        out[d + "mlp.c_fc.weight"] = hf_sd[s + "mlp.fc1.weight"]
        out[d + "mlp.c_fc.bias"] = hf_sd[s + "mlp.fc1.bias"]
        out[d + "mlp.c_proj.weight"] = hf_sd[s + "mlp.fc2.weight"]
        out[d + "mlp.c_proj.bias"] = hf_sd[s + "mlp.fc2.bias"]
    return out


def make_clip_vit():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    outs = {}
    sd_oc = None
    px = None
    for act in ("gelu", "quick_gelu"):
        torch.manual_seed(3)
        cfg = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                               image_size=64, patch_size=16, projection_dim=64, hidden_act=act, layer_norm_eps=1e-5,
                               attention_dropout=0.0)
        m = CLIPVisionModelWithProjection(cfg).eval()
        _jitter(m, 4)
        hf = m.state_dict()
        g = torch.Generator().manual_seed(5)
        px = torch.randn(5, 3, 64, 64, generator=g)
        with torch.no_grad():
            emb = m(pixel_values=px).image_embeds
        outs[act] = emb.numpy()
        sd_oc = {
            "visual.conv1.weight": hf["vision_model.embeddings.patch_embedding.weight"],
            "visual.class_embedding": hf["vision_model.embeddings.class_embedding"],
            "visual.positional_embedding": hf["vision_model.embeddings.position_embedding.weight"],
            "visual.ln_pre.weight": hf["vision_model.pre_layrnorm.weight"],
            "visual.ln_pre.bias": hf["vision_model.pre_layrnorm.bias"],
            "visual.ln_post.weight": hf["vision_model.post_layernorm.weight"],
            "visual.ln_post.bias": hf["vision_model.post_layernorm.bias"],
            "visual.proj": hf["visual_projection.weight"].t().contiguous(),
        }
        sd_oc.update(_clip_blocks_to_open_clip(hf, "vision_model.", "visual.transformer.", 2))
    np.savez_compressed(os.path.join(HERE, "clip_vit_small.npz"), pixels=px.numpy(),
                        emb_gelu=outs["gelu"], emb_quick_gelu=outs["quick_gelu"],
                        cfg=np.array([64, 16, 128, 2, 2, 256, 64], dtype=np.int64),  # S P W layers heads F D
                        **{"w:" + k: v for k, v in _np(sd_oc).items()})


def make_clip_text():
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    torch.manual_seed(6)
    V, ctx = 300, 16
    cfg = CLIPTextConfig(vocab_size=V, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, max_position_embeddings=ctx, projection_dim=64, hidden_act="gelu",
                         layer_norm_eps=1e-5, attention_dropout=0.0, eos_token_id=2, bos_token_id=1, pad_token_id=0)
    m = CLIPTextModelWithProjection(cfg).eval()
    _jitter(m, 7)
    hf = m.state_dict()
    g = torch.Generator().manual_seed(8)
    lens = [3, 16, 9, 5, 12]
    ids = torch.zeros(len(lens), ctx, dtype=torch.int64)
    for i, L in enumerate(lens):
        ids[i, 0] = V - 2
        ids[i, 1:L - 1] = torch.randint(1, V - 2, (L - 2,), generator=g)
        ids[i, L - 1] = V - 1  # EOT = max id -> argmax pooling
    with torch.no_grad():
        emb = m(input_ids=ids).text_embeds
    sd_oc = {
        "token_embedding.weight": hf["text_model.embeddings.token_embedding.weight"],
        "positional_embedding": hf["text_model.embeddings.position_embedding.weight"],
        "ln_final.weight": hf["text_model.final_layer_norm.weight"],
        "ln_final.bias": hf["text_model.final_layer_norm.bias"],
        "text_projection": hf["text_projection.weight"].t().contiguous(),
    }
    sd_oc.update(_clip_blocks_to_open_clip(hf, "text_model.", "transformer.", 2))
    np.savez_compressed(os.path.join(HERE, "clip_text_small.npz"), ids=ids.numpy(), emb=emb.numpy(),
                        cfg=np.array([V, ctx, 128, 2, 2, 256, 64], dtype=np.int64),  # V ctx W layers heads F D
                        **{"w:" + k: v for k, v in _np(sd_oc).items()})


def make_siglip():
    from transformers import SiglipTextConfig, SiglipTextModel, SiglipVisionConfig, SiglipVisionModel
    torch.manual_seed(9)
    W, Fd, Lyr, H, S, P = 128, 256, 2, 2, 64, 16
    vc = SiglipVisionConfig(hidden_size=W, intermediate_size=Fd, num_hidden_layers=Lyr, num_attention_heads=H, image_size=S,
                            patch_size=P, hidden_act="gelu", layer_norm_eps=1e-6, attention_dropout=0.0)
    vm = SiglipVisionModel(vc).eval()
    _jitter(vm, 10)
    with torch.no_grad():
        dict(vm.named_parameters())[[k for k, _ in vm.named_parameters() if k.endswith("head.probe")][0]].mul_(1.0 / 3.0).add_(0.3 * torch.randn(1, 1, W, generator=torch.Generator().manual_seed(11)))
    hf = {k.replace("vision_model.", "", 1) if k.startswith("vision_model.") else k: v for k, v in vm.state_dict().items()}
    g = torch.Generator().manual_seed(12)
    px = torch.randn(5, 3, S, S, generator=g)
    with torch.no_grad():
        img = vm(pixel_values=px).pooler_output
    t = "visual.trunk."
    sd = {t + "patch_embed.proj.weight": hf["embeddings.patch_embedding.weight"], t + "patch_embed.proj.bias": hf["embeddings.patch_embedding.bias"],
          t + "pos_embed": hf["embeddings.position_embedding.weight"].unsqueeze(0),
          t + "norm.weight": hf["post_layernorm.weight"], t + "norm.bias": hf["post_layernorm.bias"]}
    for i in range(Lyr):
        s_, d = f"encoder.layers.{i}.", f"{t}blocks.{i}."
        sd[d + "norm1.weight"], sd[d + "norm1.bias"] = hf[s_ + "layer_norm1.weight"], hf[s_ + "layer_norm1.bias"]
        sd[d + "attn.qkv.weight"] = torch.cat([hf[s_ + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        sd[d + "attn.qkv.bias"] = torch.cat([hf[s_ + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        sd[d + "attn.proj.weight"], sd[d + "attn.proj.bias"] = hf[s_ + "self_attn.out_proj.weight"], hf[s_ + "self_attn.out_proj.bias"]
        sd[d + "norm2.weight"], sd[d + "norm2.bias"] = hf[s_ + "layer_norm2.weight"], hf[s_ + "layer_norm2.bias"]
        for n in ("fc1", "fc2"):
            sd[d + f"mlp.{n}.weight"], sd[d + f"mlp.{n}.bias"] = hf[s_ + f"mlp.{n}.weight"], hf[s_ + f"mlp.{n}.bias"]
    a = t + "attn_pool."
    ipw, ipb = hf["head.attention.in_proj_weight"], hf["head.attention.in_proj_bias"]
    sd.update({a + "latent": hf["head.probe"], a + "q.weight": ipw[:W], a + "q.bias": ipb[:W], a + "kv.weight": ipw[W:], a + "kv.bias": ipb[W:],
               a + "proj.weight": hf["head.attention.out_proj.weight"], a + "proj.bias": hf["head.attention.out_proj.bias"],
               a + "norm.weight": hf["head.layernorm.weight"], a + "norm.bias": hf["head.layernorm.bias"]})
    for n in ("fc1", "fc2"):
        sd[a + f"mlp.{n}.weight"], sd[a + f"mlp.{n}.bias"] = hf[f"head.mlp.{n}.weight"], hf[f"head.mlp.{n}.bias"]

    V, ctx, D = 300, 16, 64
    tc = SiglipTextConfig(vocab_size=V, hidden_size=W, intermediate_size=Fd, num_hidden_layers=Lyr, num_attention_heads=H,
                          max_position_embeddings=ctx, hidden_act="gelu", layer_norm_eps=1e-6, attention_dropout=0.0, projection_size=D)
    tm = SiglipTextModel(tc).eval()
    _jitter(tm, 13)
    ht = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in tm.state_dict().items()}
    lens = [3, 16, 9, 5, 12]
    ids = torch.ones(len(lens), ctx, dtype=torch.int64)  # pad id 1 (= </s>), which is also the EOS that ends each text
    for i, n in enumerate(lens):
        ids[i, :n - 1] = torch.randint(2, V, (n - 1,), generator=g)
    with torch.no_grad():
        txt = tm(input_ids=ids).pooler_output
    tp = "text_model."
    sd.update({"text.token_embedding.weight": ht[tp + "embeddings.token_embedding.weight"],
               "text.positional_embedding": ht[tp + "embeddings.position_embedding.weight"],
               "text.ln_final.weight": ht[tp + "final_layer_norm.weight"], "text.ln_final.bias": ht[tp + "final_layer_norm.bias"],
               "text.text_projection.weight": ht[tp + "head.weight"], "text.text_projection.bias": ht[tp + "head.bias"]})
    sd.update(_clip_blocks_to_open_clip(ht, tp, "text.transformer.", Lyr))
    np.savez_compressed(os.path.join(HERE, "siglip_small.npz"), pixels=px.numpy(), image_emb=img.numpy(), ids=ids.numpy(), text_emb=txt.numpy(),
                        cfg=np.array([S, P, W, Lyr, H, Fd, V, ctx, D], dtype=np.int64),  # S P W layers heads F | vocab ctx D (text: same W/L/H/F)
                        **{"w:" + k: v for k, v in _np(sd).items()})


if __name__ == "__main__":
    make_bert()
    make_bert(heads=4, name="bert_small_h32.npz")  # 32-wide heads (e5-small / bge-small / MiniLM class)
    make_xlmr()
    make_mpnet()
    make_clip_vit()
    make_clip_text()
    make_siglip()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
