"""CLIP-family loaders of the engine: `OPEN_CLIP` (type 'open_clip'), `CLIP` (type 'clip'), `FP16_CLIP`.

Drop-in for the reference's loader classes
  OPEN_CLIP  src/marqo/core/inference/embedding_models/open_clip_model.py:28-285
  CLIP       src/marqo/s2_inference/clip_utils.py:295-492        FP16_CLIP  :495-518
with the same constructor / load() / encode() / encode_text() / encode_image() / .preprocess surface
(SURVEY.md §8b), but `self.model` is a pair of HIP towers (marqo_amd.engine.towers) instead of a torch module:
every FLOP runs in libmarqo_hip.so on the MI355X, and images are resized on the GPU too.

Precision rule (reference open_clip_model.py:255-260: fp16 autocast on cuda, fp32 on cpu): here the device is
always an AMD GPU and the towers always run bf16 MFMA with fp32 accumulation / residual stream; outputs are fp32.
"""
from __future__ import annotations

import json
import os
import threading
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch
from PIL.Image import Image as ImageType

from marqo_amd import _lib as L
from marqo_amd.engine import archs, checkpoint, synthetic
from marqo_amd.engine.gpu_tokenizers import prefers_host
from marqo_amd.engine.towers import request_stream
from marqo_amd.engine.tokenizers import (ClipBpeTokenizer, RobertaBpeTokenizer, SiglipTokenizer, SyntheticTokenizer, WordPieceTokenizer,
                                          XlmRobertaTokenizer, _clean_text)
from marqo_amd.s2_inference.abstract_models import AbstractCLIPModel
from marqo_amd.s2_inference.errors import InvalidModelPropertiesError, ModelLoadError
from marqo_amd.s2_inference.image_input import format_and_load_CLIP_image, format_and_load_CLIP_images, pil_to_pixels, pil_to_rgb_u8

HF_HUB_PREFIX = "hf-hub:"
MARQO_OPEN_CLIP_REGISTRY_PREFIX = "open_clip/"
BPE_VOCAB_FILE = "bpe_simple_vocab_16e6.txt.gz"

_PREPROCESSOR_NORMS = {
    # image_preprocessor -> (mean, std, resize_mode, interpolation): open_clip's _pcfg() / _slpcfg() / _apcfg() base configs the
    # reference starts from (open_clip_model.py:87-97).  OpenCLIP / OpenAI share the OpenAI dataset statistics (clip_utils.py:32-33) and
    # resize the shorter side (bicubic) then centre-crop; SigLIP normalises with 0.5 / 0.5 and squashes the image to S x S (bicubic);
    # CLIPA uses the ImageNet statistics and a BILINEAR squash.
    "OpenCLIP": (archs.OPENAI_DATASET_MEAN, archs.OPENAI_DATASET_STD, "shortest", "bicubic"),
    "OpenAI": (archs.OPENAI_DATASET_MEAN, archs.OPENAI_DATASET_STD, "shortest", "bicubic"),
    "SigLIP": ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5), "squash", "bicubic"),
    "CLIPA": ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225), "squash", "bilinear"),
}
_TIMM_SIGLIP = __import__("re").compile(r"^vit_(base|large|so400m)_patch1[46]_siglip_(\d+)$")


class OpenCLIPModelProperties:
    """Validated view of the user's model_properties (reference: open_clip_model_properties.py:24-73)."""
    _KNOWN = {"name", "dimensions", "type", "jit", "precision", "url", "localpath", "model_location", "modelLocation",
              "tokenizer", "image_preprocessor", "imagePreprocessor", "mean", "std", "size", "note", "notes", "pretrained",
              "model_size", "text_query_prefix", "text_chunk_prefix", "tokens", "enginePrecision", "engine_precision", "fp8Budget", "fp8_budget"}

    def __init__(self, **p):
        if not isinstance(p.get("name"), str) or not p["name"]:
            raise ValueError("'name' is required and must be a string")
        dims = p.get("dimensions")
        if not isinstance(dims, int) or dims < 1:
            raise ValueError("'dimensions' must be a positive integer")
        if "type" not in p:
            raise ValueError("'type' is required")
        self.name: str = p["name"]
        self.dimensions: int = dims
        self.type: str = p["type"]
        self.jit: bool = bool(p.get("jit", False))
        self.precision: str = p.get("precision", "fp32")
        if self.precision not in ("fp32", "fp16"):
            raise ValueError("'precision' must be 'fp32' or 'fp16'")
        self.url: Optional[str] = p.get("url")
        self.localpath: Optional[str] = p.get("localpath")
        self.model_location = p.get("model_location", p.get("modelLocation"))
        if sum(1 for f in (self.url, self.localpath, self.model_location) if f is not None) > 1:
            raise ValueError("Only one of 'url', 'localpath', or 'model_location' should be provided.")
        self.tokenizer: Optional[str] = p.get("tokenizer")
        self.image_preprocessor: str = p.get("image_preprocessor", p.get("imagePreprocessor", "OpenCLIP"))
        if self.image_preprocessor not in ("SigLIP", "OpenAI", "OpenCLIP", "CLIPA"):
            raise ValueError(f"invalid image_preprocessor {self.image_preprocessor!r}")
        self.mean: Optional[List[float]] = p.get("mean")
        self.std: Optional[List[float]] = p.get("std")
        self.size: Optional[int] = p.get("size")
        self.pretrained: Optional[str] = p.get("pretrained")
        # engine extension (BASELINE config 5): operand type of the encoder-block GEMMs.  "bf16" (default) or "fp8" (OCP e4m3,
        # MX-MFMA at twice the bf16 rate).  fp8 is calibrated deterministically at load on fixed seeded inputs: static activation
        # scales and the number of trailing blocks that run on fp8 while max (1 - cos) vs the bf16 tower stays <= 'fp8Budget'
        # (default MARQO_AMD_FP8_BUDGET = 5e-4; a larger budget moves more blocks to fp8).  The reference's own 'precision' key
        # (fp32 / fp16 autocast) keeps its meaning and is accepted for both.
        self.engine_precision: str = p.get("enginePrecision", p.get("engine_precision", os.environ.get("MARQO_AMD_PRECISION", "bf16")))
        if self.engine_precision not in ("bf16", "fp8"):
            raise ValueError("'enginePrecision' must be 'bf16' or 'fp8'")
        b = p.get("fp8Budget", p.get("fp8_budget"))
        if b is not None and not (isinstance(b, (int, float)) and b > 0):
            raise ValueError("'fp8Budget' must be a positive number")
        self.fp8_budget: Optional[float] = None if b is None else float(b)

    def dict(self) -> dict:
        return dict(self.__dict__)


STAGE_BYTES = int(os.environ.get("MARQO_AMD_IMAGE_STAGE_BYTES", str(1 << 30)))   # decoded pixel bytes staged per resize call (pinned host + HBM)
# A list call of >= PIPELINE_MIN images goes through in STAGES: everything the GPU does is enqueued asynchronously, so the host packs stage k + 1
# (pinned staging, a few copy threads) while the GPU resizes and encodes stage k.  Stage size is a trade: the pack in front of the first stage
# is the GPU's idle time (1.3 ms for 256 Pillow images, 5 ms for 1 024), a small stage's GEMMs fill fewer of the chip's tile slots (64-image stages
# ran 256 images in 5.5 ms instead of 3.2, profiles/r02g_e2e_profile.txt).  Measured in round 5 (profiles/r05w_e2e_stages.txt, one synchronous
# caller, ViT-B/32, Pillow 224 x 224): 256 images 5.21 ms in one batch / 4.97 in two stages / 4.90 with the stages on two streams; 384: 7.96 / 6.88;
# 512: 8.70 / 7.87 / 7.74; 1 024: 18.3 in one batch, 15.1 in the 512-image stages of rounds 2-4, 13.7 in 256-image ones.  So: stages of about
# PIPELINE_CHUNK images, equal in size, at least two of them.
PIPELINE_CHUNK = max(1, int(os.environ.get("MARQO_AMD_IMAGE_PIPELINE_CHUNK", "256")))
PIPELINE_MIN = max(2, int(os.environ.get("MARQO_AMD_IMAGE_PIPELINE_MIN", "256")))
# The stages alternate between this many HIP streams (1 = all on the request stream): stage k + 1's launches interleave with stage k's and fill
# their tails instead of queueing behind the whole tower (+1.5 ... 8 %, same bits; profiles/r05ah_e2e_two_stream_timeline.txt).
PIPELINE_STREAMS = max(1, int(os.environ.get("MARQO_AMD_IMAGE_PIPELINE_STREAMS", "2")))
# A stage's tower is ~90 launches = 0.35 ms of host time through the boundary (GIL released inside the op): enqueued by a helper thread, the calling
# thread packs the next stage meanwhile instead of afterwards.  Used where the GPU waits for the host — stages smaller than PIPELINE_CHUNK, i.e. calls
# of fewer than 2 * PIPELINE_CHUNK images (256 images: 4.58 -> 4.36 ms; with 256-image stages the host is ahead anyway and the hand-over costs 2 %:
# profiles/r05w_e2e_stages.txt).  MARQO_AMD_IMAGE_PIPELINE_THREAD=0: everything on the calling thread.
PIPELINE_THREAD = os.environ.get("MARQO_AMD_IMAGE_PIPELINE_THREAD", "1") != "0"
PIPELINE_ALWAYS = os.environ.get("MARQO_AMD_IMAGE_PIPELINE_ALWAYS", "0") == "1"   # stage a call also while other image calls are in flight on the model
# `.preprocess` results live in per-model device slabs of this many image slots (0 = one tensor per image, as before round 6)
PREPROCESS_SLAB_SLOTS = int(os.environ.get("MARQO_AMD_PREPROCESS_SLAB", "128"))


def _pipeline_stages(n: int) -> list:
    """[(first, last + 1)] of the stages of a pipelined image call: max(2, round(n / PIPELINE_CHUNK)) stages of equal size"""
    k = max(2, int(n / PIPELINE_CHUNK + 0.5))
    size = -(-n // k)
    return [(a, min(a + size, n)) for a in range(0, n, size)]


class HfClipTokenizer:
    """open_clip's HFTokenizer (tokenizer.py: `clean_fn` = ftfy / html-unescape / whitespace clean, then the Hugging Face tokenizer with
    max_length = context_length, padding = 'max_length', truncation = True -> input_ids): `hf` is the XLM-RoBERTa SentencePiece tokenizer
    of record (engine/tokenizers.py), rows <s> pieces </s> padded to ctx with <pad>."""

    def __init__(self, hf, context_length: int = 77):
        self.hf, self.context_length = hf, context_length
        self.pad_id = getattr(hf, "pad_id", 1)

    def ids(self, cleaned_texts) -> np.ndarray:
        out = np.full((len(cleaned_texts), self.context_length), self.pad_id, dtype=np.int64)
        enc = self.hf(list(cleaned_texts), max_length=self.context_length)["input_ids"]
        out[:, :enc.shape[1]] = enc
        return out

    def __call__(self, texts) -> np.ndarray:
        if isinstance(texts, str):
            texts = [texts]
        return self.ids([_clean_text(t) for t in texts])


class OPEN_CLIP(AbstractCLIPModel):
    supports_dynamic_batching = True
    _own_text_tower = False   # subclasses that pair the CLIP image tower with a text encoder of their own (MULTILINGUAL_CLIP)

    def __init__(self, device: Optional[str] = None, model_properties: Optional[Dict] = None, model_auth=None) -> None:
        super().__init__(device, model_properties, model_auth)
        self.model_properties = self._build_model_properties(model_properties or {})
        self.preprocess_config = None
        self._local = threading.local()
        self._image_calls = 0               # encode_image calls in flight on this model (a call stages itself only when it is alone)
        self._image_calls_lock = threading.Lock()
        self._slab_lock = threading.Lock()
        self._slab_block: Optional[torch.Tensor] = None      # the preprocess slab's current block (fp32 [slots, 3, S, S]) and its next free slot
        self._slab_next = 0
        self.vision_arch: Optional[archs.VitArch] = None
        self.text_arch: Optional[archs.ClipTextArch] = None
        self.weights_source = None

    def _build_model_properties(self, model_properties: dict) -> OpenCLIPModelProperties:
        try:
            return OpenCLIPModelProperties(**model_properties)
        except (ValueError, TypeError) as e:
            raise InvalidModelPropertiesError(f"Invalid model properties: {model_properties}. Original error: {e}") from e

    # ---- loading -------------------------------------------------------------------------------------------
    def _architecture_and_tag(self):
        name, props = self.model_properties.name, self.model_properties
        if props.url is not None or props.localpath is not None or props.model_location is not None:
            return name, props.pretrained
        if name.startswith(HF_HUB_PREFIX):
            return name, None
        if name.startswith(MARQO_OPEN_CLIP_REGISTRY_PREFIX):
            parts = name.split("/", 3)
            if len(parts) < 3:
                raise InvalidModelPropertiesError(f"open_clip registry names look like open_clip/<arch>/<pretrained>, got {name}")
            return parts[1], parts[2]
        raise InvalidModelPropertiesError("Marqo cannot load the provided open_clip model: expected a custom checkpoint "
                                          "('url' / 'localpath' / 'model_location'), an 'hf-hub:' name or an 'open_clip/' registry name")

    def _resolve_archs(self, arch_name: str, tag: Optional[str], ckpt_dir: Optional[str]):
        if arch_name.startswith(HF_HUB_PREFIX):
            cfg_path = os.path.join(ckpt_dir or "", "open_clip_config.json")
            if not os.path.isfile(cfg_path) and arch_name in archs.KNOWN_HF_HUB_ARCHS:
                return archs.resolve_open_clip(archs.KNOWN_HF_HUB_ARCHS[arch_name])
            if not os.path.isfile(cfg_path):
                raise ModelLoadError(f"{arch_name}: open_clip_config.json not found next to the checkpoint")
            with open(cfg_path) as f:
                mc = json.load(f)["model_cfg"]
            v, t = mc["vision_cfg"], mc["text_cfg"]
            m = _TIMM_SIGLIP.match(str(v.get("timm_model_name", "")))
            if m and "hf_model_name" not in t:
                # SigLIP: timm trunk + TextTransformer(no_causal_mask, pool 'last', proj_bias) — marqo-fashionSigLIP, marqo-ecommerce-*
                if v.get("timm_pool", "map") != "map" or v.get("timm_proj", "none") not in ("none", None, ""):
                    raise InvalidModelPropertiesError(f"{arch_name}: timm towers are supported with pool 'map' and no projection only")
                vision, text = archs._siglip(int(v.get("image_size", m.group(2))), large=m.group(1) == "large", so400m=m.group(1) == "so400m")
                if mc["embed_dim"] != vision.width or t.get("width", vision.width) != text.width or t.get("layers", text.layers) != text.layers:
                    raise InvalidModelPropertiesError(f"{arch_name}: unexpected SigLIP dimensions in open_clip_config.json")
                from dataclasses import replace
                return vision, replace(text, vocab=t.get("vocab_size", text.vocab), ctx=t.get("context_length", text.ctx),
                                       causal=not t.get("no_causal_mask", True), proj_bias=bool(t.get("proj_bias", True)))
            if not isinstance(v.get("layers"), int) or "hf_model_name" in t:
                raise InvalidModelPropertiesError(f"{arch_name}: only plain CLIP ViT + CLIP text towers are supported. {archs.UNSUPPORTED_HINT}")
            head = v.get("head_width", 64)
            vision = archs.VitArch(v.get("image_size", 224), v["patch_size"], v["width"], v["layers"], v["width"] // head,
                                   int(v["width"] * v.get("mlp_ratio", 4.0)), mc["embed_dim"], bool(mc.get("quick_gelu", False)))
            text = archs.ClipTextArch(t.get("vocab_size", 49408), t.get("context_length", 77), t.get("width", 512), t.get("layers", 12),
                                      t.get("heads", 8), int(t.get("width", 512) * t.get("mlp_ratio", 4.0)), mc["embed_dim"],
                                      bool(mc.get("quick_gelu", False)))
            return vision, text
        try:
            return archs.resolve_open_clip(arch_name, tag)
        except KeyError as e:
            raise InvalidModelPropertiesError(str(e)) from e

    def _load_necessary_components(self) -> None:
        if not str(self.device).startswith("cuda"):
            raise L.MarqoHipUnavailableError(
                f"marqo_amd runs its towers on an AMD GPU only (device 'cuda' / 'cuda:N' on ROCm); got {self.device!r}")
        from marqo_amd.engine import towers
        from marqo_amd.engine.preprocess import ImagePreprocessor
        arch_name, tag = self._architecture_and_tag()
        props = self.model_properties
        if props.url is not None or props.model_location is not None:
            raise ModelLoadError("downloading checkpoints ('url' / 'model_location') is control-plane work outside the marqo_amd "
                                 "engine; place the file on disk and pass 'localpath'")
        if props.localpath is not None and not os.path.exists(props.localpath):
            raise InvalidModelPropertiesError(f"The localpath '{props.localpath}' does not exist. Please provide a valid localpath "
                                              f"to load the model.")
        ckpt = checkpoint.find_open_clip_checkpoint(props.name, props.localpath)
        ckpt_dir = os.path.dirname(ckpt) if ckpt and os.path.isfile(ckpt) else ckpt
        self.vision_arch, self.text_arch = self._resolve_archs(arch_name, tag, ckpt_dir)
        if props.size is not None and props.size != self.vision_arch.image_size:
            raise InvalidModelPropertiesError(f"'size'={props.size} does not match the architecture's image size {self.vision_arch.image_size}")
        if self.vision_arch.out_dim != props.dimensions:
            raise InvalidModelPropertiesError(f"'dimensions'={props.dimensions} but {arch_name} produces {self.vision_arch.out_dim}-d embeddings")
        if ckpt is not None:
            sd = checkpoint.load_state_dict(ckpt)
            self.weights_source = ckpt
        elif checkpoint.synthetic_weights_enabled():
            sd = synthetic.random_open_clip_state_dict(vision=self.vision_arch, text=None if self._own_text_tower else self.text_arch, seed=0)
            self.weights_source = "synthetic(seed=0)"
        else:
            raise ModelLoadError(f"no checkpoint for {props.name} under {checkpoint.model_dir()} (and no 'localpath'). There is no "
                                 f"network download in the marqo_amd engine; set MARQO_AMD_SYNTHETIC_WEIGHTS=1 for random-init weights.")
        # preprocessing: a custom checkpoint follows 'image_preprocessor' (open_clip_model.py:87-104); registry and hf-hub names
        # get open_clip's own transform for that model (create_model_and_transforms, :183-205) — the SigLIP pipeline for SigLIP towers
        custom = props.localpath is not None
        kind = props.image_preprocessor if custom else (self.vision_arch.preprocessor or ("SigLIP" if self.vision_arch.pool == "map" else "OpenCLIP"))
        mean, std, self._resize_mode, self._interpolation = _PREPROCESSOR_NORMS[kind]
        self._mean = tuple(props.mean) if props.mean is not None else mean
        self._std = tuple(props.std) if props.std is not None else std
        self.preprocess_config = {"size": self.vision_arch.image_size, "mean": self._mean, "std": self._std,
                                  "interpolation": self._interpolation, "resize_mode": self._resize_mode}
        try:
            self.vision = towers.VitTower(self.vision_arch, sd, self.device, mean=self._mean, std=self._std, precision=props.engine_precision)
            self.text = self._make_text_tower(sd, props.engine_precision)
        except ValueError as e:  # e.g. fp8 needs width / mlp_dim multiples of 128
            raise InvalidModelPropertiesError(str(e)) from e
        if props.engine_precision == "fp8":
            # deterministic load-time calibration on fixed seeded inputs (NOT on whatever request arrives first): static activation
            # scales with head-room + the bf16 / fp8 block split that keeps the measured error inside props.fp8_budget
            self.vision.tune_fp8_default(props.fp8_budget)
            self.text.tune_fp8_default(props.fp8_budget)
        # the load-time policies are fixed now: folded-LayerNorm weight copies no block can use any more go back to the allocator
        self.vision.release_unused_folded()
        self.text.release_unused_folded()
        self.model = (self.vision, self.text)
        self.tokenizer = self._load_tokenizer(ckpt_dir)
        # K14: byte-level BPE on the device for ASCII texts (identical ids; the host tokeniser handles the rest)
        self._device_tokenizer = None
        if isinstance(self.tokenizer, ClipBpeTokenizer) and os.environ.get("MARQO_AMD_HOST_TOKENIZER", "0") != "1" \
                and not getattr(self.text_arch, "cls_embed", False):
            from marqo_amd.engine.gpu_tokenizers import DeviceClipBpeTokenizer
            self._device_tokenizer = DeviceClipBpeTokenizer(self.tokenizer, self.device)
        elif isinstance(self.tokenizer, XlmRobertaTokenizer) and os.environ.get("MARQO_AMD_HOST_TOKENIZER", "0") != "1":
            from marqo_amd.engine.gpu_tokenizers import DeviceSentencePieceTokenizer
            try:
                self._device_tokenizer = DeviceSentencePieceTokenizer(self.tokenizer, self.device)
            except ValueError:
                self._device_tokenizer = None
        elif isinstance(self.tokenizer, WordPieceTokenizer) and os.environ.get("MARQO_AMD_HOST_TOKENIZER", "0") != "1":
            from marqo_amd.engine.gpu_tokenizers import DeviceWordPieceTokenizer
            self._device_tokenizer = DeviceWordPieceTokenizer(self.tokenizer, self.device)
        elif isinstance(self.tokenizer, HfClipTokenizer) and isinstance(self.tokenizer.hf, XlmRobertaTokenizer) and \
                os.environ.get("MARQO_AMD_HOST_TOKENIZER", "0") != "1":
            from marqo_amd.engine.gpu_tokenizers import DeviceSentencePieceTokenizer
            try:
                self._device_tokenizer = DeviceSentencePieceTokenizer(self.tokenizer.hf, self.device)
            except ValueError:
                self._device_tokenizer = None
        elif isinstance(self.tokenizer, SiglipTokenizer) and self.tokenizer._sp is not None and os.environ.get("MARQO_AMD_HOST_TOKENIZER", "0") != "1":
            from marqo_amd.engine.gpu_tokenizers import DeviceSentencePieceTokenizer
            try:
                self._device_tokenizer = DeviceSentencePieceTokenizer(self.tokenizer, self.device)
            except ValueError:
                self._device_tokenizer = None
        self._ImagePreprocessor = ImagePreprocessor
        self.preprocess = self._preprocess_one

    def _make_text_tower(self, sd, precision: str):
        from marqo_amd.engine import towers
        if isinstance(self.text_arch, archs.HfClipTextArch):   # open_clip HFTextEncoder: XLM-RoBERTa encoder + mean pooler + projection MLP
            return towers.HfClipTextTower(self.text_arch, sd, self.device, precision=precision)
        return towers.ClipTextTower(self.text_arch, sd, self.device, precision=precision)

    def _load_tokenizer(self, ckpt_dir: Optional[str]):
        props = self.model_properties
        hf_name = props.tokenizer or getattr(self.text_arch, "hf_tokenizer", None)
        if hf_name and not isinstance(self.text_arch, archs.HfClipTextArch):
            d = checkpoint.find_hf_dir(hf_name) or (hf_name if os.path.isdir(hf_name) else None) or \
                (ckpt_dir if ckpt_dir and os.path.isfile(os.path.join(ckpt_dir, "vocab.txt")) else None)
            if d is None:
                if self.weights_source and str(self.weights_source).startswith("synthetic"):
                    return SyntheticTokenizer("clip", self.text_arch.vocab, self.text_arch.ctx)
                raise ModelLoadError(f"tokenizer {hf_name!r} (vocab.txt / tokenizer.json) not found on disk")
            wp = WordPieceTokenizer(d)
            ctx = self.text_arch.ctx
            strip_sep = bool(getattr(self.text_arch, "strip_sep", False))
            # open_clip HFTokenizer: clean, padding='max_length', truncation=True -> ids only; strip_sep_token: [SEP] -> 0 (tokenizer.py)
            def hf_tok(texts):
                if isinstance(texts, str):
                    texts = [texts]
                out = np.zeros((len(texts), ctx), dtype=np.int64)
                for i, t in enumerate(texts):
                    ids = wp.encode(_clean_text(t), max_length=ctx)
                    out[i, :len(ids)] = ids
                if strip_sep:
                    out[out == wp.sep_id] = 0
                return out
            return hf_tok
        if isinstance(self.text_arch, archs.HfClipTextArch):
            # open_clip HFTokenizer(hf_tokenizer_name = xlm-roberta-base / -large): the SentencePiece model next to the checkpoint, or the HF
            # repo of that name on disk
            size = "large" if self.text_arch.bert.width == 1024 else "base"
            if self.text_arch.bert.vocab == 50265:   # roberta-base: GPT-2 byte-level BPE (vocab.json + merges.txt)
                for d in filter(None, (ckpt_dir, checkpoint.find_hf_dir("roberta-base"), checkpoint.find_hf_dir("FacebookAI/roberta-base"))):
                    if os.path.isfile(os.path.join(d, "vocab.json")) and os.path.isfile(os.path.join(d, "merges.txt")):
                        return HfClipTokenizer(RobertaBpeTokenizer(d), self.text_arch.ctx)
            for d in filter(None, (ckpt_dir, checkpoint.find_hf_dir(f"xlm-roberta-{size}"), checkpoint.find_hf_dir(f"FacebookAI/xlm-roberta-{size}"))):
                if os.path.isfile(os.path.join(d, "sentencepiece.bpe.model")):
                    return HfClipTokenizer(XlmRobertaTokenizer(d), self.text_arch.ctx)
            if self.weights_source and str(self.weights_source).startswith("synthetic"):
                return HfClipTokenizer(SyntheticTokenizer("xlmr", self.text_arch.vocab), self.text_arch.ctx)
            raise ModelLoadError(f"the text tower's Hugging Face tokenizer (sentencepiece.bpe.model, or vocab.json + merges.txt for roberta-base) "
                                 f"was not found next to the checkpoint or under {os.path.join(checkpoint.model_dir(), 'hf')}")
        if not self.text_arch.causal:  # SigLIP: T5-style SentencePiece vocabulary next to the checkpoint (tokenizer.json / spiece.model)
            for d in filter(None, (ckpt_dir, os.path.join(checkpoint.model_dir(), "siglip"))):
                if os.path.isfile(os.path.join(d, "tokenizer.json")) or os.path.isfile(os.path.join(d, "spiece.model")):
                    return SiglipTokenizer(d, context_length=self.text_arch.ctx)
            if self.weights_source and str(self.weights_source).startswith("synthetic"):
                return SyntheticTokenizer("siglip", self.text_arch.vocab, self.text_arch.ctx)
            raise ModelLoadError(f"SigLIP tokenizer (tokenizer.json / spiece.model) not found next to the checkpoint or under "
                                 f"{os.path.join(checkpoint.model_dir(), 'siglip')}")
        # (CoCa: the tokenizer fills ctx - 1 positions, the tower appends the class embedding as the ctx-th)
        text_ctx = self.text_arch.ctx - (1 if getattr(self.text_arch, "cls_embed", False) else 0)
        for d in filter(None, (ckpt_dir, checkpoint.model_dir(), os.path.join(checkpoint.model_dir(), "open_clip"))):
            p = os.path.join(d, BPE_VOCAB_FILE)
            if os.path.isfile(p):
                return ClipBpeTokenizer(p, context_length=text_ctx)
        if self.weights_source and str(self.weights_source).startswith("synthetic"):
            return SyntheticTokenizer("clip", self.text_arch.vocab, text_ctx)
        raise ModelLoadError(f"CLIP BPE vocabulary {BPE_VOCAB_FILE} not found next to the checkpoint or under {checkpoint.model_dir()}")

    def _check_loaded_components(self):
        if self.model is None:
            raise RuntimeError("The open_clip model is not loaded. Please load the model before inference.")
        if self.tokenizer is None:
            raise RuntimeError("The open_clip tokenizer is not loaded. Please load the tokenizer before inference.")
        if self.preprocess is None:
            raise RuntimeError("The open_clip image preprocessor is not loaded. Please load the image preprocessor before inference.")

    # ---- preprocessing ---------------------------------------------------------------------------------------
    def _pre(self):
        """one ImagePreprocessor (scratch workspace) per calling thread: `.preprocess` is invoked concurrently from the
        media download threads (add_docs.py:354-375)"""
        p = getattr(self._local, "pre", None)
        if p is None:
            p = self._local.pre = self._ImagePreprocessor(self.device, self.vision_arch.image_size, self._mean, self._std)
        return p

    def _preprocess_one(self, image: ImageType) -> torch.Tensor:
        """PIL image -> Tensor[3, S, S] fp32 (normalised), ALREADY on the device: resize / crop / normalise run on the GPU.
        Callers' `.to(device)` (add_docs.py:134) is then a no-op.
        The result is written into the next slot of a per-model device SLAB (blocks of PREPROCESS_SLAB_SLOTS images) and returned as a view of it:
        the reference's download threads call this image by image (add_docs.py:129-141) and hand encode_image a LIST of such tensors, which used
        to be gathered again into one batch (a 154 MB copy per 256 images + a Python loop).  encode_image recognises runs of neighbouring slots
        (`_mq_block` / `_mq_slot` on the view: attributes of that tensor object, gone with any copy of it) and hands the tower the slab slice
        itself.  A slot is written once; a block lives as long as any of its views."""
        pre = self._pre()
        u8 = self._resize(pre, [pil_to_pixels(image)])
        if PREPROCESS_SLAB_SLOTS <= 0:
            return pre.to_tensor_normalize(u8)[0]
        block, k = self._slab_take()
        pre.to_tensor_normalize(u8, out=block[k:k + 1])
        view = block[k]
        view._mq_block, view._mq_slot = block, k
        return view

    def _slab_take(self):
        S = self.vision_arch.image_size
        with self._slab_lock:
            if self._slab_block is None or self._slab_next >= self._slab_block.shape[0]:
                with torch.cuda.device(self.device):
                    # (allocated on the caller's current stream; the views are consumed on request streams: torch's allocator would only re-use the
                    # block after every view is gone, and a slot is never rewritten, so no stream hand-over is needed beyond the tensors' own)
                    self._slab_block = torch.empty(PREPROCESS_SLAB_SLOTS, 3, S, S, dtype=torch.float32, device=self.device)
                self._slab_next = 0
            k = self._slab_next
            self._slab_next += 1
            return self._slab_block, k

    @staticmethod
    def _slab_runs(tensors):
        """a list of preprocess-slab views -> [slab slices] in list order (neighbouring slots of one block merge into one slice), or None when any
        item is not a slab view (its tensor object carries no slot: a copy, a caller's own tensor)"""
        runs, cur_block, k0, k1 = [], None, 0, 0
        for t in tensors:
            block = getattr(t, "_mq_block", None)
            if block is None:
                return None
            k = t._mq_slot
            if block is cur_block and k == k1:
                k1 += 1
                continue
            if cur_block is not None:
                runs.append(cur_block[k0:k1])
            cur_block, k0, k1 = block, k, k + 1
        if cur_block is not None:
            runs.append(cur_block[k0:k1])
        return runs

    def _resize(self, pre, raw, pil_sizes=None) -> torch.Tensor:
        """list of uint8 [H, W, 3] -> uint8 [n, S, S, 3] on the device: Resize(S) + CenterCrop(S), or Resize((S, S)) ('squash').
        The decoded pixels are staged (pinned host buffer + HBM copy) in groups of at most STAGE_BYTES, so a request of thousands of
        multi-megapixel images cannot pin tens of GB at once; the resized outputs (S x S x 3 bytes each) are what accumulates."""
        S = self.vision_arch.image_size
        if pil_sizes is not None:
            # a batch of loaded Pillow RGB images: sizes came from ONE native scan, the staging groups from a cumulative sum
            h, w = pil_sizes
            nbytes = h.astype(np.int64) * w.astype(np.int64) * 4
            bounds, start, acc = [0], 0, 0
            if int(nbytes.sum()) > STAGE_BYTES:
                for k, b in enumerate(nbytes.tolist()):
                    if k > start and acc + b > STAGE_BYTES:
                        bounds.append(k)
                        start, acc = k, 0
                    acc += b
            bounds.append(len(raw))
            outs = []
            for a0, a1 in zip(bounds[:-1], bounds[1:]):
                g, sz = (raw, pil_sizes) if (a0, a1) == (0, len(raw)) else (raw[a0:a1], (h[a0:a1], w[a0:a1]))
                outs.append(pre.resize_u8(g, S, S, self._interpolation, pil_sizes=sz) if self._resize_mode == "squash" else pre.resize_crop_u8(g, pil_sizes=sz))
            return outs[0] if len(outs) == 1 else torch.cat(outs)
        run = (lambda g: pre.resize_u8(g, S, S, self._interpolation)) if self._resize_mode == "squash" else pre.resize_crop_u8
        groups, cur, cur_bytes = [], [], 0
        for img in raw:
            nbytes = int(img.shape[0]) * int(img.shape[1]) * 4
            if cur and cur_bytes + nbytes > STAGE_BYTES:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(img)
            cur_bytes += nbytes
        if len(groups) == 0:
            return run(raw)
        groups.append(cur)
        return torch.cat([run(g) for g in groups])

    def _preprocess_images(self, images, image_download_headers: Optional[Dict] = None):
        """-> ('u8', uint8 [n,S,S,3]) or ('f32', fp32 [n,3,S,S]) on the device."""
        if self.model is None:
            self.load()
        headers = image_download_headers or dict()
        if isinstance(images, torch.Tensor) and images.ndim == 4:  # an already stacked batch
            return "f32", images

        def load(i):
            # decoded pixels (uint8 [H, W, 3] arrays, e.g. from a decoder pool) go to the pinned staging buffer as they are; the
            # reference wraps them in a PIL image first (image_download.py:107-108), which changes no pixel
            if isinstance(i, np.ndarray) and i.dtype == np.uint8 and i.ndim == 3 and i.shape[2] == 3:
                return i
            return format_and_load_CLIP_image(i, headers)
        if isinstance(images, list):
            # (a list of loaded Pillow RGB images — what format_and_load_CLIP_image would hand back unchanged — skips every per-image Python
            # step below: one native scan vouches for the whole batch, engine/preprocess.py::pil_rgb_sizes)
            from marqo_amd.engine.preprocess import pil_rgb_sizes
            sizes = pil_rgb_sizes(images) if len(images) >= 8 and os.environ.get("MARQO_AMD_PIL_BATCH_FAST", "1") != "0" else None
            if sizes is not None:
                return "u8", self._resize(self._pre(), images, pil_sizes=sizes)
            loaded = [load(i) for i in images]
        else:
            loaded = [load(images)]
        tensors = [i for i in loaded if isinstance(i, torch.Tensor)]
        pre = self._pre()
        if len(tensors) == len(loaded):
            runs = self._slab_runs(tensors)         # what `.preprocess` handed out: slices of the slab instead of a 256-tensor gather
            if runs is not None and len(runs) <= max(1, len(tensors) // 8):
                return "f32", runs[0] if len(runs) == 1 else torch.cat(runs)
            return "f32", torch.stack([t.to(self.device) for t in tensors])
        raw = [i if isinstance(i, np.ndarray) else pil_to_pixels(i) for i in loaded if not isinstance(i, torch.Tensor)]
        u8 = self._resize(pre, raw)
        if not tensors:
            return "u8", u8
        f32 = iter(pre.to_tensor_normalize(u8))
        return "f32", torch.stack([i.to(self.device) if isinstance(i, torch.Tensor) else next(f32) for i in loaded])

    # ---- encode ------------------------------------------------------------------------------------------------
    def _convert_output(self, output: torch.Tensor) -> np.ndarray:
        return output.cpu().numpy()

    def encode_image(self, images, image_download_headers: Optional[Dict] = None, normalize=True, return_device: bool = False):
        """-> np.ndarray [n, D] fp32 (reference contract), or with the engine extension `return_device=True` the same rows as a
        device tensor (no D2H: bulk ingest gathers shards over RCCL straight from HBM)"""
        if self.model is None:
            self.load()
        if not return_device and self.native_queue_takes_images(images):
            # a request thread's few `.preprocess` tensors (one document field): their addresses go to the image tower's native queue, this thread
            # blocks outside the interpreter while a worker gathers them with the other callers' and runs one tower call.  The tensors were written on
            # the producers' current stream — the device's default stream (`_preprocess_one`); the workers' streams are ordered behind nothing
            for st in (torch.cuda.current_stream(self.device), torch.cuda.default_stream(self.device)):
                if not st.query():
                    st.synchronize()
            rows = self.vision.queue_rows_images(images, bool(normalize))
            if rows is not None:
                return rows
        with self._image_calls_lock:
            self._image_calls += 1
            alone = self._image_calls == 1
        try:
            return self._encode_image(images, image_download_headers, normalize, return_device, alone)
        finally:
            with self._image_calls_lock:
                self._image_calls -= 1

    def _encode_image(self, images, image_download_headers, normalize, return_device: bool, alone: bool):
        with request_stream(self.device, device_output=return_device):
            run = lambda kind, px: (self.vision.encode_u8(px, normalize=bool(normalize)) if kind == "u8"
                                    else self.vision.encode_f32(px, normalize=bool(normalize)))
            # in stages (above), one D2H copy at the end — when this call has the model to itself: with other image calls in flight the GPU is
            # kept busy by them, and smaller towers only cost GEMM efficiency (4 concurrent 256-image callers: 77 k embeddings/s in one batch
            # each, 71 k staged; profiles/r05w_e2e_stages.txt).  Not for callers that take device rows either: they are the ingest pipelines
            # (ingest.py), which pack group g + 1 while group g runs — their chip-filling groups stay whole (the stream: 119.9 k embeddings/s at
            # gemm_frac 0.31 whole, 119.0 k at 0.26 staged)
            if isinstance(images, list) and images and isinstance(images[0], torch.Tensor) and getattr(images[0], "_mq_block", None) is not None:
                # what `.preprocess` handed out (the reference's download threads, add_docs.py:129-141): the images already sit side by side in
                # the preprocess slab — the tower reads the slab slice(s); no per-image Python, no gather, no staging (nothing to overlap: the
                # pixels are in HBM)
                runs = self._slab_runs(images)
                if runs is not None and len(runs) <= max(1, len(images) // 8):
                    cur = torch.cuda.current_stream(self.device)
                    for r in runs:
                        r.record_stream(cur)           # (the block was allocated on a download thread's stream)
                    px = runs[0] if len(runs) == 1 else torch.cat(runs)
                    self.image_input_processed = px
                    out = run("f32", px)
                    return out if return_device else self._convert_output(out)
            if isinstance(images, list) and len(images) >= PIPELINE_MIN and ((alone and not return_device) or PIPELINE_ALWAYS):
                outs, pxs = [], []
                main = torch.cuda.current_stream(self.device)
                sides = self._pipeline_streams(main) if PIPELINE_STREAMS > 1 else [main]
                stages = _pipeline_stages(len(images))
                helper = self._pipeline_helper() if PIPELINE_THREAD and max(b - a for a, b in stages) < PIPELINE_CHUNK else None

                def tower(st, kind, px):     # (on the helper thread: its own current-stream / current-device state)
                    with torch.cuda.device(self.device), torch.cuda.stream(st):
                        return run(kind, px)
                try:
                    for k, (a, b) in enumerate(stages):
                        st = sides[k % len(sides)]
                        with torch.cuda.stream(st):
                            kind, px = self._preprocess_images(images[a:b], image_download_headers)
                            o = helper.submit(tower, st, kind, px) if helper is not None else run(kind, px)
                        pxs.append(px)
                        outs.append(o)
                finally:      # (also when a later stage raised on this thread: nothing of this call is left running on the helper)
                    if helper is not None:
                        errs = [o.exception() for o in outs]
                if helper is not None:
                    for e in errs:
                        if e is not None:
                            raise e
                    outs = [o.result() for o in outs]
                for k, (px, o) in enumerate(zip(pxs, outs)):
                    if sides[k % len(sides)] is not main:      # allocated on a side stream, read by the request stream below
                        px.record_stream(main)
                        o.record_stream(main)
                for st in sides:
                    if st is not main:
                        main.wait_stream(st)
                self.image_input_processed = torch.cat(pxs) if len({(p.dtype, p.shape[1:]) for p in pxs}) == 1 else pxs[-1]
                out = torch.cat(outs)
            else:
                kind, px = self._preprocess_images(images, image_download_headers)
                self.image_input_processed = px
                out = run(kind, px)
            return out if return_device else self._convert_output(out)

    _pipeline_tls = threading.local()

    def _pipeline_helper(self):
        """this calling thread's helper (one worker: the stages' towers are enqueued in stage order)"""
        ex, pid = getattr(self._pipeline_tls, "helper", (None, None))
        if ex is None or pid != os.getpid():      # (a fork()ed child inherits the object but not its worker thread)
            from concurrent.futures import ThreadPoolExecutor
            ex = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mq-stage-tower")
            self._pipeline_tls.helper = (ex, os.getpid())
        return ex

    def _pipeline_streams(self, main) -> list:
        """[the request stream, this thread's side streams ...] of the two-stream image pipeline; the side streams start behind the request
        stream's current position"""
        side = getattr(self._pipeline_tls, "streams", None)
        if side is None or len(side) != PIPELINE_STREAMS - 1 or side[0].device != main.device:
            side = self._pipeline_tls.streams = [torch.cuda.Stream(main.device) for _ in range(PIPELINE_STREAMS - 1)]
        for st in side:
            st.wait_stream(main)
        return [main] + side

    def native_queue_takes_images(self, images) -> bool:
        """True when `encode_image(images)` goes through the image tower's native request queue: a short list of the views `.preprocess` handed out
        (`vectorise()` then leaves the merging of concurrent calls to it instead of the Python coalescer)"""
        from marqo_amd.engine import native_queue as NQ
        if not NQ.ENABLED or self.model is None or not isinstance(images, list) or not (1 <= len(images) <= NQ.IMAGE_REQUEST_MAX):
            return False
        if getattr(self, "vision", None) is None or not hasattr(self.vision, "queue_rows_images"):
            return False
        return all(isinstance(t, torch.Tensor) and getattr(t, "_mq_block", None) is not None for t in images)

    def native_queue_takes(self, texts) -> bool:
        """True when `encode_text(texts)` goes through the text tower's native request queue (engine/native_queue.py): `vectorise()` then leaves
        the merging of concurrent calls to it instead of the Python coalescer.  Host-tokenised small calls only — ids from the device tokeniser
        stay in HBM and take the direct path."""
        from marqo_amd.engine import native_queue as NQ
        if not NQ.ENABLED or self.model is None or getattr(self, "text", None) is None or not hasattr(self.text, "_small_call"):
            return False
        texts = [texts] if isinstance(texts, str) else texts
        if not (1 <= len(texts) <= NQ.MAX_SEQS) or not all(isinstance(t, str) for t in texts):
            return False
        return getattr(self, "_device_tokenizer", None) is None or prefers_host(texts)

    def encode_text(self, sentence: Union[str, List[str]], normalize=True, return_device: bool = False):
        if self.model is None:
            self.load()
        ids_np = None
        if not return_device and not isinstance(self.text_arch, archs.HfClipTextArch) and self.native_queue_takes(sentence):
            # a request thread's small call: tokenise here, hand the ids to the tower's native queue, block outside the interpreter; a LONE single
            # query comes back None and replays its captured graph below (with the ids made here)
            ids_np = np.asarray(self.tokenizer(sentence))
            rows = self.text.queue_rows_ids(ids_np, bool(normalize))
            if rows is not None:
                return rows
        with request_stream(self.device, device_output=return_device):
            if isinstance(self.text_arch, archs.HfClipTextArch):
                texts = [_clean_text(t) for t in ([sentence] if isinstance(sentence, str) else list(sentence))]
                if getattr(self, "_device_tokenizer", None) is not None and not prefers_host(texts):
                    d_ids, lens = self._device_tokenizer.encode_device(texts, self.text_arch.ctx)
                    out = self.text.encode_device(d_ids, lens, normalize=bool(normalize))
                else:
                    out = self.text.encode_padded(torch.as_tensor(self.tokenizer.ids(texts)), normalize=bool(normalize))
            elif getattr(self, "_device_tokenizer", None) is not None and not prefers_host([sentence] if isinstance(sentence, str) else sentence):
                texts = [sentence] if isinstance(sentence, str) else list(sentence)
                if self.text_arch.causal:
                    d_ids, lens = self._device_tokenizer.encode_device(texts)
                else:  # SigLIP: every row is ctx positions (pieces, </s>, </s> padding) and all of them run
                    d_ids, _ = self._device_tokenizer.encode_device(texts, self.text_arch.ctx)
                    lens = torch.full((len(texts),), self.text_arch.ctx, dtype=torch.int64)
                out = self.text.encode_device(d_ids, lens, normalize=bool(normalize))
            else:
                ids = torch.as_tensor(ids_np if ids_np is not None else np.asarray(self.tokenizer(sentence)))
                out = self.text.encode_ids(ids, normalize=bool(normalize))
            return out if return_device else self._convert_output(out)

    # engine extensions used by the chunking / bulk-ingest path ------------------------------------------------------
    def encode_image_chunks(self, images: Sequence, hn: int = 3, wn: int = 3, overlap: bool = False, normalize=True):
        """'simple' / 'overlap' patch methods entirely on the device: -> (embeddings [n, count, D], boxes [n, count, 4])."""
        raw = [pil_to_pixels(i) if isinstance(i, ImageType) else np.asarray(i) for i in images]
        u8, boxes = self._pre().chunk_grid_u8(raw, hn, wn, overlap)
        if self._resize_mode == "squash" and len(raw):
            # the grid crops are square (shorter-side resize + crop == squash); chunk 0, the whole image, is not
            u8[0::u8.shape[0] // len(raw)] = self._resize(self._pre(), raw)
        with request_stream(self.device):
            emb = self._convert_output(self.vision.encode_u8(u8, normalize=bool(normalize)))
        return emb.reshape(len(raw), -1, emb.shape[-1]), boxes


class CLIP(OPEN_CLIP):
    """OpenAI-CLIP names ('ViT-B/32', 'ViT-L/14' ...), legacy loader signature (s2_inference.py:559-566):
    CLIP(name, device=, embedding_dim=, model_properties=, model_auth=, max_seq_length=).  Same towers with QuickGELU."""

    def __init__(self, model_type: str = "ViT-B/32", device: str = None, embedding_dim: int = None, truncate: bool = True,
                 model_properties: Optional[dict] = None, model_auth=None, **kwargs) -> None:
        props = dict(model_properties or {})
        name = props.get("name", model_type)
        base = name[len("fp16/"):] if name.startswith("fp16/") else name
        if base in archs.OPENAI_CLIP_NAMES and not (props.get("localpath") or props.get("url")):
            props["name"] = f"open_clip/{archs.OPENAI_CLIP_NAMES[base]}/openai"
        elif base in archs.OPENAI_CLIP_NAMES:
            props["name"] = archs.OPENAI_CLIP_NAMES[base] + "-quickgelu"
        props.setdefault("dimensions", embedding_dim)
        props.setdefault("type", "clip")
        super().__init__(device=device, model_properties=props, model_auth=model_auth)
        self.model_type = model_type
        self.truncate = truncate

    def encode_image(self, images, normalize=True, image_download_headers: Optional[Dict] = None, return_device: bool = False):
        """the legacy loaders take `normalize` BEFORE the headers (clip_utils.py:397-399; OPEN_CLIP has them the other way round,
        open_clip_model.py:249-251)"""
        return super().encode_image(images, image_download_headers=image_download_headers, normalize=normalize, return_device=return_device)


class FP16_CLIP(CLIP):
    """'fp16/ViT-*' names (clip_utils.py:495-518: cuda only).  On MI355X every CLIP tower already runs bf16 MFMA."""

    def __init__(self, model_type: str = "fp16/ViT-B/32", device: str = None, embedding_dim: int = None, truncate: bool = True,
                 model_properties: Optional[dict] = None, model_auth=None, **kwargs) -> None:
        from marqo_amd.s2_inference.errors import IncompatibleModelDeviceError
        if not str(device).startswith("cuda"):
            raise IncompatibleModelDeviceError("FP16 clip model `{}` is only available with device `cuda`.".format(model_type))
        super().__init__(model_type, device, embedding_dim, truncate, model_properties, model_auth, **kwargs)


# ---- multilingual_clip ------------------------------------------------------------------------------------------------------------------
def get_multilingual_clip_properties() -> Dict:
    """the reference's multilingual_clip registry entries (clip_utils.py:599-641): an OpenAI / open_clip image tower paired with an M-CLIP
    text encoder (github.com/FreddeFrallan/Multilingual-CLIP)"""
    return {
        "multilingual-clip/XLM-Roberta-Large-Vit-L-14": {
            "name": "multilingual-clip/XLM-Roberta-Large-Vit-L-14", "visual_model": "openai/ViT-L/14",
            "textual_model": "M-CLIP/XLM-Roberta-Large-Vit-L-14", "dimensions": 768, "type": "multilingual_clip"},
        "multilingual-clip/XLM-R Large Vit-B/16+": {
            "name": "multilingual-clip/XLM-R Large Vit-B/16+", "visual_model": "open_clip/ViT-B-16-plus-240/laion400m_e32",
            "textual_model": "M-CLIP/XLM-Roberta-Large-Vit-B-16Plus", "dimensions": 640, "type": "multilingual_clip"},
        "multilingual-clip/XLM-Roberta-Large-Vit-B-32": {
            "name": "multilingual-clip/XLM-Roberta-Large-Vit-B-32", "visual_model": "openai/ViT-B/32",
            "textual_model": "M-CLIP/XLM-Roberta-Large-Vit-B-32", "dimensions": 512, "type": "multilingual_clip"},
        "multilingual-clip/LABSE-Vit-L-14": {
            "name": "multilingual-clip/LABSE-Vit-L-14", "visual_model": "openai/ViT-L/14",
            "textual_model": "M-CLIP/LABSE-Vit-L-14", "dimensions": 768, "type": "multilingual_clip"},
    }


# M-CLIP text encoders: `modelBase` of their MCLIPConfig -> the encoder architecture (the M-CLIP repos carry no Hugging Face config of it)
_MCLIP_BASES = {
    "xlm-roberta-large": archs.BertArch(vocab=250002, max_pos=512, width=1024, layers=24, heads=16, mlp_dim=4096, ln_eps=1e-5, pos_offset=2,
                                        type_vocab=1),
    "sentence-transformers/LaBSE": archs.BertArch(vocab=501153, max_pos=512),
}
_MCLIP_TEXTUAL = {"M-CLIP/XLM-Roberta-Large-Vit-L-14": "xlm-roberta-large", "M-CLIP/XLM-Roberta-Large-Vit-B-16Plus": "xlm-roberta-large",
                  "M-CLIP/XLM-Roberta-Large-Vit-B-32": "xlm-roberta-large", "M-CLIP/LABSE-Vit-L-14": "sentence-transformers/LaBSE"}


class MULTILINGUAL_CLIP(OPEN_CLIP):
    """`multilingual_clip` loader type (clip_utils.py:521-597): `MULTILINGUAL_CLIP(model_type, device=, embedding_dim=, truncate=, **kwargs)`;
    `visual_model` = the image tower of an OpenAI / open_clip checkpoint (the projected image features, as `model.visual.forward`),
    `textual_model` = pt_multilingual_clip.MultilingualCLIP: HF encoder -> attention-masked mean -> one biased Linear
    (engine/towers.py::MclipTextTower), tokenised by the textual model's own HF tokenizer with padding (sequences beyond the encoder's
    512 positions, which the reference cannot run at all, are truncated)."""
    _own_text_tower = True

    def __init__(self, model_type: str = "multilingual-clip/ViT-L/14", device: str = None, embedding_dim: int = None, truncate: bool = True,
                 model_properties: Optional[dict] = None, model_auth=None, **kwargs) -> None:
        from marqo_amd.s2_inference.errors import InternalError
        if not device:
            raise InternalError("`device` is required for loading MULTILINGUAL CLIP models!")
        table = get_multilingual_clip_properties()
        if model_type not in table:
            raise InvalidModelPropertiesError(f"unknown multilingual_clip model {model_type!r}")
        self.model_name = model_type
        self.model_info = table[model_type]
        self.visual_name, self.textual_name = self.model_info["visual_model"], self.model_info["textual_model"]
        v = self.visual_name
        oc = (f"open_clip/{archs.OPENAI_CLIP_NAMES[v[len('openai/'):]]}/openai" if v.startswith("openai/") else v)
        props = {"name": oc, "dimensions": self.model_info["dimensions"], "type": "open_clip"}
        for k in ("enginePrecision", "fp8Budget"):
            if model_properties and k in model_properties:
                props[k] = model_properties[k]
        super().__init__(device=device, model_properties=props, model_auth=model_auth)
        self.truncate = truncate

    def encode_image(self, images, normalize=True, image_download_headers: Optional[Dict] = None, return_device: bool = False):
        """legacy parameter order (clip_utils.py:573-575)"""
        return super().encode_image(images, image_download_headers=image_download_headers, normalize=normalize, return_device=return_device)

    def _make_text_tower(self, sd, precision: str):
        from marqo_amd.engine import towers
        base = _MCLIP_TEXTUAL.get(self.textual_name)
        bert_arch = _MCLIP_BASES[base]
        self._text_dir = checkpoint.find_hf_dir(self.textual_name)
        if self._text_dir is not None:
            cfg, tsd = checkpoint.load_hf_dir(self._text_dir)
            if cfg.get("numDims", self.model_info["dimensions"]) != self.model_info["dimensions"]:
                raise InvalidModelPropertiesError(f"{self.textual_name}: numDims {cfg.get('numDims')} != {self.model_info['dimensions']}")
        elif checkpoint.synthetic_weights_enabled():
            tsd = {"transformer." + k: v for k, v in synthetic.random_bert_state_dict(bert_arch, seed=1).items()}
            g = torch.Generator().manual_seed(2)
            D, W = self.model_info["dimensions"], bert_arch.width
            tsd["LinearTransformation.weight"] = torch.randn(D, W, generator=g) / W ** 0.5
            tsd["LinearTransformation.bias"] = 0.02 * torch.randn(D, generator=g)
        else:
            raise ModelLoadError(f"no checkpoint for {self.textual_name} under {checkpoint.model_dir()} (there is no network download in the "
                                 f"marqo_amd engine; set MARQO_AMD_SYNTHETIC_WEIGHTS=1 for random-init weights)")
        self.text_arch = archs.HfClipTextArch(bert=bert_arch, out_dim=self.model_info["dimensions"], ctx=bert_arch.max_pos)
        return towers.MclipTextTower(bert_arch, self.model_info["dimensions"], tsd, self.device, precision=precision)

    def _load_tokenizer(self, ckpt_dir: Optional[str]):
        d = getattr(self, "_text_dir", None)
        if d is not None:
            if os.path.isfile(os.path.join(d, "sentencepiece.bpe.model")):
                return XlmRobertaTokenizer(d)
            if os.path.isfile(os.path.join(d, "vocab.txt")):
                return WordPieceTokenizer(d, do_lower_case=False)   # LaBSE: cased WordPiece
            raise ModelLoadError(f"{self.textual_name}: no tokenizer files (sentencepiece.bpe.model / vocab.txt) in {d}")
        kind = "xlmr" if self.text_arch.bert.pos_offset else "bert"
        return SyntheticTokenizer(kind, self.text_arch.vocab)

    def encode_text(self, sentence: Union[str, List[str]], normalize=True, return_device: bool = False):
        if self.model is None:
            self.load()
        texts = [sentence] if isinstance(sentence, str) else list(sentence)
        max_len = self.text_arch.bert.max_pos
        with request_stream(self.device, device_output=return_device):
            if getattr(self, "_device_tokenizer", None) is not None and not prefers_host(texts):
                d_ids, lens = self._device_tokenizer.encode_device(texts, max_len)
                out = self.text.encode_device(d_ids, lens, normalize=bool(normalize))
            else:
                tok = self.tokenizer(texts, max_length=max_len)
                out = self.text.encode_ids(torch.from_numpy(tok["input_ids"]), torch.from_numpy(tok["attention_mask"]), normalize=bool(normalize))
            return out if return_device else self._convert_output(out)


# ---- clip_onnx ------------------------------------------------------------------------------------------------------------------------
class CLIP_ONNX(OPEN_CLIP):
    """`clip_onnx` loader type (onnx_clip_utils.py:55-175): `onnx32/...` / `onnx16/...` names are ONNX exports of OpenAI / open_clip CLIP
    checkpoints (visual + textual graphs run by onnxruntime in fp32 / fp16).  Same weights: here they are served by the HIP towers of the
    checkpoint they were exported from (there is no onnxruntime in the engine; both precisions run the bf16 MFMA path), behind the
    reference's constructor `CLIP_ONNX(model_name, device=, embedding_dim=, truncate=, load=True)`.  Only the ViT exports are registered."""

    def __init__(self, model_name: str = "onnx32/openai/ViT-L/14", device: str = None, embedding_dim: int = None, truncate: bool = True,
                 load: bool = True, model_properties: Optional[dict] = None, model_auth=None, **kwargs) -> None:
        from marqo_amd.s2_inference.errors import InternalError
        if not device:
            raise InternalError("`device` is required for loading CLIP ONNX models!")
        try:
            self.onnx_type, self.source, self.clip_model = model_name.split("/", 2)
        except ValueError:
            raise InvalidModelPropertiesError(f"clip_onnx names look like onnx32/<source>/<model>, got {model_name!r}")
        if self.onnx_type not in ("onnx16", "onnx32") or self.source not in ("openai", "open_clip"):
            raise InvalidModelPropertiesError(f"clip_onnx names look like onnx16|onnx32/openai|open_clip/<model>, got {model_name!r}")
        if self.source == "openai":
            if self.clip_model not in archs.OPENAI_CLIP_NAMES:
                raise InvalidModelPropertiesError(f"{model_name}: {archs.UNSUPPORTED_HINT}")
            name = f"open_clip/{archs.OPENAI_CLIP_NAMES[self.clip_model]}/openai"
        else:
            name = "open_clip/" + self.clip_model
        props = {"name": name, "dimensions": (model_properties or {}).get("dimensions", embedding_dim), "type": "open_clip"}
        for k in ("enginePrecision", "fp8Budget"):
            if model_properties and k in model_properties:
                props[k] = model_properties[k]
        super().__init__(device=device, model_properties=props, model_auth=model_auth)
        self.model_name, self.truncate = model_name, truncate

    def encode_image(self, images, normalize=True, image_download_headers: Optional[Dict] = None, return_device: bool = False):
        """onnx_clip_utils.py:126: (images, normalize=True) — headers are accepted as an extra keyword"""
        return super().encode_image(images, image_download_headers=image_download_headers, normalize=normalize, return_device=return_device)
