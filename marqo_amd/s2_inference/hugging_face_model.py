"""`hf` / `hf_stella` loaders of the engine.

Drop-in for HuggingFaceModel (src/marqo/core/inference/embedding_models/hugging_face_model.py:24-214): same ctor,
load(), encode(sentence, normalize=True, **kwargs) -> np.ndarray[N, D] fp32.  Tokenisation keeps the reference's flags
(padding=True, truncation=True, max_length=tokens, :179-185); the BERT encoder, masked mean / CLS pooling (:205-214) and
F.normalize (:194-195) run in libmarqo_hip.so on packed (un-padded) sequences.
"""
from __future__ import annotations

import os
import threading
from typing import List, Optional, Union

import numpy as np
import torch

from marqo_amd import _lib as L
from marqo_amd.engine import archs, checkpoint, synthetic
from marqo_amd.engine.gpu_tokenizers import prefers_host
from marqo_amd.engine.towers import request_stream
from marqo_amd.engine.tokenizers import SyntheticTokenizer, WordPieceTokenizer, XlmRobertaTokenizer
from marqo_amd.s2_inference.abstract_models import AbstractEmbeddingModel
from marqo_amd.s2_inference.errors import InternalError, InvalidModelPropertiesError, ModelLoadError


class HuggingFaceModelProperties:
    """Validated view of model_properties (reference: hugging_face_model_properties.py:40-137)."""

    def __init__(self, **p):
        dims = p.get("dimensions")
        if not isinstance(dims, int) or dims < 1:
            raise ValueError("'dimensions' must be a positive integer")
        self.dimensions: int = dims
        self.type: str = p.get("type")
        if self.type not in ("hf", "hf_stella"):
            raise ValueError("The type of the model should be 'hf' or 'hf_stella'.")
        self.name: Optional[str] = p.get("name")
        self.tokens: int = int(p.get("tokens", 128))
        self.url: Optional[str] = p.get("url")
        self.model_location = p.get("model_location", p.get("modelLocation"))
        if self.url and self.model_location:
            raise ValueError("Only one of 'url' and 'model_location' should be provided.")
        if not self.name and not (self.url or self.model_location):
            raise ValueError("At least one of 'name', 'url', or 'model_location' should be provided.")
        pm = p.get("pooling_method", p.get("poolingMethod"))
        if pm is not None and pm not in ("mean", "cls"):
            raise ValueError("pooling_method must be 'mean' or 'cls'")
        self.pooling_method: Optional[str] = pm  # None: inferred at load from 1_Pooling/config.json, default mean
        self.trust_remote_code: bool = bool(p.get("trust_remote_code", p.get("trustRemoteCode", False)))
        # engine extension: operand type of the encoder-block GEMMs, "bf16" (default) or "fp8" (e4m3 MX-MFMA; calibrated at load on
        # fixed seeded inputs, 'fp8Budget' bounds the error), see open_clip_model.OpenCLIPModelProperties
        self.engine_precision: str = p.get("enginePrecision", p.get("engine_precision", os.environ.get("MARQO_AMD_PRECISION", "bf16")))
        if self.engine_precision not in ("bf16", "fp8"):
            raise ValueError("'enginePrecision' must be 'bf16' or 'fp8'")
        b = p.get("fp8Budget", p.get("fp8_budget"))
        if b is not None and not (isinstance(b, (int, float)) and b > 0):
            raise ValueError("'fp8Budget' must be a positive number")
        self.fp8_budget: Optional[float] = None if b is None else float(b)

    def dict(self) -> dict:
        return dict(self.__dict__)


class HuggingFaceModel(AbstractEmbeddingModel):
    supports_dynamic_batching = True
    _requires_trust_remote_code = False
    _check_dimensions = True

    def __init__(self, model_properties: dict, device: str, model_auth=None, model_flags=None, tokenizer_flags=None):
        super().__init__(model_properties, device, model_auth)
        self.model_properties = self._build_model_properties(model_properties or {})
        self._model = None
        self._tokenizer = None
        self._pooling_func = None
        self.weights_source = None
        self._calib_lock = threading.Lock()

    def _build_model_properties(self, model_properties: dict) -> HuggingFaceModelProperties:
        try:
            parsed = HuggingFaceModelProperties(**model_properties)
        except (ValueError, TypeError) as e:
            raise InvalidModelPropertiesError(f"Invalid model properties: {model_properties}. Original error {e}") from e
        if self._requires_trust_remote_code and not parsed.trust_remote_code:
            raise InvalidModelPropertiesError("The specified model requires the 'trustRemoteCode' attribute to be set to True. "
                                              "Setting this attribute to True may have security implications.")
        return parsed

    def _check_loaded_components(self):
        if self._model is None:
            raise InternalError("Model is not loaded!")
        if self._tokenizer is None:
            raise InternalError("Tokenizer is not loaded!")
        if self._pooling_func is None:
            raise InternalError("Pooling function is not loaded!")

    def _load_necessary_components(self):
        if not str(self.device).startswith("cuda"):
            raise L.MarqoHipUnavailableError(
                f"marqo_amd runs its towers on an AMD GPU only (device 'cuda' / 'cuda:N' on ROCm); got {self.device!r}")
        from marqo_amd.engine import towers
        props = self.model_properties
        if not props.name:
            raise ModelLoadError("downloading model archives ('url' / 'model_location') is control-plane work outside the marqo_amd "
                                 "engine; unpack the archive on disk and pass its directory as 'name'")
        directory = checkpoint.find_hf_dir(props.name)
        if directory is not None:
            cfg, sd = checkpoint.load_hf_dir(directory)
            try:
                arch = archs.bert_arch_from_hf_config(cfg)
            except KeyError as e:
                raise InvalidModelPropertiesError(f"{props.name}: {e}. Only BERT-family encoders run on the marqo_amd engine.") from e
            if (arch.glu or arch.rope_theta is not None) and not props.trust_remote_code:
                # a NewModel checkpoint is custom remote code to the reference's AutoModel: refused unless trustRemoteCode is set
                raise InvalidModelPropertiesError(f"{props.name} is a custom-code (NewModel) checkpoint: the reference loads it only with "
                                                  f"'trustRemoteCode': True (type 'hf_stella')")
            if os.path.isfile(os.path.join(directory, "sentencepiece.bpe.model")):  # XLM-RoBERTa checkpoints (multilingual-e5)
                self._tokenizer = XlmRobertaTokenizer(directory)
            elif arch.rel_buckets:
                # MPNetTokenizer = BERT's basic + WordPiece tokenisation around "<s> ... </s>" (transformers tokenization_mpnet.py); the
                # special-token spellings come from the checkpoint's tokenizer_config.json / special_tokens_map.json
                self._tokenizer = WordPieceTokenizer(directory, do_lower_case=self._do_lower_case(directory),
                                                     **self._special_tokens(directory, unk="[UNK]", cls="<s>", sep="</s>", pad="<pad>", mask="<mask>"))
            else:
                self._tokenizer = WordPieceTokenizer(directory, do_lower_case=self._do_lower_case(directory))
            self.weights_source = directory
        elif checkpoint.synthetic_weights_enabled() and props.name in archs.HF_BERT_ARCHS:
            arch = archs.HF_BERT_ARCHS[props.name]
            sd = synthetic.random_bert_state_dict(arch, seed=0)
            self._tokenizer = SyntheticTokenizer("bert", arch.vocab)  # (for XLM-R archs too: ids only need to be in range)
            self.weights_source = "synthetic(seed=0)"
        else:
            raise InvalidModelPropertiesError(
                f"Marqo encountered an error loading the Hugging Face model, modelProperties={props.dict()}. No local copy under "
                f"{checkpoint.model_dir()} or the Hugging Face cache (there is no network download in the marqo_amd engine).")
        if self._check_dimensions and arch.width != props.dimensions:
            raise InvalidModelPropertiesError(f"'dimensions'={props.dimensions} but the encoder width is {arch.width}")
        pooling = props.pooling_method or checkpoint.read_pooling_config(directory) or "mean"
        self.arch = arch
        try:
            self._model = towers.BertTower(arch, sd, self.device, pooling=pooling, precision=props.engine_precision)
        except ValueError as e:  # e.g. fp8 needs width / mlp_dim multiples of 128
            raise InvalidModelPropertiesError(str(e)) from e
        self._pooling_func = pooling
        if props.engine_precision == "fp8":   # deterministic load-time calibration on fixed seeded inputs (see open_clip_model.py)
            self._model.tune_fp8_default(props.fp8_budget)
        self._model.release_unused_folded()
        # K14: WordPiece on the device for ASCII texts (identical ids; the host tokeniser stays the definition of record and
        # handles every other text).  MARQO_AMD_HOST_TOKENIZER=1 keeps everything on the host.
        self._device_tokenizer = None
        if isinstance(self._tokenizer, WordPieceTokenizer) and os.environ.get("MARQO_AMD_HOST_TOKENIZER", "0") != "1":
            from marqo_amd.engine.gpu_tokenizers import DeviceWordPieceTokenizer
            self._device_tokenizer = DeviceWordPieceTokenizer(self._tokenizer, self.device)
        elif isinstance(self._tokenizer, XlmRobertaTokenizer) and os.environ.get("MARQO_AMD_HOST_TOKENIZER", "0") != "1":
            from marqo_amd.engine.gpu_tokenizers import DeviceSentencePieceTokenizer
            try:   # unigram Viterbi on the device (multilingual-e5); models it cannot express (byte fallback ...) stay on the host
                self._device_tokenizer = DeviceSentencePieceTokenizer(self._tokenizer, self.device)
            except ValueError:
                self._device_tokenizer = None

    @staticmethod
    def _special_tokens(directory: str, **defaults) -> dict:
        """unk / cls / sep / pad / mask spellings of a checkpoint (tokenizer_config.json, then special_tokens_map.json), else the family's"""
        import json
        out = dict(defaults)
        for name in ("special_tokens_map.json", "tokenizer_config.json"):
            p = os.path.join(directory, name)
            if not os.path.isfile(p):
                continue
            try:
                with open(p) as f:
                    j = json.load(f)
            except (OSError, ValueError):
                continue
            for k in out:
                v = j.get(k + "_token")
                if isinstance(v, dict):   # AddedToken serialisation
                    v = v.get("content")
                if isinstance(v, str) and v:
                    out[k] = v
        return out

    @staticmethod
    def _do_lower_case(directory: str) -> bool:
        import json
        for name in ("tokenizer_config.json",):
            p = os.path.join(directory, name)
            if os.path.isfile(p):
                try:
                    with open(p) as f:
                        return bool(json.load(f).get("do_lower_case", True))
                except (OSError, ValueError):
                    pass
        return True

    def native_queue_takes(self, texts) -> bool:
        """True when `encode(texts)` goes through the tower's native request queue (engine/native_queue.py): `vectorise()` then leaves the merging
        of concurrent calls to it instead of the Python coalescer (host-tokenised small calls only)"""
        from marqo_amd.engine import native_queue as NQ
        if not NQ.ENABLED or self._model is None or not hasattr(self._model, "_small_call"):
            return False
        texts = [texts] if isinstance(texts, str) else texts
        if not (1 <= len(texts) <= NQ.MAX_SEQS) or not all(isinstance(t, str) for t in texts):
            return False
        return getattr(self, "_device_tokenizer", None) is None or prefers_host(texts)

    def encode(self, sentence: Union[str, List[str]], normalize=True, **kwargs) -> np.ndarray:
        """-> np.ndarray [n, D] fp32 (hugging_face_model.py:172-203); engine extension `return_device=True`: the same rows as a device
        tensor.  Other kwargs the callers pass (`modality`, `infer`, `image_download_headers`) are tolerated and ignored."""
        if isinstance(sentence, str):
            sentence = [sentence]
        if self._model is None:
            self.load()
        return_device = bool(kwargs.get("return_device", False))
        tok = None
        if not return_device and self.native_queue_takes(sentence):
            # a request thread's small call: tokenise here, hand the ids to the tower's native queue, block outside the interpreter; a LONE single
            # query comes back None and replays its captured graph below (with the ids made here)
            tok = self._tokenizer(sentence, max_length=self.model_properties.tokens)
            rows = self._model.queue_rows_ids(tok["input_ids"], tok["attention_mask"], bool(normalize))
            if rows is not None:
                return rows
        with request_stream(self.device, device_output=return_device):
            if tok is None and getattr(self, "_device_tokenizer", None) is not None and not prefers_host(sentence):
                d_ids, lens = self._device_tokenizer.encode_device(sentence, self.model_properties.tokens)
                out = self._model.encode_device(d_ids, lens, normalize=bool(normalize))
            else:
                if tok is None:
                    tok = self._tokenizer(sentence, max_length=self.model_properties.tokens)
                ids, mask = torch.from_numpy(tok["input_ids"]), torch.from_numpy(tok["attention_mask"])
                out = self._model.encode_ids(ids, mask, normalize=bool(normalize))
            return out if return_device else out.cpu().numpy()


class HuggingFaceStellaModel(HuggingFaceModel):
    """hf_stella (hugging_face_stella_model.py:9-23): the reference loads `Marqo/dunzhang-stella_en_400M_v5` through
    AutoModel(trust_remote_code=True, use_memory_efficient_attention=False, unpad_inputs=False) — Alibaba-NLP's `NewModel` encoder
    (rotary positions, packed qkv, gated-GELU MLP, post-LN) — and keeps HuggingFaceModel.encode as is: tokenizer -> last_hidden_state
    -> mean pooling -> F.normalize (the sentence-transformers `2_Dense` head is NOT applied: AutoModel returns the bare encoder, and
    the registry's 1024 dimensions are its hidden size).  The engine runs that encoder natively (engine/towers.py::BertTower with
    BertArch.glu / rope_theta; kernels rope_kernel / glu_kernel): no remote code is executed, `trustRemoteCode` is only validated like
    the reference validates it."""
    _requires_trust_remote_code = True
