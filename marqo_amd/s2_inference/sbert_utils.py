"""`sbert` / `test` loader types of the reference (src/marqo/s2_inference/sbert_utils.py:11-111) on the MI355X towers.

The reference wraps `sentence_transformers.SentenceTransformer(model_name)` and calls `model.encode(..., normalize_embeddings=False)`
followed by its own `F.normalize`.  A SentenceTransformer checkpoint of the registry's sbert entries is a pipeline read from
`modules.json`: `Transformer` (a Hugging Face BERT / MPNet / XLM-RoBERTa encoder, max_seq_length from `sentence_bert_config.json`) ->
`1_Pooling` (mean over the attention mask for every registry entry) [-> `2_Normalize`].  That is exactly the `hf` path of this engine
(marqo_amd/s2_inference/hugging_face_model.py: tokenizer -> BertTower -> masked mean -> L2), so the loaders below are the reference's
constructor / `encode` contract over it:

  * `SBERT(model_name, device=, embedding_dim=, max_seq_length=, ...)`, `.load()`, `.encode(sentence, normalize=True)` -> ndarray
    (`_convert_output`: numpy on every device);
  * a checkpoint whose pipeline ends in a `Normalize` module returns unit vectors even for `normalize=False`, as it does in the reference
    (`model.encode` runs the whole pipeline);
  * `TEST` keeps the first 16 dimensions before normalising (sbert_utils.py:80-111) and returns a tensor, as the reference does.

There is no CPU path: a non-cuda device raises like every other engine loader."""
from __future__ import annotations

import json
import os
from typing import List, Optional, Union

import numpy as np
import torch

from marqo_amd.engine import checkpoint
from marqo_amd.s2_inference.errors import InternalError
from marqo_amd.s2_inference.hugging_face_model import HuggingFaceModel


class Model:
    """generic model wrapper class (sbert_utils.py:11-36)"""

    def __init__(self, model_name: Optional[str] = None, device: str = None, batch_size: int = 2048, embedding_dim=None,
                 max_seq_length=None, **kwargs) -> None:
        self.model_name = model_name
        if not device:
            raise InternalError("`device` is required to be set when loading models!")
        self.device = device
        self.model = None
        self.embedding_dimension = embedding_dim
        self.batch_size = batch_size
        self.max_seq_length = max_seq_length

    def load(self) -> None:
        pass

    def encode(self, sentence: Union[str, List[str]]) -> None:
        pass


class _SentenceTransformerHF(HuggingFaceModel):
    """the `hf` loader without its 'dimensions == encoder width' check (the `test` entries register a truncated width)"""
    _check_dimensions = False


def _sentence_transformer_info(directory: Optional[str]):
    """(max_seq_length or None, pipeline ends in Normalize?) of a SentenceTransformer checkpoint directory"""
    max_len, normalizes = None, False
    if not directory:
        return max_len, normalizes
    try:
        with open(os.path.join(directory, "sentence_bert_config.json")) as f:
            max_len = json.load(f).get("max_seq_length")
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(directory, "modules.json")) as f:
            mods = json.load(f)
        normalizes = any(str(m.get("type", "")).endswith("models.Normalize") for m in mods if isinstance(m, dict))
    except (OSError, ValueError):
        pass
    return max_len, normalizes


class SBERT(Model):
    """class for SBERT models (sbert_utils.py:39-77)"""
    supports_dynamic_batching = True

    def __init__(self, *args, model_properties: Optional[dict] = None, model_auth=None, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.model_properties = dict(model_properties or {})
        self.always_normalized = False

    def load(self) -> None:
        directory = checkpoint.find_hf_dir(self.model_name)
        st_len, self.always_normalized = _sentence_transformer_info(directory)
        # if one provided, overwrite (sbert_utils.py:50-54)
        if self.max_seq_length is None:
            self.max_seq_length = st_len or 128
        dims = self.model_properties.get("dimensions", self.embedding_dimension)
        props = {"name": self.model_name, "dimensions": dims or 1, "tokens": int(self.max_seq_length), "type": "hf", "poolingMethod": "mean"}   # (the width check is off: `test` entries register a truncated width, direct construction may give none)
        for k in ("enginePrecision", "fp8Budget"):
            if k in self.model_properties:
                props[k] = self.model_properties[k]
        pooled = checkpoint.read_pooling_config(directory)
        if pooled:
            props["poolingMethod"] = pooled
        self.model = _SentenceTransformerHF(props, self.device)
        self.model.load()

    def _convert_output(self, output):
        return output if isinstance(output, np.ndarray) else output.cpu().numpy()

    def _embed(self, sentence, normalize: bool, **kwargs):
        if self.model is None:
            self.load()
        if isinstance(sentence, str):
            sentence = [sentence]
        return self.model.encode(sentence, normalize=bool(normalize) or self.always_normalized, **kwargs)

    def native_queue_takes(self, texts) -> bool:
        """the wrapped Hugging Face loader's answer (engine/native_queue.py): its small text calls merge in the tower's native request queue, so
        `vectorise()` keeps them out of the Python coalescer.  (TEST truncates device rows and never takes that path.)"""
        return type(self) is SBERT and self.model is not None and self.model.native_queue_takes(texts)

    def encode(self, sentence: Union[str, List[str]], normalize=True, **kwargs) -> np.ndarray:
        kw = {"return_device": True} if kwargs.get("return_device") else {}
        out = self._embed(sentence, normalize, **kw)
        return out if kw else self._convert_output(out)


class TEST(SBERT):
    """the reference's plumbing model (sbert_utils.py:80-111): the first 16 dimensions of an SBERT model, normalised afterwards"""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.truncated_embedding_dim = 16

    def encode(self, sentence: Union[str, List[str]], normalize: bool = True, **kwargs) -> torch.Tensor:
        emb = self._embed(sentence, False, return_device=True)[:, :self.truncated_embedding_dim].contiguous()
        if normalize:   # F.normalize of the truncated rows, on the device (mq_l2_normalize)
            from marqo_amd import _lib as L
            with torch.cuda.device(emb.device):
                L.check(L.load().mq_l2_normalize(emb.data_ptr(), emb.data_ptr(), emb.shape[0], emb.shape[1],
                                                 torch.cuda.current_stream(emb.device).cuda_stream), "mq_l2_normalize")
        return emb if kwargs.get("return_device") else emb.cpu()


class SBERT_ONNX(SBERT):
    """`sbert_onnx` loader type (sbert_onnx_utils.py:19-218: the encoder exported to ONNX, run by onnxruntime, then the loader's own
    attention-masked mean and optional L2).  Same weights, same arithmetic as the `sbert` entries: here the `onnx/*` names are served by
    the HIP towers (there is no onnxruntime in the engine); the checkpoint's SentenceTransformer Normalize module is NOT applied, as in the
    reference's exported graph.  Constructor as the reference: SBERT_ONNX(model_name_or_path, device=, embedding_dim=, max_seq_length=128, ...)."""

    def __init__(self, model_name_or_path: Optional[str] = None, device: Optional[str] = None, embedding_dim=None, max_seq_length: int = 128,
                 lower_case: bool = True, **kwargs) -> None:
        kwargs.pop("cache_folder", None), kwargs.pop("onnx_folder", None), kwargs.pop("onnx_model_name", None), kwargs.pop("enable_overwrite", None)
        super().__init__(model_name_or_path, device=device, embedding_dim=embedding_dim, max_seq_length=max_seq_length, **kwargs)
        self.model_name_or_path = model_name_or_path

    def load(self) -> None:
        super().load()
        self.always_normalized = False

    def encode(self, sentences: Union[str, List[str]], normalize: bool = True, **kwargs):
        kw = {"return_device": True} if kwargs.get("return_device") else {}
        out = self._embed(sentences, normalize, **kw)
        return out if kw else torch.from_numpy(self._convert_output(out))   # (the reference returns a CPU FloatTensor)
