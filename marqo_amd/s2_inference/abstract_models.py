"""Base classes of the loader interface (reference: core/inference/embedding_models/abstract_embedding_model.py:7-53,
abstract_clip_model.py:19-113)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import numpy as np
from PIL import UnidentifiedImageError

from marqo_amd.s2_inference.image_input import _is_image


class AbstractEmbeddingModel(ABC):
    # engine models take a whole request and micro-batch it on the device themselves
    supports_dynamic_batching = False

    def __init__(self, model_properties: Optional[dict] = None, device: Optional[str] = None, model_auth=None):
        if device is None:
            raise ValueError("`device` is required for loading CLIP models!")
        self.device = device
        self.model_auth = model_auth

    def load(self):
        self._load_necessary_components()
        self._check_loaded_components()

    @abstractmethod
    def _load_necessary_components(self):
        pass

    @abstractmethod
    def _check_loaded_components(self):
        pass

    @abstractmethod
    def encode(self):
        pass


class AbstractCLIPModel(AbstractEmbeddingModel):
    def __init__(self, device: Optional[str] = None, model_properties: Optional[dict] = None, model_auth=None):
        super().__init__(model_properties, device, model_auth)
        self.model = None
        self.tokenizer = None
        self.preprocess = None

    @abstractmethod
    def encode_text(self, inputs, normalize: bool = True) -> np.ndarray:
        pass

    @abstractmethod
    def encode_image(self, inputs, normalize: bool = True, image_download_headers: dict = None) -> np.ndarray:
        pass

    @staticmethod
    def normalize(outputs):
        """abstract_clip_model.py:83-85: the row norms a caller divides by (the towers normalise on the device themselves)"""
        return outputs.norm(dim=-1, keepdim=True)

    def encode(self, inputs, default: str = "text", normalize=True, **kwargs) -> np.ndarray:
        """image-vs-text dispatch (abstract_clip_model.py:56-75): `infer` + first-element sniffing, else `default`.
        `modality` and any other kwarg the callers pass are tolerated and ignored."""
        infer = kwargs.pop("infer", True)
        if infer and _is_image(inputs):
            is_image = True
        elif default == "text":
            is_image = False
        elif default == "image":
            is_image = True
        else:
            raise UnidentifiedImageError(f"expected default='image' or default='text' but received {default}")
        extra = {"return_device": True} if kwargs.get("return_device") else {}   # engine extension: keep the result in HBM
        if is_image:
            return self.encode_image(inputs, normalize=normalize,
                                     image_download_headers=kwargs.get("image_download_headers", dict()), **extra)
        return self.encode_text(inputs, normalize=normalize, **extra)
