"""The reference's own plumbing fakes: `random` (returns seeded random vectors, src/marqo/s2_inference/random_utils.py:30-64)
and `no_model` (src/marqo/s2_inference/no_model_utils.py).  Host-only numpy; they exist so the vectorise() control flow can be
tested without any tower — they never stand in for the HIP path."""
from __future__ import annotations

import functools
import hashlib

import numpy as np
from PIL.Image import Image as ImageType

from marqo_amd.s2_inference.errors import VectoriseError


class Model:
    """legacy loader base (sbert_utils.Model): ctor (name, device=, embedding_dim=, max_seq_length=, **kwargs)"""

    def __init__(self, model_name: str = None, device: str = None, batch_size: int = 2048, embedding_dim=None,
                 max_seq_length=None, **kwargs) -> None:
        self.model_name = model_name
        self.device = device
        self.model = None
        self.embedding_dimension = embedding_dim
        self.max_seq_length = max_seq_length
        self.batch_size = batch_size

    def load(self) -> None:
        raise NotImplementedError

    def encode(self, *args, **kwargs):
        raise NotImplementedError


def sentence_to_hash(sentence) -> int:
    if isinstance(sentence, ImageType):
        pixel_data = list(sentence.getdata())
        if isinstance(pixel_data[0], int):
            image_average = functools.reduce(lambda x, y: x + y, pixel_data) / len(pixel_data)
        else:
            pixel_averages = [sum(ch) / len(ch) for ch in pixel_data]
            image_average = functools.reduce(lambda x, y: x + y, pixel_averages) / len(pixel_data)
        return int(hashlib.sha256(str(image_average).encode("utf-8")).hexdigest(), 16) % 10 ** 8
    return int(hashlib.sha256(sentence.encode("utf-8")).hexdigest(), 16) % 10 ** 8


class Random(Model):
    def load(self) -> None:
        self.model = None

    def _get_sentences_hash(self, sentences) -> int:
        hashes, i = 0, 0
        for i, s in enumerate(sentences):
            hashes += sentence_to_hash(s)
        return hashes // (i + 1)

    def encode(self, sentence, normalize: bool = True, **kwargs) -> np.ndarray:
        if self.embedding_dimension is None or self.embedding_dimension == 0:
            raise ValueError("invalid embedding dimension size. check the model registry is correct")
        if isinstance(sentence, str):
            np.random.seed(sentence_to_hash(sentence))
            return np.random.rand(1, self.embedding_dimension)
        if len(sentence) == 0:
            raise ValueError("recevied empty sentence")
        np.random.seed(self._get_sentences_hash(sentence))
        return np.random.rand(len(sentence), self.embedding_dimension)


class NO_MODEL(Model):
    def load(self, *args, **kwargs) -> None:
        pass

    def encode(self, *args, **kwargs) -> None:
        raise VectoriseError("Cannot vectorise anything with 'no_model'. This model is intended for adding documents and "
                             "searching with custom vectors only. If vectorisation is needed, please use a different model ")
