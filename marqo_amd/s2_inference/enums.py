"""Enums of the vectorise() path (reference: s2_inference/multimodal_model_load.py:35-39,
s2_inference/models/model_type.py, tensor_search/enums.py AvailableModelsKey, api/configs.py EnvVars)."""
from enum import Enum


class Modality(str, Enum):
    TEXT = "language"
    IMAGE = "image"
    VIDEO = "video"
    AUDIO = "audio"


class ModelType(str, Enum):
    OpenCLIP = "open_clip"
    CLIP = "clip"
    SBERT = "sbert"
    Test = "test"
    SBERT_ONNX = "sbert_onnx"
    CLIP_ONNX = "clip_onnx"
    MultilingualClip = "multilingual_clip"
    FP16_CLIP = "fp16_clip"
    Random = "random"
    HF_MODEL = "hf"
    HF_STELLA = "hf_stella"
    NO_MODEL = "no_model"
    LanguageBind = "languagebind"


class AvailableModelsKey:
    model = "model"
    most_recently_used_time = "most_recently_used_time"
    model_size = "model_size"


class EnvVars:
    MARQO_MAX_CPU_MODEL_MEMORY = "MARQO_MAX_CPU_MODEL_MEMORY"
    MARQO_MAX_CUDA_MODEL_MEMORY = "MARQO_MAX_CUDA_MODEL_MEMORY"
    MARQO_MAX_VECTORISE_BATCH_SIZE = "MARQO_MAX_VECTORISE_BATCH_SIZE"
    MARQO_INFERENCE_CACHE_SIZE = "MARQO_INFERENCE_CACHE_SIZE"
    MARQO_INFERENCE_CACHE_TYPE = "MARQO_INFERENCE_CACHE_TYPE"
    MARQO_BEST_AVAILABLE_DEVICE = "MARQO_BEST_AVAILABLE_DEVICE"
    # engine-specific (no reference equivalent)
    MARQO_AMD_MODEL_DIR = "MARQO_AMD_MODEL_DIR"                  # where checkpoints / vocab files are looked up
    MARQO_AMD_SYNTHETIC_WEIGHTS = "MARQO_AMD_SYNTHETIC_WEIGHTS"  # "1": random-init weights when a checkpoint is absent
    MARQO_AMD_MICRO_BATCH_ROWS = "MARQO_AMD_MICRO_BATCH_ROWS"    # token rows per device micro-batch
    MARQO_AMD_MAX_ITEMS_PER_ENCODE = "MARQO_AMD_MAX_ITEMS_PER_ENCODE"  # non-text items (images) staged per engine encode() call
