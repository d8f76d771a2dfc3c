"""Quantization methods served by the MI355X hot path (mirror of
aphrodite/quantization/__init__.py:28-62 restricted to SURVEY 2.1's in-scope
set)."""
from typing import Dict, Type

from .awq import AWQConfig
from .base_config import QuantizationConfig
from .compressed_tensors import CompressedTensorsConfig
from .fp8 import Fp8Config
from .gptq import GPTQConfig

QUANTIZATION_METHODS: Dict[str, Type[QuantizationConfig]] = {
    "awq": AWQConfig,
    "gptq": GPTQConfig,
    "fp8": Fp8Config,
    "compressed-tensors": CompressedTensorsConfig,   # W8A8-FP8, W8A16-FP8 and pack-quantized int4 schemes
}


def get_quantization_config(quantization: str) -> Type[QuantizationConfig]:
    if quantization not in QUANTIZATION_METHODS:
        raise ValueError(f"Invalid quantization method: {quantization}")
    return QUANTIZATION_METHODS[quantization]


def register_with_reference(methods: dict) -> None:
    """Overwrite the reference's QUANTIZATION_METHODS entries
    (aphrodite/quantization/__init__.py:28-62) with the MI355X classes;
    idempotent; call from an ``aphrodite.general_plugins`` entry point."""
    methods.update(QUANTIZATION_METHODS)
