"""Plugin base types, mirroring aphrodite/quantization/base_config.py:9-131 and
aphrodite/modeling/layers/linear.py:91-122 (same method names and signatures so
the concrete methods below drop into the reference unchanged)."""
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional

import torch
from torch import nn


class QuantizeMethodBase(ABC):
    @abstractmethod
    def create_weights(self, layer: nn.Module, *weight_args, **extra_weight_attrs):
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def embedding(self, layer: nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        return


class LinearMethodBase(QuantizeMethodBase):
    @abstractmethod
    def create_weights(self, layer: nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int,
                       output_size: int, params_dtype: torch.dtype,
                       **extra_weight_attrs):
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError


class QuantizationConfig(ABC):
    @abstractmethod
    def get_name(self) -> str:
        raise NotImplementedError

    @abstractmethod
    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        raise NotImplementedError

    @classmethod
    @abstractmethod
    def get_min_capability(cls) -> int:
        raise NotImplementedError

    @staticmethod
    @abstractmethod
    def get_config_filenames() -> List[str]:
        raise NotImplementedError

    @classmethod
    @abstractmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig":
        raise NotImplementedError

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        return None

    @staticmethod
    def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
        for key in keys:
            if key in config:
                return config[key]
        raise ValueError(f"Cannot find any of {keys} in the model's "
                         "quantization config.")

    @staticmethod
    def get_from_keys_or(config: Dict[str, Any], keys: List[str], default: Any) -> Any:
        try:
            return QuantizationConfig.get_from_keys(config, keys)
        except ValueError:
            return default

    @abstractmethod
    def get_quant_method(self, layer: nn.Module, prefix: str) -> Optional[QuantizeMethodBase]:
        raise NotImplementedError

    @abstractmethod
    def get_scaled_act_names(self) -> List[str]:
        raise NotImplementedError


def set_weight_attrs(weight: torch.Tensor, attrs: Optional[Dict[str, Any]]):
    """aphrodite/modeling/utils.py set_weight_attrs."""
    if attrs is None:
        return
    for k, v in attrs.items():
        assert not hasattr(weight, k), f"Overwriting existing attribute {k}"
        setattr(weight, k, v)


def _param(data: torch.Tensor, **attrs) -> nn.Parameter:
    p = nn.Parameter(data, requires_grad=False)
    set_weight_attrs(p, {k: v for k, v in attrs.items() if v is not None})
    return p
