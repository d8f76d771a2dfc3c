"""The three plugin base types a quantisation method has to satisfy to be picked up by the
reference's model loader: ``QuantizeMethodBase`` / ``LinearMethodBase`` (per-layer behaviour) and
``QuantizationConfig`` (per-checkpoint description).  Contract restated from
aphrodite/quantization/base_config.py:9-131 and aphrodite/modeling/layers/linear.py:91-122 --
method names, argument order and defaults are the reference's so the concrete GPTQ / AWQ / FP8
methods in this package can be registered in ``QUANTIZATION_METHODS`` unchanged; the bodies are
ours (a small ``_abstract`` helper instead of one ``raise`` per method)."""
import abc
import functools
from typing import Any, Dict, List, Optional

import torch
from torch import nn


def _abstract(owner: Any, what: str):
    raise NotImplementedError(f"{type(owner).__name__} must implement {what}")


# --------------------------------------------------------------------------- per-layer methods
class QuantizeMethodBase(abc.ABC):
    """What a layer delegates to: allocate its parameters, post-process them once the checkpoint
    is loaded, run the forward."""

    @abc.abstractmethod
    def create_weights(self, layer: nn.Module, *weight_args, **extra_weight_attrs):
        _abstract(self, "create_weights")

    @abc.abstractmethod
    def apply(self, layer: nn.Module, *args, **kwargs) -> torch.Tensor:
        _abstract(self, "apply")

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        """Optional hook (repack / requantise); the default keeps the checkpoint layout."""

    def __init_subclass__(cls, **kwargs):
        """Every subclass's ``process_weights_after_loading`` runs ONCE per layer: the hook repacks / transposes in place,
        and a layer can meet it twice -- from the reference's loader pass over the modules (model_loader/loader.py:402-408)
        and from this package's own model code (reference_model.py, LlamaForCausalLM.process_weights_after_loading)."""
        super().__init_subclass__(**kwargs)
        fn = cls.__dict__.get("process_weights_after_loading")
        if fn is None or getattr(fn, "_runs_once", False):
            return

        @functools.wraps(fn)
        def once(self, layer, _fn=fn):
            done = layer.__dict__.setdefault("_pwal_done", set())
            if _fn.__qualname__ in done:
                return None
            out = _fn(self, layer)
            done.add(_fn.__qualname__)
            return out
        once._runs_once = True
        cls.process_weights_after_loading = once

    def embedding(self, layer: nn.Module, *args, **kwargs) -> torch.Tensor:
        """Only embedding-capable methods override this."""
        _abstract(self, "embedding")


class LinearMethodBase(QuantizeMethodBase):
    """A QuantizeMethodBase for (row / column parallel) linear layers."""

    @abc.abstractmethod
    def create_weights(self, layer: nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        _abstract(self, "create_weights")

    @abc.abstractmethod
    def apply(self, layer: nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        _abstract(self, "apply")


# --------------------------------------------------------------------------- per-checkpoint config
class QuantizationConfig(abc.ABC):
    """Describes one quantised checkpoint format and hands out the per-layer method."""

    # -- identity / capability gates checked by the loader (model_loader/loader.py:90-112) --------
    @abc.abstractmethod
    def get_name(self) -> str:
        _abstract(self, "get_name")

    @abc.abstractmethod
    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        _abstract(self, "get_supported_act_dtypes")

    @classmethod
    @abc.abstractmethod
    def get_min_capability(cls) -> int:
        """major * 10 + minor of the oldest device served (gfx950 reports 95)."""
        raise NotImplementedError

    # -- construction from the checkpoint's quantisation json ---------------------------------------
    @staticmethod
    @abc.abstractmethod
    def get_config_filenames() -> List[str]:
        raise NotImplementedError

    @classmethod
    @abc.abstractmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig":
        raise NotImplementedError

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        """Formats that can take over another method's checkpoints return their name here."""
        return None

    @staticmethod
    def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
        """First of `keys` present in the json (formats disagree on spelling)."""
        hit = next((k for k in keys if k in config), None)
        if hit is None:
            raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")
        return config[hit]

    @staticmethod
    def get_from_keys_or(config: Dict[str, Any], keys: List[str], default: Any) -> Any:
        hit = next((k for k in keys if k in config), None)
        return default if hit is None else config[hit]

    # -- per-layer dispatch --------------------------------------------------------------------------
    @abc.abstractmethod
    def get_quant_method(self, layer: nn.Module, prefix: str) -> Optional[QuantizeMethodBase]:
        _abstract(self, "get_quant_method")

    @abc.abstractmethod
    def get_scaled_act_names(self) -> List[str]:
        _abstract(self, "get_scaled_act_names")


# --------------------------------------------------------------------------- parameter helpers
def set_weight_attrs(weight: torch.Tensor, attrs: Optional[Dict[str, Any]]):
    """Attach loader metadata (input_dim, output_dim, packed_dim, ...) to a parameter; refuses to
    overwrite (aphrodite/modeling/utils.py set_weight_attrs)."""
    for key, value in (attrs or {}).items():
        if hasattr(weight, key):
            raise AssertionError(f"Overwriting existing attribute {key}")
        setattr(weight, key, value)


def _param(data: torch.Tensor, **attrs) -> nn.Parameter:
    p = nn.Parameter(data, requires_grad=False)
    set_weight_attrs(p, {k: v for k, v in attrs.items() if v is not None})
    return p
