"""Mixed-precision linear kernel seam: the object a ``*LinearMethod`` delegates weight
post-processing and the GEMM to, selected per layer from ``_POSSIBLE_KERNELS``.  Interface restated
from aphrodite/quantization/kernels/MPLinearKernel.py:11-83 (field and method names are the
reference's -- ``gptq_marlin.py`` / ``compressed_tensors_wNa16.py`` construct these objects)."""
import abc
import dataclasses
from typing import Callable, Optional, Tuple

import torch

from ...scalar_type import ScalarType


@dataclasses.dataclass
class MPLinearLayerConfig:
    """Static description of one quantised linear layer (shapes are [in, out])."""
    full_weight_shape: Tuple[int, int]
    partition_weight_shape: Tuple[int, int]
    weight_type: ScalarType
    act_type: torch.dtype
    group_size: int
    zero_points: bool
    has_g_idx: bool


class MPLinearKernel(abc.ABC):
    # ---- class-level capability questions (asked before an instance exists) ------------------------
    @classmethod
    @abc.abstractmethod
    def get_min_capability(cls) -> int:
        ...

    @classmethod
    @abc.abstractmethod
    def can_implement(cls, c: MPLinearLayerConfig) -> Tuple[bool, Optional[str]]:
        """(True, None) or (False, reason)."""
        ...

    # ---- instance: remembers which layer attributes hold the tensors ------------------------------
    def __init__(self, c: MPLinearLayerConfig, w_q_param_name: str, w_s_param_name: str,
                 w_zp_param_name: Optional[str] = None,
                 w_gidx_param_name: Optional[str] = None) -> None:
        ok, why = self.can_implement(c)
        if not ok:
            raise AssertionError(f"{type(self).__name__} cannot implement this layer: {why}")
        self.config = c
        self.w_q_name, self.w_s_name = w_q_param_name, w_s_param_name
        self.w_zp_name, self.w_gidx_name = w_zp_param_name, w_gidx_param_name

    @abc.abstractmethod
    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        ...

    @abc.abstractmethod
    def apply_weights(self, layer: torch.nn.Module, x: torch.Tensor,
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        ...

    # ---- helpers for subclasses ------------------------------------------------------------------
    def _transform_param(self, layer: torch.nn.Module, name: Optional[str], fn: Callable) -> None:
        """Replace layer.<name> by fn(layer.<name>) as a frozen parameter (no-op if absent)."""
        current = getattr(layer, name, None) if name is not None else None
        if current is None:
            return
        replacement = fn(current)
        delattr(layer, name)
        layer.register_parameter(name, torch.nn.Parameter(replacement.data, requires_grad=False))

    def _get_weight_params(self, layer: torch.nn.Module):
        """(w_q, w_s, w_zp or None, g_idx or None)."""
        opt = lambda n: getattr(layer, n, None) if n else None  # noqa: E731
        return getattr(layer, self.w_q_name), getattr(layer, self.w_s_name), opt(self.w_zp_name), \
            opt(self.w_gidx_name)
