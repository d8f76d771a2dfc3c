"""compressed-tensors (llm-compressor) checkpoints on MI355X -- the config of
aphrodite/quantization/compressed_tensors/compressed_tensors.py:29-330 reduced to the schemes the
hot path serves, each bound to a CDNA4 kernel:

  format            weights                    activations             method here
  float-quantized   fp8, tensor / channel      fp8 dynamic per token   CompressedTensorsW8A8Fp8Method (fp8.py)
                                               or static per tensor
  float-quantized   fp8, tensor / channel      none                    CompressedTensorsW8A16Fp8Method
  pack-quantized    int4 / int8 symmetric,     none                    CompressedTensorsWNA16Method ->
                    group / channel (+actorder)                        MPLinearKernel (kernels/cdna4.py)
  pack-quantized    int4 symmetric experts     none                    CompressedTensorsMoEMethod (moe.py) ->
                    (FusedMoE), group / channel                        the grouped int4 GEMM

The reference wraps "schemes" in one CompressedTensorsLinearMethod; here every scheme IS a
LinearMethodBase (same tensors, names and forward), and scheme selection is a table of predicates
over the parsed ``weights`` / ``input_activations`` blocks instead of an if-chain.  Anything else
(int8 W8A8, 2:4 sparse) raises NotImplementedError like the reference does for unknown
combinations (:252-253)."""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch
from torch import nn

from .. import _custom_ops as ops
from ..scalar_type import scalar_types
from .base_config import LinearMethodBase, QuantizationConfig, _param
from .fp8 import CompressedTensorsW8A8Fp8Config, CompressedTensorsW8A8Fp8Method
from .kernels import choose_mp_linear_kernel
from .utils import (FUSED_LAYER_SHARDS, layer_is_ignored, layer_kind, name_matches as _matches,
                    unquantized_linear_method)
from .kernels.MPLinearKernel import MPLinearLayerConfig

ACTIVATION_QUANT_FORMATS = ("naive-quantized", "int-quantized", "float-quantized")


@dataclass(frozen=True)
class QuantArgs:
    """One ``weights`` / ``input_activations`` block (compressed_tensors/utils.py QuantizationArgs)."""
    num_bits: int = 8
    type: str = "int"               # "int" | "float"
    symmetric: bool = True
    strategy: Optional[str] = None  # tensor | channel | group | block | token
    group_size: Optional[int] = None
    dynamic: bool = False
    actorder: Optional[str] = None  # None | "group" | "weight"

    @classmethod
    def parse(cls, d: Optional[Dict[str, Any]]) -> Optional["QuantArgs"]:
        if not d:
            return None
        act = d.get("actorder")
        if isinstance(act, bool):
            act = "group" if act else None
        return cls(num_bits=int(d.get("num_bits", 8)), type=str(d.get("type", "int")).lower(),
                   symmetric=bool(d.get("symmetric", True)), strategy=d.get("strategy"),
                   group_size=d.get("group_size"), dynamic=bool(d.get("dynamic", False)), actorder=act)


class CompressedTensorsConfig(QuantizationConfig):
    def __init__(self, target_scheme_map: Dict[str, Dict[str, Optional[QuantArgs]]], ignore: Optional[List[str]],
                 quant_format: Optional[str], kv_cache_scheme: Optional[Dict[str, Any]] = None) -> None:
        self.target_scheme_map = target_scheme_map
        self.ignore = ignore or []
        self.quant_format = quant_format
        self.kv_cache_scheme = kv_cache_scheme

    def get_name(self) -> str:
        return "compressed_tensors"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 70

    @staticmethod
    def get_config_filenames() -> List[str]:
        return []

    def get_scaled_act_names(self) -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "CompressedTensorsConfig":
        table: Dict[str, Dict[str, Optional[QuantArgs]]] = {}
        for group in config.get("config_groups", {}).values():
            entry = {"weights": QuantArgs.parse(group.get("weights")),
                     "input_activations": QuantArgs.parse(group.get("input_activations"))}
            for target in group.get("targets", []):
                table[target] = entry
        return cls(table, config.get("ignore"), config.get("format"), config.get("kv_cache_scheme"))

    # -- scheme selection -----------------------------------------------------------------------------
    def _scheme_for(self, prefix: str) -> Dict[str, Optional[QuantArgs]]:
        """Target match: the layer's name (exact / ``re:``), else the module class ``Linear``
        (compressed_tensors/utils.py find_matched_target)."""
        proj = prefix.split(".")[-1]
        names = [prefix.replace(proj, s) for s in FUSED_LAYER_SHARDS[proj]] if proj in FUSED_LAYER_SHARDS \
            else [prefix]
        found = []
        for n in names:
            hit = next((t for t in self.target_scheme_map if _matches(n, t)), None)
            if hit is None:
                hit = next((t for t in self.target_scheme_map if t in ("Linear", "re:.*")), None)
            if hit is None:
                raise ValueError(f"Unable to find matching target for {n} in the compressed-tensors config.")
            found.append(self.target_scheme_map[hit])
        if any(f is not found[0] and f != found[0] for f in found):
            raise ValueError(f"Found a different quantization schemes for the shards of {prefix}.")
        return found[0]

    def get_quant_method(self, layer: nn.Module, prefix: str) -> Optional[LinearMethodBase]:
        kind = layer_kind(layer)             # compressed_tensors.py:60-79
        if kind == "attention":
            from .kv_cache import make_kv_cache_method
            return make_kv_cache_method(self)
        if kind == "moe":                    # compressed_tensors.py:77-78
            from ..moe import CompressedTensorsMoEMethod
            return CompressedTensorsMoEMethod(self)
        if kind != "linear":
            return None
        if layer_is_ignored(prefix, self.ignore):
            return unquantized_linear_method()   # the layer keeps its 16-bit weight
        scheme = self._scheme_for(prefix)
        w, a = scheme["weights"], scheme["input_activations"]
        if w is None:
            raise NotImplementedError("No compressed-tensors compatible scheme was found.")
        static_w = not w.dynamic and w.symmetric
        if (a is None and static_w and w.type == "int" and w.strategy in ("channel", "group")
                and self.quant_format == "pack-quantized"):
            if w.num_bits not in (4, 8):           # WNA16_SUPPORTED_BITS (compressed_tensors_wNa16.py:16-20)
                raise ValueError(f"Unsupported num_bits = {w.num_bits}. Supported num_bits = [4, 8]")
            return CompressedTensorsWNA16Method(w.num_bits, w.strategy, w.group_size, w.actorder)
        if self.quant_format in ACTIVATION_QUANT_FORMATS and w.type == "float" and w.num_bits == 8 and static_w \
                and w.strategy in ("tensor", "channel"):
            if a is None:
                return CompressedTensorsW8A16Fp8Method(w.strategy)
            if a.type == "float" and a.num_bits == 8 and (a.dynamic or (a.symmetric and a.strategy == "tensor")):
                return CompressedTensorsW8A8Fp8Method(
                    CompressedTensorsW8A8Fp8Config(w.strategy, is_static_input_scheme=not a.dynamic))
        raise NotImplementedError("No compressed-tensors compatible scheme was found.")


# --------------------------------------------------------------------------------------------------
class CompressedTensorsWNA16Method(LinearMethodBase):
    """``pack-quantized`` int4 (compressed_tensors/schemes/compressed_tensors_wNa16.py:25-176):
    ``weight_packed`` int32 [N, K/8] packed along K, ``weight_scale`` [N, K/g], ``weight_shape`` [2],
    optional ``weight_g_idx`` [K]; symmetric (stored value = q + 8).  The GEMM and the repack belong to
    the MPLinearKernel chosen for the layer."""

    def __init__(self, num_bits: int, strategy: str, group_size: Optional[int] = None,
                 actorder: Optional[str] = None) -> None:
        self.pack_factor = 32 // num_bits
        self.strategy = strategy
        self.group_size = -1 if group_size is None else group_size
        self.has_g_idx = actorder == "group"
        if self.group_size == -1 and strategy != "channel":
            raise ValueError("Marlin kernels require group quantization or channelwise quantization, but found no "
                             "group size and strategy is not channelwise.")
        self.quant_type = {4: scalar_types.uint4b8, 8: scalar_types.uint8b128}[num_bits]    # (wNa16.py:16-20)

    def create_weights(self, layer: nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        loader = extra_weight_attrs.get("weight_loader")
        n = sum(output_partition_sizes)
        cfg = MPLinearLayerConfig(full_weight_shape=(input_size, output_size),
                                  partition_weight_shape=(input_size_per_partition, n),
                                  weight_type=self.quant_type, act_type=params_dtype, group_size=self.group_size,
                                  zero_points=False, has_g_idx=self.has_g_idx)
        kernel_type = choose_mp_linear_kernel(cfg)
        group = self.group_size if self.group_size != -1 else input_size
        row_parallel = input_size != input_size_per_partition
        # scales follow the K split unless act-order needs every group on every rank
        # (marlin_repeat_scales_on_all_ranks, quantization/utils/marlin_utils.py)
        repeat = self.has_g_idx or (self.group_size == -1 and row_parallel)
        groups = input_size // group if repeat else input_size_per_partition // group
        layer.register_parameter("weight_packed", _param(
            torch.empty(n, input_size_per_partition // self.pack_factor, dtype=torch.int32),
            input_dim=1, output_dim=0, packed_dim=1, pack_factor=self.pack_factor, weight_loader=loader))
        layer.register_parameter("weight_scale", _param(
            torch.empty(n, groups, dtype=params_dtype),
            input_dim=None if repeat else 1, output_dim=0, weight_loader=loader))
        layer.register_parameter("weight_shape", _param(torch.empty(2, dtype=torch.int64), weight_loader=loader))
        if self.has_g_idx:
            layer.register_parameter("weight_g_idx", _param(
                torch.empty(input_size_per_partition, dtype=torch.int32), input_dim=0, weight_loader=loader))
        layer.kernel = kernel_type(cfg, w_q_param_name="weight_packed", w_s_param_name="weight_scale",
                                   w_zp_param_name=None, w_gidx_param_name="weight_g_idx")

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        layer.kernel.process_weights_after_loading(layer)

    def apply(self, layer: nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        return layer.kernel.apply_weights(layer, x, bias)


class CompressedTensorsW8A16Fp8Method(LinearMethodBase):
    """FP8 weights, 16-bit activations (compressed_tensors_w8a16_fp8.py): the ``fp8_marlin_gemm`` role,
    weights used as stored ([N, K] e4m3, per-tensor or per-channel scale)."""

    def __init__(self, strategy: str) -> None:
        if strategy not in ("tensor", "channel"):
            raise ValueError(f"Unknown quantization strategy {strategy}")
        self.strategy = strategy

    def create_weights(self, layer: nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        del input_size, output_size, params_dtype
        loader = extra_weight_attrs.get("weight_loader")
        n = sum(output_partition_sizes)
        layer.logical_widths = output_partition_sizes
        layer.register_parameter("weight", _param(
            torch.empty(n, input_size_per_partition, dtype=torch.float8_e4m3fn),
            input_dim=1, output_dim=0, weight_loader=loader))
        if self.strategy == "channel":
            scale = _param(torch.empty((n, 1), dtype=torch.float32), output_dim=0, weight_loader=loader)
        else:
            scale = _param(torch.empty(len(output_partition_sizes), dtype=torch.float32),
                           needs_scalar_to_array=True, weight_loader=loader)
        scale[:] = torch.finfo(torch.float32).min
        layer.register_parameter("weight_scale", scale)

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        if self.strategy == "tensor":   # one scale per logical matrix -> per-channel, no requantisation
            widths = torch.tensor(layer.logical_widths, device=layer.weight_scale.device)
            scale = torch.repeat_interleave(layer.weight_scale.data.reshape(-1), widths)
        else:
            scale = layer.weight_scale.data.reshape(-1)
        layer.weight = nn.Parameter(layer.weight.data, requires_grad=False)
        layer.weight_scale = nn.Parameter(scale.float().contiguous(), requires_grad=False)

    def apply(self, layer: nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        x2 = x.reshape(-1, x.shape[-1])
        w = layer.weight
        out = ops.fp8_marlin_gemm(x2, w, layer.weight_scale, None, 8, x2.shape[0], w.shape[0], w.shape[1], bias)
        return out.reshape(x.shape[:-1] + (w.shape[0], ))
