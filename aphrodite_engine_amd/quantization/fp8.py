"""FP8 linear on MI355X -- mirror of aphrodite/quantization/fp8.py (Fp8Config
:31-93, Fp8LinearMethod :96-270) and quantization/utils/w8a8_utils.py
(apply_fp8_linear :83-183, requantize_with_max_scale :54-80).

gfx950 implements OCP e4m3fn natively, so -- unlike the reference's MI300
branch (fp8.py:225-234, w8a8_utils.py:207-228) -- checkpoints are used as
stored: no e4m3fnuz re-labelling, no scale doubling, and the fused
cutlass_scaled_mm role is available (per-token x per-channel included)."""
from typing import Any, Dict, List, Optional

import torch
from torch import nn

from .. import _custom_ops as ops
from .base_config import LinearMethodBase, QuantizationConfig, _param
from .utils import layer_is_ignored, layer_kind, unquantized_linear_method

ACTIVATION_SCHEMES = ["static", "dynamic"]


def apply_fp8_linear(input: torch.Tensor, weight: torch.Tensor,
                     weight_scale: torch.Tensor,
                     input_scale: Optional[torch.Tensor] = None,
                     input_scale_ub: Optional[torch.Tensor] = None,
                     bias: Optional[torch.Tensor] = None,
                     cutlass_fp8_supported: bool = True,
                     use_per_token_if_dynamic: bool = False) -> torch.Tensor:
    """w8a8_utils.py:83-183, fused branch (:99-113)."""
    x2 = input.reshape(-1, input.shape[-1])
    qinput, x_scale = ops.scaled_fp8_quant(
        x2, input_scale, scale_ub=input_scale_ub,
        use_per_token_if_dynamic=use_per_token_if_dynamic)
    out = ops.cutlass_scaled_mm(qinput, weight, out_dtype=input.dtype,
                                scale_a=x_scale, scale_b=weight_scale, bias=bias)
    return out.reshape(input.shape[:-1] + (weight.shape[1], ))


def requantize_with_max_scale(weight: torch.Tensor, weight_scale: torch.Tensor,
                              logical_widths: List[int]):
    """w8a8_utils.py:54-80: fused shards with per-shard scales -> one scale."""
    max_w_scale = weight_scale.max()
    unfused = (weight_scale[-1] > torch.finfo(torch.float8_e4m3fn).min)
    if unfused:
        start = 0
        for idx, width in enumerate(logical_widths):
            end = start + width
            # per_tensor_dequantize (:23-28): through fp16 -- a 0-dim fp32 scale does not promote an fp16 tensor --
            # then the static scaled_fp8_quant arithmetic, x * (1 / scale) (fp8/common.cu:187-199)
            w_dq = weight[start:end, :].to(torch.float16) * weight_scale[idx]
            q = (w_dq.float() * (1.0 / max_w_scale.float())).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
            weight[start:end, :] = q
            start = end
    return max_w_scale, weight


class Fp8Config(QuantizationConfig):
    def __init__(self, is_checkpoint_fp8_serialized: bool = False,
                 activation_scheme: str = "dynamic",
                 ignored_layers: Optional[List[str]] = None,
                 weight_only: bool = False) -> None:
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized
        if activation_scheme not in ACTIVATION_SCHEMES:
            raise ValueError(f"Unsupported activation scheme {activation_scheme}")
        self.activation_scheme = activation_scheme
        self.ignored_layers = ignored_layers or []
        self.weight_only = weight_only  # W8A16: the fp8_marlin role (fp8.py:125-127)

    @classmethod
    def get_name(cls) -> str:
        return "fp8"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.bfloat16, torch.half]

    @classmethod
    def get_min_capability(cls) -> int:
        return 80

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "Fp8Config":
        quant_method = cls.get_from_keys(config, ["quant_method"])
        is_checkpoint_fp8_serialized = ("fp8" in quant_method)
        activation_scheme = cls.get_from_keys(config, ["activation_scheme"])
        ignored_layers = cls.get_from_keys_or(config, ["ignored_layers"], None)
        return cls(is_checkpoint_fp8_serialized, activation_scheme, ignored_layers)

    def get_quant_method(self, layer: nn.Module, prefix: str) -> Optional["Fp8LinearMethod"]:
        kind = layer_kind(layer)        # fp8.py:79-92
        if kind == "linear":
            if layer_is_ignored(prefix, self.ignored_layers):   # is_layer_skipped
                return unquantized_linear_method()
            return Fp8LinearMethod(self)
        if kind == "attention":           # k_scale / v_scale of an FP8 KV cache come with the checkpoint
            from .kv_cache import make_kv_cache_method
            return make_kv_cache_method(self)
        if kind == "moe":                 # fp8.py:86-87
            from ..moe import Fp8MoEMethod
            return Fp8MoEMethod(self)
        return None

    def get_scaled_act_names(self) -> List[str]:
        return []


class CDNA4Fp8LinearMethod(LinearMethodBase):
    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config
        self.use_marlin = quant_config.weight_only

    def create_weights(self, layer: nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int,
                       output_size: int, params_dtype: torch.dtype,
                       **extra_weight_attrs):
        del input_size, output_size
        output_size_per_partition = sum(output_partition_sizes)
        weight_loader = extra_weight_attrs.get("weight_loader")
        layer.logical_widths = output_partition_sizes
        layer.input_size_per_partition = input_size_per_partition
        layer.output_size_per_partition = output_size_per_partition
        layer.orig_dtype = params_dtype
        weight_dtype = (torch.float8_e4m3fn
                        if self.quant_config.is_checkpoint_fp8_serialized else params_dtype)
        layer.register_parameter("weight", _param(
            torch.empty(output_size_per_partition, input_size_per_partition,
                        dtype=weight_dtype),
            input_dim=1, output_dim=0, weight_loader=weight_loader))
        if self.quant_config.is_checkpoint_fp8_serialized:
            # one scale per logical matrix of a fused layer (fp8.py:150-170: PerTensorScaleParameter)
            scale = _param(torch.empty(len(output_partition_sizes), dtype=torch.float32),
                           needs_scalar_to_array=True, weight_loader=weight_loader)
            scale[:] = torch.finfo(torch.float32).min
            layer.register_parameter("weight_scale", scale)
            if self.quant_config.activation_scheme == "static":
                iscale = _param(torch.empty(len(output_partition_sizes),
                                            dtype=torch.float32),
                                needs_scalar_to_array=True, weight_loader=weight_loader)
                iscale[:] = torch.finfo(torch.float32).min
                layer.register_parameter("input_scale", iscale)
            else:
                layer.register_parameter("input_scale", None)

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        if not self.quant_config.is_checkpoint_fp8_serialized:
            # fp8.py:185-199: quantise a 16-bit checkpoint per tensor
            qweight, weight_scale = ops.scaled_fp8_quant(
                layer.weight.data.reshape(-1, layer.weight.shape[-1]), scale=None)
            layer.weight = nn.Parameter(qweight.t(), requires_grad=False)
            layer.weight_scale = nn.Parameter(weight_scale, requires_grad=False)
            layer.input_scale = None
            return
        weight, weight_scale = layer.weight.data, layer.weight_scale.data
        if weight_scale.numel() > 1 and weight_scale.dim() == 1 and \
                weight_scale.numel() == len(layer.logical_widths):
            weight_scale, weight = requantize_with_max_scale(
                weight, weight_scale, layer.logical_widths)
        layer.weight = nn.Parameter(weight.t(), requires_grad=False)
        layer.weight_scale = nn.Parameter(weight_scale.reshape(-1).float(),
                                          requires_grad=False)
        if self.quant_config.activation_scheme == "static":
            layer.input_scale = nn.Parameter(layer.input_scale.max().reshape(1),
                                             requires_grad=False)

    def apply(self, layer: nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.use_marlin:
            x2 = x.reshape(-1, x.shape[-1])
            w = layer.weight.t()  # back to the [N,K] row-major checkpoint layout
            out = ops.fp8_marlin_gemm(x2, w, layer.weight_scale, None, 8,
                                      x2.shape[0], w.shape[0], w.shape[1], bias)
            return out.reshape(x.shape[:-1] + (w.shape[0], ))
        return apply_fp8_linear(input=x, weight=layer.weight,
                                weight_scale=layer.weight_scale,
                                input_scale=layer.input_scale, bias=bias,
                                cutlass_fp8_supported=True,
                                use_per_token_if_dynamic=False)


Fp8LinearMethod = CDNA4Fp8LinearMethod   # the reference's name (see the note in gptq.py on weight_loader_v2)


# --------------------------------------------------------------------------------------------------
# llm-compressor / compressed-tensors "float-quantized" W8A8 checkpoints (BASELINE configs[2]):
# per-channel (or per-tensor) weight scales, dynamic PER-TOKEN activation scales.  Scheme restated
# from quantization/compressed_tensors/schemes/compressed_tensors_w8a8_fp8.py:19-148 as a
# LinearMethodBase (the reference wraps schemes in CompressedTensorsLinearMethod; the tensors,
# their names and the forward are the scheme's).  On gfx950 the checkpoint's OCP e4m3fn bytes are
# used as stored (no fnuz normalisation, :38-43, :52-60).
# --------------------------------------------------------------------------------------------------
class CompressedTensorsW8A8Fp8Config(QuantizationConfig):
    def __init__(self, strategy: str = "channel", is_static_input_scheme: bool = False) -> None:
        if strategy not in ("channel", "tensor"):
            raise ValueError(f"Unknown quantization strategy {strategy}")
        self.strategy = strategy
        self.is_static_input_scheme = is_static_input_scheme

    def get_name(self) -> str:
        return "compressed-tensors"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 89

    @staticmethod
    def get_config_filenames() -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "CompressedTensorsW8A8Fp8Config":
        groups = config.get("config_groups", {})
        first = next(iter(groups.values()), {})
        w = first.get("weights", {}) or {}
        a = first.get("input_activations", {}) or {}
        return cls(strategy=w.get("strategy", "channel"), is_static_input_scheme=not a.get("dynamic", True))

    def get_quant_method(self, layer: nn.Module, prefix: str) -> Optional["CompressedTensorsW8A8Fp8Method"]:
        return CompressedTensorsW8A8Fp8Method(self) if layer_kind(layer) == "linear" else None

    def get_scaled_act_names(self) -> List[str]:
        return []


class CompressedTensorsW8A8Fp8Method(LinearMethodBase):
    def __init__(self, quant_config: CompressedTensorsW8A8Fp8Config):
        self.quant_config = quant_config

    def create_weights(self, layer: nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        del input_size, output_size, params_dtype
        n = sum(output_partition_sizes)
        loader = extra_weight_attrs.get("weight_loader")
        layer.logical_widths = output_partition_sizes
        layer.register_parameter("weight", _param(
            torch.empty(n, input_size_per_partition, dtype=torch.float8_e4m3fn),
            input_dim=1, output_dim=0, weight_loader=loader))
        if self.quant_config.strategy == "channel":
            scale = _param(torch.empty((n, 1), dtype=torch.float32), output_dim=0, weight_loader=loader)
        else:
            scale = _param(torch.empty(len(output_partition_sizes), dtype=torch.float32),
                           needs_scalar_to_array=True, weight_loader=loader)
        scale[:] = torch.finfo(torch.float32).min
        layer.register_parameter("weight_scale", scale)
        if self.quant_config.is_static_input_scheme:
            iscale = _param(torch.empty(len(output_partition_sizes), dtype=torch.float32),
                            needs_scalar_to_array=True, weight_loader=loader)
            iscale[:] = torch.finfo(torch.float32).min
            layer.register_parameter("input_scale", iscale)
        else:
            layer.register_parameter("input_scale", None)

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        if self.quant_config.strategy == "tensor":
            scale, weight = requantize_with_max_scale(layer.weight.data, layer.weight_scale.data,
                                                      layer.logical_widths)
            scale = scale.reshape(1).float()
        else:
            weight, scale = layer.weight.data, layer.weight_scale.data.reshape(-1).float()
        layer.weight = nn.Parameter(weight.t(), requires_grad=False)
        layer.weight_scale = nn.Parameter(scale.contiguous(), requires_grad=False)
        if self.quant_config.is_static_input_scheme:
            layer.input_scale = nn.Parameter(layer.input_scale.max().reshape(1), requires_grad=False)
        else:
            layer.input_scale = None

    def apply(self, layer: nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        return apply_fp8_linear(input=x, weight=layer.weight, weight_scale=layer.weight_scale,
                                input_scale=layer.input_scale, bias=bias, cutlass_fp8_supported=True,
                                use_per_token_if_dynamic=True)

