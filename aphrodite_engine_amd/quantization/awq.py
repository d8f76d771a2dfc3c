"""AWQ on MI355X -- mirror of aphrodite/quantization/awq.py (AWQConfig :19-78,
AWQLinearMethod :81-169).  ``apply`` keeps the reference's op calls
(awq_dequantize + matmul for >= 256 tokens, awq_gemm below).  With
``prepack=True`` (the awq_marlin role, quantization/awq_marlin.py) weights are
transposed once at load time into the CDNA4 K-packed layout and the fast
W4A16 kernel is used directly."""
import os
from typing import Any, Dict, List, Optional

import torch
from torch import nn

from ..switches import switch
from .. import _custom_ops as ops
from .base_config import LinearMethodBase, QuantizationConfig, _param
from .utils import layer_kind


class AWQConfig(QuantizationConfig):
    def __init__(self, weight_bits: int, group_size: int, zero_point: bool,
                 prepack: bool = False) -> None:
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.zero_point = zero_point
        self.prepack = prepack
        if self.weight_bits != 4:
            raise ValueError("Currently, only 4-bit weight quantization is "
                             f"supported for AWQ, but got {self.weight_bits} bits.")
        self.pack_factor = 32 // self.weight_bits

    def __repr__(self) -> str:
        return (f"AWQConfig(weight_bits={self.weight_bits}, "
                f"group_size={self.group_size}, zero_point={self.zero_point})")

    def get_name(self) -> str:
        return "awq"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 75

    @staticmethod
    def get_config_filenames() -> List[str]:
        return ["quant_config.json", "quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "AWQConfig":
        weight_bits = cls.get_from_keys(config, ["w_bit", "bits"])
        group_size = cls.get_from_keys(config, ["q_group_size", "group_size"])
        zero_point = cls.get_from_keys(config, ["zero_point"])
        # checkpoints are re-laid into the CDNA4 K-packed order at load time (the decode fast path
        # needs it); APHRODITE_AWQ_NO_PREPACK=1 keeps the on-disk layout and the awq_gemm op
        return cls(weight_bits, group_size, zero_point,
                   prepack=(switch("APHRODITE_AWQ_NO_PREPACK") or "0") != "1")

    def get_quant_method(self, layer: nn.Module, prefix: str):
        kind = layer_kind(layer)
        if kind == "moe":                 # int4 experts: grouped CDNA4 GEMM (moe.py)
            from ..moe import Wna16MoEMethod
            return Wna16MoEMethod("awq", self.group_size)
        if kind == "linear":
            return AWQLinearMethod(self)
        return None                       # awq.py:63-67

    def get_scaled_act_names(self) -> List[str]:
        return ["gelu", "gelu_fast", "gelu_new", "gelu_pytorch_tanh"]


class CDNA4AWQLinearMethod(LinearMethodBase):
    def __init__(self, quant_config: AWQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer: nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int,
                       output_size: int, params_dtype: torch.dtype,
                       **extra_weight_attrs):
        cfg = self.quant_config
        if input_size_per_partition % cfg.group_size != 0:
            raise ValueError("The input size is not aligned with the quantized "
                             "weight shape. This can be caused by too large "
                             "tensor parallel size.")
        output_size_per_partition = sum(output_partition_sizes)
        if output_size_per_partition % cfg.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized "
                             "weight shape. This can be caused by too large "
                             "tensor parallel size.")
        weight_loader = extra_weight_attrs.get("weight_loader")
        layer.register_parameter("qweight", _param(
            torch.empty(input_size_per_partition,
                        output_size_per_partition // cfg.pack_factor, dtype=torch.int32),
            input_dim=0, output_dim=1, packed_dim=1, pack_factor=cfg.pack_factor,
            weight_loader=weight_loader))
        layer.register_parameter("qzeros", _param(
            torch.empty(input_size_per_partition // cfg.group_size,
                        output_size_per_partition // cfg.pack_factor, dtype=torch.int32),
            input_dim=0, output_dim=1, packed_dim=1, pack_factor=cfg.pack_factor,
            weight_loader=weight_loader))
        layer.register_parameter("scales", _param(
            torch.empty(input_size_per_partition // cfg.group_size,
                        output_size_per_partition, dtype=params_dtype),
            input_dim=0, output_dim=1, weight_loader=weight_loader))
        layer.awq_prepacked = False

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        layer.qweight = nn.Parameter(layer.qweight.data, requires_grad=False)
        layer.qzeros = nn.Parameter(layer.qzeros.data, requires_grad=False)
        layer.scales = nn.Parameter(layer.scales.data, requires_grad=False)
        if self.quant_config.prepack:
            k, n = layer.qweight.shape[0], layer.qweight.shape[1] * 8
            layer.qweight = nn.Parameter(
                ops.awq_marlin_repack(layer.qweight.data, k, n, 4), requires_grad=False)
            layer.qzeros = nn.Parameter(
                ops.awq_repack_zeros(layer.qzeros.data, n), requires_grad=False)
            layer.awq_prepacked = True
            layer.qweight_strip = ops.wna16_decode_strip_copy(layer.qweight.data, layer.scales.data)

    def apply(self, layer: nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        qweight, scales, qzeros = layer.qweight, layer.scales, layer.qzeros
        pack_factor = self.quant_config.pack_factor
        reshaped_x = x.reshape(-1, x.shape[-1])
        if getattr(layer, "awq_prepacked", False):
            out_shape = x.shape[:-1] + (qweight.shape[-1], )
            out = ops.wna16_decode_linear(reshaped_x, qweight, qzeros, scales, 0, getattr(layer, "qweight_strip", None))
            if out is None:
                out = ops.wna16_gemm(reshaped_x, qweight, qzeros, scales, None, 0)
        else:
            out_shape = x.shape[:-1] + (qweight.shape[-1] * pack_factor, )
            # num_tokens >= threshold (awq.py:159-163)
            if x.shape[:-1].numel() >= 256:
                # (the reference's own rule for the unprepacked AWQ layout; prepack=True runs the hand-written kernels)
                ops._library_fallback("awq linear", "unprepacked AWQ layout at >= 256 tokens: awq_dequantize + matmul (awq.py:159-163)")
                out = ops.awq_dequantize(qweight, scales, qzeros, 0, 0, 0)
                out = torch.matmul(reshaped_x, out)
            else:
                out = ops.awq_gemm(reshaped_x, qweight, scales, qzeros, pack_factor)
        if bias is not None:
            out.add_(bias)
        return out.reshape(out_shape)


# The reference picks its parameter-object loader (``weight_loader_v2``: ``param.load_qkv_weight(...)`` on
# BaseAphroditeParameter subclasses) for methods whose CLASS NAME is in WEIGHT_LOADER_V2_SUPPORTED
# (modeling/layers/linear.py:28-44, 330-332).  Our parameters are plain nn.Parameters carrying the
# v1 metadata (input_dim / output_dim / packed_dim / pack_factor), which the reference's v1
# ``weight_loader`` methods consume -- so the class carries a name of its own and the reference's
# name stays available as an alias for imports.
AWQLinearMethod = CDNA4AWQLinearMethod
