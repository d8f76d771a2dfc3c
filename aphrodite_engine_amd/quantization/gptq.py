"""GPTQ on MI355X -- mirror of aphrodite/quantization/gptq.py (GPTQConfig
:18-83, ExllamaState :85-89, GPTQLinearMethod :92-243) bound to the CDNA4
kernels.  Beyond the reference: bf16 activations are accepted
(get_supported_act_dtypes, cf. gptq.py:54-55 which is fp16-only)."""
import enum
from enum import Enum
from fractions import Fraction
from typing import Any, Dict, List, Optional

import torch
from torch import nn

from .. import _custom_ops as ops
from .base_config import LinearMethodBase, QuantizationConfig, _param
from .utils import layer_kind


class GPTQConfig(QuantizationConfig):
    def __init__(self, weight_bits: int, group_size: int, desc_act: bool,
                 lm_head_quantized: bool = False) -> None:
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.desc_act = desc_act
        self.lm_head_quantized = lm_head_quantized
        self.pack_factor = Fraction(32, self.weight_bits)
        if self.weight_bits not in [2, 3, 4, 8]:
            raise ValueError(
                "Currently, only 2/3/4/8-bit weight quantization is "
                f"supported for GPTQ, but got {self.weight_bits} bits.")

    def __repr__(self) -> str:
        return (f"GPTQConfig(weight_bits={self.weight_bits}, "
                f"group_size={self.group_size}, desc_act={self.desc_act}), "
                f"lm_head_quantized={self.lm_head_quantized}")

    @classmethod
    def get_name(cls) -> str:
        return "gptq"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 60

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "GPTQConfig":
        weight_bits = cls.get_from_keys(config, ["bits"])
        group_size = cls.get_from_keys(config, ["group_size"])
        desc_act = cls.get_from_keys(config, ["desc_act"])
        lm_head_quantized = cls.get_from_keys_or(config, ["lm_head"], default=False)
        return cls(weight_bits, group_size, desc_act, lm_head_quantized)

    def get_quant_method(self, layer: nn.Module, prefix: str):
        kind = layer_kind(layer)
        if kind == "moe":                 # int4 experts: grouped CDNA4 GEMM (moe.py)
            if self.weight_bits != 4:
                raise ValueError(f"GPTQ experts are served at 4 bits only (got {self.weight_bits})")
            from ..moe import Wna16MoEMethod
            return Wna16MoEMethod("gptq", self.group_size, self.desc_act)
        if kind == "linear" or (kind == "embedding" and self.lm_head_quantized
                                and "ParallelLMHead" in {c.__name__ for c in type(layer).__mro__}):
            return GPTQLinearMethod(self)
        return None                       # attention, embeddings: nothing to quantise (gptq.py:75-80)

    def get_scaled_act_names(self) -> List[str]:
        return []


class ExllamaState(Enum):
    UNUSED = enum.auto()
    UNINITIALIZED = enum.auto()
    READY = enum.auto()


class CDNA4GPTQLinearMethod(LinearMethodBase):
    def __init__(self, quant_config: GPTQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer: nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int,
                       output_size: int, params_dtype: torch.dtype,
                       **extra_weight_attrs):
        del output_size
        weight_loader = extra_weight_attrs.get("weight_loader")
        cfg = self.quant_config
        if input_size_per_partition % cfg.group_size != 0:
            raise ValueError("The input size is not aligned with the quantized "
                             "weight shape. This can be caused by too large "
                             "tensor parallel size.")
        output_size_per_partition = sum(output_partition_sizes)
        if output_size_per_partition % cfg.pack_factor.numerator != 0:
            raise ValueError("The output size is not aligned with the quantized "
                             "weight shape. This can be caused by too large "
                             "tensor parallel size.")
        group_size = cfg.group_size if cfg.group_size != -1 else input_size
        exllama_state = ExllamaState.UNINITIALIZED
        scale_and_zero_size = input_size // group_size
        scale_and_zero_input_dim = None
        if input_size != input_size_per_partition and cfg.group_size != -1:
            if cfg.desc_act:  # act-order + row parallel: gptq.py:137-139
                exllama_state = ExllamaState.UNUSED
            else:
                scale_and_zero_size = input_size_per_partition // group_size
                scale_and_zero_input_dim = 0
        # 3-bit: 32 values per 3 words -- a Fraction, as in the reference (gptq.py:37); `n // pf` / `n % pf` stay exact
        pf = int(cfg.pack_factor) if cfg.pack_factor.denominator == 1 else cfg.pack_factor
        layer.register_parameter("qweight", _param(
            torch.empty(input_size_per_partition // pf, output_size_per_partition,
                        dtype=torch.int32),
            input_dim=0, output_dim=1, packed_dim=0, pack_factor=pf,
            weight_loader=weight_loader))
        layer.register_parameter("g_idx", _param(
            torch.tensor([i // cfg.group_size for i in range(input_size_per_partition)],
                         dtype=torch.int32),
            input_dim=0, weight_loader=weight_loader))
        layer.register_parameter("qzeros", _param(
            torch.empty(scale_and_zero_size, output_size_per_partition // pf,
                        dtype=torch.int32),
            input_dim=scale_and_zero_input_dim, output_dim=1, packed_dim=1,
            pack_factor=pf, weight_loader=weight_loader))
        layer.register_parameter("scales", _param(
            torch.empty(scale_and_zero_size, output_size_per_partition,
                        dtype=params_dtype),
            input_dim=scale_and_zero_input_dim, output_dim=1,
            weight_loader=weight_loader))
        layer.exllama_state = exllama_state

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        layer.qweight = nn.Parameter(layer.qweight.data, requires_grad=False)
        layer.qzeros = nn.Parameter(layer.qzeros.data, requires_grad=False)
        layer.scales = nn.Parameter(layer.scales.data, requires_grad=False)
        layer.g_idx = nn.Parameter(layer.g_idx.data, requires_grad=False)
        if layer.exllama_state == ExllamaState.UNINITIALIZED:
            if self.quant_config.desc_act:
                layer.g_idx.data = torch.argsort(layer.g_idx).to(torch.int)
            else:
                layer.g_idx.data = torch.empty((0, ), dtype=torch.int,
                                               device=layer.g_idx.device)
            layer.exllama_state = ExllamaState.READY
            ops.gptq_shuffle(layer.qweight, layer.g_idx, self.quant_config.weight_bits)
            # big matrices also get the strip-major copy the one-launch decode GEMM reads fastest (ops.wna16_decode_strip_copy)
            layer.qweight_strip = None if (self.quant_config.desc_act or self.quant_config.weight_bits != 4) \
                else ops.wna16_decode_strip_copy(layer.qweight.data, layer.scales.data)

    def apply(self, layer: nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        out_shape = x.shape[:-1] + (layer.qweight.shape[-1], )
        reshaped_x = x.reshape(-1, x.shape[-1])
        output = ops.wna16_decode_linear(reshaped_x, layer.qweight, layer.qzeros, layer.scales, 1,
                                         getattr(layer, "qweight_strip", None))
        if output is not None:
            if bias is not None:
                output.add_(bias)
            return output.reshape(out_shape)
        output = ops.gptq_gemm(reshaped_x, layer.qweight, layer.qzeros,
                               layer.scales, layer.g_idx,
                               layer.exllama_state == ExllamaState.READY,
                               self.quant_config.weight_bits)
        if bias is not None:
            output.add_(bias)
        return output.reshape(out_shape)


# The reference picks its parameter-object loader (``weight_loader_v2``: ``param.load_qkv_weight(...)`` on
# BaseAphroditeParameter subclasses) for methods whose CLASS NAME is in WEIGHT_LOADER_V2_SUPPORTED
# (modeling/layers/linear.py:28-44, 330-332).  Our parameters are plain nn.Parameters carrying the
# v1 metadata (input_dim / output_dim / packed_dim / pack_factor), which the reference's v1
# ``weight_loader`` methods consume -- so the class carries a name of its own and the reference's
# name stays available as an alias for imports.
GPTQLinearMethod = CDNA4GPTQLinearMethod
