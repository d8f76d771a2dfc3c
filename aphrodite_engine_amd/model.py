"""Llama decoder wired exactly like the reference's hot loop
(aphrodite/modeling/models/llama.py: LlamaMLP :58-103, LlamaAttention :106-190,
LlamaDecoderLayer :193-270, LlamaModel :273-357; call stack SURVEY 3.2), with
every layer op going through the MI355X ``_custom_ops`` / quant-method /
attention-backend surface.  Used by bench.py, smoke() and the GPU tests; weights
are synthetic but in the real on-disk formats (GPTQ v1 int4 g128, FP8 e4m3
per-tensor/per-channel), created directly on the device.

Tensor parallelism follows the reference: QKV / gate_up column
parallel, o_proj / down_proj row parallel + all-reduce
(modeling/layers/linear.py:1139-1143).
"""
import os
import math
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import nn

from .switches import switch
from . import _custom_ops as ops
from .attention.backend import MI355XAttentionImpl, MI355XAttentionMetadata
from .moe import DeferredCombine
from .distributed import (DeferredAllReduce, defer_all_reduce,
                          get_tensor_model_parallel_rank,
                          get_tensor_model_parallel_world_size,
                          tensor_model_parallel_all_reduce)
from .quantization.awq import AWQConfig
from .quantization.base_config import QuantizationConfig
from .quantization.fp8 import CompressedTensorsW8A8Fp8Config, Fp8Config
from .quantization.gptq import GPTQConfig


@dataclass
class LlamaConfig:
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    max_position_embeddings: int = 8192
    rope_scaling: Optional[dict] = None   # {"rope_type": "llama3", factor, low_freq_factor, high_freq_factor, original_max_position_embeddings}
    num_local_experts: int = 0        # > 0: Mixtral-style sparse MLP (block_sparse_moe), SURVEY 8f row 2
    num_experts_per_tok: int = 2
    attention_bias: bool = False      # q / k / v / o projection biases (models/llama.py:109, 135-150: config.attention_bias or config.bias)
    mlp_bias: bool = False            # gate / up / down projection biases (models/llama.py:62-82: config.mlp_bias)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def padded_vocab_size(vocab_size: int, tp: int = 1, pad_to: int = 64) -> int:
    """pad_vocab_size (modeling/layers/vocab_parallel_embedding.py): next multiple of 64, made divisible by tp."""
    v = (vocab_size + pad_to - 1) // pad_to * pad_to
    if v % tp:
        step = pad_to * tp
        v = (vocab_size + step - 1) // step * step
    return v


LLAMA3_8B = LlamaConfig()
LLAMA3_70B = LlamaConfig(hidden_size=8192, intermediate_size=28672,
                         num_hidden_layers=80, num_attention_heads=64,
                         num_key_value_heads=8)
MIXTRAL_8X7B = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                          num_attention_heads=32, num_key_value_heads=8, vocab_size=32000,
                          rope_theta=1e6, max_position_embeddings=32768, num_local_experts=8,
                          num_experts_per_tok=2)
TINY_MOE = LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                       num_attention_heads=4, num_key_value_heads=2, vocab_size=1024,
                       max_position_embeddings=2048, num_local_experts=8, num_experts_per_tok=2)
TINY = LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                   num_attention_heads=4, num_key_value_heads=2, vocab_size=1024,
                   max_position_embeddings=2048)


class QuantLinear(nn.Module):
    """LinearBase (modeling/layers/linear.py:150-190) reduced to what the hot
    path needs: holds the quant method and its parameters."""

    def __init__(self, in_features: int, out_partition_sizes: List[int],
                 quant_config: Optional[QuantizationConfig], dtype: torch.dtype,
                 full_in_features: Optional[int] = None, prefix: str = "", plan=None,
                 full_out_features: Optional[int] = None, bias: bool = False):
        super().__init__()
        from .loader import make_weight_loader, merged_plan
        self.in_features = in_features
        self.out_features = sum(out_partition_sizes)
        self.quant_config = quant_config
        # how checkpoint tensors are cut into this rank's parameters (loader.py)
        self.plan = plan if plan is not None else merged_plan(
            [n * get_tensor_model_parallel_world_size() for n in out_partition_sizes])
        self.weight_loader = make_weight_loader(self.plan)
        self.quant_method = None if quant_config is None else quant_config.get_quant_method(self, prefix)
        if self.quant_method is None:   # unquantised checkpoint, or a layer the config ignores
            from .quantization.base_config import _param
            self.register_parameter("weight", _param(
                torch.empty(self.out_features, in_features, dtype=dtype),
                input_dim=1, output_dim=0, weight_loader=self.weight_loader))
        else:
            self.quant_method.create_weights(
                self, in_features, out_partition_sizes,
                full_in_features or in_features, full_out_features or self.out_features, dtype,
                weight_loader=self.weight_loader)
        if bias:
            # ColumnParallelLinear / RowParallelLinear.bias (linear.py:283-291, 1074-1084): cut along the output like the
            # weight's rows in a column-parallel layer, whole in a row-parallel one (added after the all-reduce)
            from .quantization.base_config import _param
            self.register_parameter("bias", _param(torch.zeros(self.out_features, dtype=dtype), output_dim=0,
                                                   weight_loader=self.weight_loader))
        else:
            self.bias = None

    def forward(self, x: torch.Tensor, add_bias: bool = True) -> torch.Tensor:
        """``add_bias=False``: a row-parallel layer under TP -- the caller adds ``self.bias`` once, after the all-reduce
        (RowParallelLinear.forward, linear.py:1136-1150)."""
        bias = self.bias if add_bias else None
        if self.quant_method is None:
            return torch.nn.functional.linear(x, self.weight, bias)
        if getattr(self, "qweight_strip_major", False):
            # DecoderLayer.enable_one_copy: ``qweight`` holds the strip-major order of the decode kernels, the only resident
            # copy -- any M through the kernels that read it (ops.wna16_linear_strip), never through the quant method's op
            fp = self.fast_params()
            out = ops.wna16_linear_strip(x.reshape(-1, x.shape[-1]), fp[0], fp[1], fp[2], fp[3])
            if bias is not None:
                out.add_(bias)
            return out.reshape(x.shape[:-1] + (self.out_features,))
        return self.quant_method.apply(self, x, bias)

    def fast_params(self):
        """(qweight, qzeros, scales, zero_offset) if this layer's weights are in the
        CDNA4 K-packed layout served by the packed-activation W4A16 kernel."""
        from .quantization.gptq import ExllamaState
        if isinstance(self.quant_config, GPTQConfig):
            if getattr(self, "exllama_state", None) == ExllamaState.READY and self.g_idx.numel() == 0 \
                    and self.quant_config.weight_bits == 4:
                return self.qweight, self.qzeros, self.scales, 1
        elif isinstance(self.quant_config, AWQConfig) and getattr(self, "awq_prepacked", False):
            return self.qweight, self.qzeros, self.scales, 0
        kernel = getattr(self, "kernel", None)   # compressed-tensors pack-quantized via the MPLinearKernel seam
        if kernel is not None and type(kernel).__name__ == "CDNA4LinearKernel" \
                and getattr(self, "_cdna4_perm", None) is None and hasattr(self, "_cdna4_zp") \
                and getattr(self, "_cdna4_bits", 4) == 4:
            return self.weight_packed, self._cdna4_zp, self.weight_scale, 0
        return None


def _yarn_inv_freq(head_dim: int, theta: float, factor: float, orig_max: int, beta_fast, beta_slow,
                   extrapolation_factor: float, device):
    """YaRN frequencies (rotary_embedding.py:332-364, 400-417): interpolated (1 / (factor * theta^(2i/d))) and extrapolated
    (1 / theta^(2i/d)) frequencies blended by a linear ramp over the rotary dimensions between the two correction bounds
    -- the dimensions that turn ``beta_fast`` / ``beta_slow`` times over the original context."""
    pos_freqs = theta ** (torch.arange(0, head_dim, 2, dtype=torch.float, device=device) / head_dim)
    extra = 1.0 / pos_freqs
    inter = 1.0 / (factor * pos_freqs)

    def correction_dim(rotations):
        return (head_dim * math.log(orig_max / (rotations * 2 * math.pi))) / (2 * math.log(theta))
    low = max(math.floor(correction_dim(beta_fast)), 0)
    high = min(math.ceil(correction_dim(beta_slow)), head_dim - 1)
    if low == high:
        high += 0.001
    ramp = torch.clamp((torch.arange(head_dim // 2, dtype=torch.float, device=device) - low) / (high - low), 0, 1)
    mask = (1 - ramp) * extrapolation_factor
    return inter * (1 - mask) + extra * mask


def _rope_cache(head_dim: int, max_pos: int, theta: float, dtype, device, rope_scaling: Optional[dict] = None):
    """cos | sin table [rows, head_dim] (modeling/layers/rotary_embedding.py:101-120) -- what the rotary kernels index by
    position.  ``rope_scaling`` selects the reference's scaled tables (get_rope, :902-1017):
      "llama3"  (Llama-3.1; :680-723) long wavelengths stretched: a frequency whose wavelength exceeds orig_max /
                low_freq_factor is divided by ``factor``, one below orig_max / high_freq_factor is kept, the band in between
                blended linearly in orig_max / wavelength; max_pos rows;
      "linear"  (:205-287) positions divided by ``factor``; max_pos * factor rows;
      "dynamic" (NTK; :291-329) theta grown by (factor * L / max_pos - (factor - 1))^(d / (d - 2)), L = max_pos * factor rows;
      "yarn"    (:372-430) _yarn_inv_freq, cos / sin scaled by (0.1 ln(factor) + 1) * attn_factor;
                original_max_position_embeddings * factor rows."""
    def plain_inv(base):
        return 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float, device=device) / head_dim))
    inv_freq = plain_inv(theta)
    rows, t_div, mscale = max_pos, None, None
    if rope_scaling:
        kind = rope_scaling.get("rope_type", rope_scaling.get("type"))
        if kind == "llama3":
            factor = float(rope_scaling["factor"])
            lo, hi = float(rope_scaling["low_freq_factor"]), float(rope_scaling["high_freq_factor"])
            orig = float(rope_scaling["original_max_position_embeddings"])
            wavelen = 2 * math.pi / inv_freq
            blend = (orig / wavelen - lo) / (hi - lo) if lo != hi else torch.zeros_like(inv_freq)
            mid = (1 - blend) * inv_freq / factor + blend * inv_freq
            inv_freq = torch.where(wavelen < orig / hi, inv_freq, torch.where(wavelen > orig / lo, inv_freq / factor, mid))
        elif kind == "linear":
            factor = rope_scaling["factor"]
            rows, t_div = max_pos * factor, factor
        elif kind == "dynamic":
            factor = rope_scaling["factor"]
            rows = max_pos * factor
            inv_freq = plain_inv(theta * ((factor * rows / max_pos) - (factor - 1)) ** (head_dim / (head_dim - 2)))
        elif kind == "yarn":
            factor = rope_scaling["factor"]
            orig = rope_scaling["original_max_position_embeddings"]
            inv_freq = _yarn_inv_freq(head_dim, theta, factor, orig, rope_scaling.get("beta_fast", 32),
                                      rope_scaling.get("beta_slow", 1), rope_scaling.get("extrapolation_factor", 1), device)
            rows = orig * factor
            mscale = float((0.1 * math.log(factor) + 1.0 if factor > 1 else 1.0) * rope_scaling.get("attn_factor", 1))
        else:
            # deepseek_yarn / longrope / mrope belong to model families outside the Llama / Mixtral decoder of this path
            raise NotImplementedError(f"rope_scaling type {kind!r} is not implemented (llama3, linear, dynamic, yarn)")
    t = torch.arange(rows, dtype=torch.float, device=device)
    if t_div is not None:
        t = t / t_div
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    cos, sin = freqs.cos(), freqs.sin()
    if mscale is not None:
        cos, sin = cos * mscale, sin * mscale
    return torch.cat((cos, sin), dim=-1).to(dtype)


class LlamaDecoderLayer(nn.Module):
    def __init__(self, cfg: LlamaConfig, quant_config, dtype, kv_cache_dtype: str, layer_idx: int = 0):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.cfg = cfg
        self.num_heads = cfg.num_attention_heads // tp
        self.num_kv_heads = max(1, cfg.num_key_value_heads // tp)
        self.head_dim = cfg.head_dim
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        h = cfg.hidden_size
        self.input_layernorm = nn.Parameter(torch.ones(h, dtype=dtype), requires_grad=False)
        self.post_attention_layernorm = nn.Parameter(torch.ones(h, dtype=dtype),
                                                     requires_grad=False)
        from .loader import merged_plan, qkv_plan, row_plan
        pfx = f"model.layers.{layer_idx}."
        self.qkv_proj = QuantLinear(h, [self.q_size, self.kv_size, self.kv_size],
                                    quant_config, dtype, prefix=pfx + "self_attn.qkv_proj",
                                    plan=qkv_plan(cfg.num_attention_heads, cfg.num_key_value_heads, self.head_dim),
                                    full_out_features=(cfg.num_attention_heads + 2 * cfg.num_key_value_heads)
                                    * self.head_dim, bias=cfg.attention_bias)
        self.o_proj = QuantLinear(self.q_size, [h], quant_config, dtype,
                                  full_in_features=cfg.num_attention_heads * self.head_dim,
                                  prefix=pfx + "self_attn.o_proj", plan=row_plan(), bias=cfg.attention_bias)
        # projection biases: served by the op-by-op path (the quant methods' apply(layer, x, bias)); the fused steps hand raw
        # split-K slabs from GEMM to consumer and have no place for them -- they stay off for such a layer
        self.has_bias = bool(cfg.attention_bias or (cfg.mlp_bias and cfg.num_local_experts == 0))
        inter = cfg.intermediate_size // tp
        self.is_moe = cfg.num_local_experts > 0
        if self.is_moe:
            # MixtralMoE (modeling/models/mixtral.py:60-110): replicated fp16 router + quantised experts
            from .moe import FusedMoE
            self.moe_gate = nn.Parameter(torch.empty(cfg.num_local_experts, h, dtype=dtype), requires_grad=False)
            self.experts = FusedMoE(cfg.num_local_experts, cfg.num_experts_per_tok, h, cfg.intermediate_size,
                                    params_dtype=dtype, reduce_results=True, renormalize=True,
                                    quant_config=quant_config, prefix=pfx + "block_sparse_moe.experts")
            self.gate_up_proj = self.down_proj = None
        else:
            self.gate_up_proj = QuantLinear(h, [inter, inter], quant_config, dtype, prefix=pfx + "mlp.gate_up_proj",
                                            plan=merged_plan([cfg.intermediate_size] * 2),
                                            full_out_features=2 * cfg.intermediate_size, bias=cfg.mlp_bias)
            self.down_proj = QuantLinear(inter, [h], quant_config, dtype,
                                         full_in_features=cfg.intermediate_size, prefix=pfx + "mlp.down_proj",
                                         plan=row_plan(), bias=cfg.mlp_bias)
        self.attn = MI355XAttentionImpl(self.num_heads, self.head_dim,
                                        self.head_dim ** -0.5, self.num_kv_heads,
                                        kv_cache_dtype=kv_cache_dtype)
        self.k_scale = 1.0
        self.v_scale = 1.0
        self.tp = tp
        self.gate_up_interleaved = None
        self.gate_up_strip = None
        self.strip = {}
        self.fp8_strip = {}
        self.fp8_gate_up_il = None
        self.fuse_rope_attention = True
        self.one_copy = False

    def enable_one_copy(self) -> int:
        """Release the [K/8, N] originals of every int4 matrix this layer holds a strip-major decode copy of: ONE resident copy
        per matrix (what the reference keeps: its exllama / Marlin kernels serve every M from the one repacked tensor,
        quantization/gptq.py:214-228; spare HBM is KV blocks, worker/cache_engine.py:66-86).  Call after enable_fused_silu(m <=
        32, keep_original=False).  The linear's ``qweight`` parameter then HOLDS the strip-major words (same shape, flag
        ``qweight_strip_major``) and
          * <= 32 rows: the fused decode step as before (it only ever read the strip-major copies);
          * 33..64 rows: the MLP weights stay on the one-pass kernel, which takes the strip-major addresses
            (ops.wna16_gemm_mid_packed(strip_m=32): same bits); qkv / o run two 32-row halves of the stream kernels (what
            APHRO_DECODE_ROW_HALVES=1 selects) where they have a stream plan, else the layer goes op by op;
          * prompt-sized M: the tile machines / the dequantise-transpose pass address the strip-major pieces in place
            (ops.wna16_gemm_large_strip: same loads, same bits).
        TP 1, dense, bias-free layers only (the TP / sparse steps run the round-2 kernels on [K/8, N]).  Returns the bytes
        released; restore_op_level_layouts undoes it (the permutation backwards) -- call it before exporting the parameters:
        a state_dict taken in between holds the strip-major words."""
        if self.one_copy or self.tp != 1 or self.is_moe or self.has_bias or switch("APHRO_WEIGHTS_TWO_COPIES"):
            return 0
        freed = 0
        for name in ("qkv_proj", "o_proj", "down_proj"):
            lin, st = getattr(self, name), self.strip.get(name)
            if st is None or lin.fast_params() is None or getattr(lin, "qweight_strip_major", False):
                continue
            freed += lin.qweight.numel() * 4
            lin.qweight.data = st
            lin.qweight_strip_major = True
            if getattr(lin, "qweight_strip", None) is not None:
                lin.qweight_strip = None
        lin = self.gate_up_proj
        if self.gate_up_strip is not None and self.gate_up_interleaved is not None and not self.gate_up_keep_original:
            _, qz, sc, zo = self.gate_up_interleaved
            freed += lin.qweight.numel() * 4
            lin.qweight.data = self.gate_up_strip
            lin.qweight_strip_major = True
            self.gate_up_interleaved = (self.gate_up_strip, qz, sc, zo)     # ([0]: shape only -- the words are strip-major)
        self.one_copy = freed > 0
        return freed

    def _undo_one_copy(self) -> None:
        if not self.one_copy:
            return
        for name in ("qkv_proj", "o_proj", "down_proj", "gate_up_proj"):
            lin = getattr(self, name)
            if getattr(lin, "qweight_strip_major", False):
                lin.qweight.data = ops.wna16_strip_unrelayout(lin.qweight.data, 32, lin.scales.shape[0])
                lin.qweight_strip_major = False
        if self.gate_up_interleaved is not None:
            self.gate_up_interleaved = (self.gate_up_proj.qweight.data,) + tuple(self.gate_up_interleaved[1:])
        self.one_copy = False

    def _row_halves(self, m: int) -> bool:
        """33..64 rows on two 32-row halves of the stream kernels: opt-in (measured slower than the one-pass kernel on
        [K/8, N], see enable_fused_silu) -- and the form of a layer that keeps only the strip-major copies."""
        return 32 < m <= 64 and (self.one_copy or switch("APHRO_DECODE_ROW_HALVES") == "1")

    def enable_fused_silu(self, m: int = 32, keep_original: bool = True) -> bool:
        """Re-lay the gate_up weights with interleaved (gate_j, up_j) columns so that the
        decode fast path runs SiluAndMul inside the GEMM epilogue.  With
        keep_original=False the [gate | up] copy is dropped: the parameters themselves take the interleaved
        order and the op-by-op / prompt-sized forward pairs the columns in its SiluAndMul
        (ops.silu_and_mul(..., interleaved=True)) -- one copy of the matrix in HBM."""
        if self.has_bias:                     # (op-by-op path only: no decode copies, no column interleave under the bias)
            return False
        self._undo_one_copy()
        self.enable_resident_layouts(m)       # (also where SiluAndMul cannot ride in the epilogue: TP shards, sparse layers)
        if self.is_moe:
            return False
        fp = self.gate_up_proj.fast_params()
        lin = self.gate_up_proj
        if fp is None or ops.wna16_ksplit(m, lin.out_features, lin.in_features, fp[2].shape[0]) != 1 \
                or lin.out_features % 256 != 0:
            return False
        qw, qz, sc = ops.interleave_gate_up(fp[0], fp[1], fp[2])
        self.gate_up_interleaved = (qw, qz, sc, fp[3])
        self.gate_up_keep_original = keep_original
        # <= 32 rows: the resident-activation kernel (one workgroup per CU, csrc/wna16_gemm_resident.hip) streams a
        # STRIP-MAJOR copy of the interleaved words (every wave's pieces in the order it reads them: 20.7 -> 17.8 us at
        # 4096 x 28672); the [K/8, N] copy stays for the 33..64-row and prefill kernels (2 x 59 MB per layer of 288 GB)
        # (33..64 rows, round 4: the same kernel can run on two 32-row halves, two workgroups per CU -- measured SLOWER than
        #  the 33..64-row kernel on the one-GPU shapes: gate_up 29.6-30.7 us against 26.8, down 17.3 against 14.2 at batch 64,
        #  a CU takes in both halves' activations AND the weights twice; APHRO_DECODE_ROW_HALVES=1 builds the copies anyway)
        self.gate_up_strip = None
        if m <= (64 if switch("APHRO_DECODE_ROW_HALVES") == "1" else 32) and not switch("APHRO_DECODE_NO_RESIDENT") \
                and ops.wna16_resident_ksplit(m, lin.out_features, lin.in_features, sc.shape[0]) == 1:
            self.gate_up_strip = ops.wna16_strip_relayout(qw, m, sc.shape[0])
        # The op-level strip-major copy the quant method made at load time (gptq.py / awq.py process_weights_after_loading,
        # 59 MB per layer on Llama-3-8B) is never read by the fused decode step, and with keep_original=False it would be
        # a copy of the WRONG column order: release it (ADVICE r3 -- gate_up used to live four times in HBM).  The
        # op-by-op forward of this layer then takes the generic op for gate_up, as it does for M > 32.
        if getattr(lin, "qweight_strip", None) is not None:
            lin.qweight_strip = None
        if not keep_original:
            lin.qweight.data, lin.qzeros.data, lin.scales.data = qw, qz, sc
        self.strip.pop("gate_up_proj", None)      # (the [gate | up] copy serves layers WITHOUT the SiluAndMul epilogue only)
        return True

    def restore_op_level_layouts(self) -> None:
        """Undo enable_fused_silu / enable_resident_layouts: the parameters and the op-level strip-major copy as the quant
        method's process_weights_after_loading leaves them (what the reference's own LlamaDecoderLayer runs on through the
        plugin).  bench.py measures its op-by-op leg on this state."""
        self._undo_one_copy()
        self.strip = {}
        self.gate_up_strip = None
        if self.gate_up_interleaved is not None:
            lin = self.gate_up_proj
            if not getattr(self, "gate_up_keep_original", True):
                qw, qz, sc = ops.deinterleave_gate_up(*self.gate_up_interleaved[:3])
                lin.qweight.data, lin.qzeros.data, lin.scales.data = qw, qz, sc
            self.gate_up_interleaved = None
            if hasattr(lin, "qweight_strip") and lin.fast_params() is not None:
                lin.qweight_strip = ops.wna16_decode_strip_copy(lin.qweight.data, lin.scales.data)

    def enable_fp8_strips(self, m: int = 32) -> None:
        """Strip-major copies of the FP8 projections for the resident W8A8 decode GEMM at <= 32 rows
        (csrc/fp8_gemm_resident.hip: one workgroup per CU, a wave's weights as one stream of lane-linear 1 KiB pieces).
        A second copy of every matrix it serves (7 GB on Llama-3-8B: the [N, K] originals keep serving > 32 rows and
        prefill); APHRO_DECODE_NO_FP8_RESIDENT=1 keeps the round-3 kernels."""
        self.fp8_strip = {}
        self.fp8_gate_up_il = None      # round 6: the gate_up strip copy with (gate_j, up_j) adjacent -- SiluAndMul in the epilogue
        if switch("APHRO_DECODE_NO_FP8_RESIDENT") or self.is_moe or self.has_bias:
            return
        for name in ("qkv_proj", "o_proj", "gate_up_proj", "down_proj"):
            lin = getattr(self, name)
            w = getattr(lin, "weight", None)
            if w is None or w.dtype != torch.float8_e4m3fn or w.dim() != 2:
                continue
            wt = w.t()                                   # the [N, K] checkpoint tensor behind the column-major [K, N] view
            if wt.is_contiguous() and ops.fp8_gemm_resident_ksplit(m, wt.shape[0], wt.shape[1]) > 0:
                # (a TP rank under the dynamic scheme keeps the plain copy: its down projection is not on the quantise-on-load form)
                if name == "gate_up_proj" and getattr(lin, "bias", None) is None and not switch("APHRO_FP8_NO_LAUNCH_DIET") \
                        and (self.tp == 1 or getattr(lin, "input_scale", None) is not None) \
                        and ops.fp8_gemm_resident_silu_supported(m, wt.shape[0], wt.shape[1]):
                    self.fp8_gate_up_il = ops.fp8_strip_relayout_interleaved(wt, m)     # instead of, not beside, the plain strip copy
                else:
                    self.fp8_strip[name] = ops.fp8_strip_relayout(wt, m)

    def _fp8_slabs(self, name: str, qx: torch.Tensor) -> torch.Tensor:
        """Raw fp32 split-K slabs of FP8 projection ``name``: the resident kernel on its strip-major copy at <= 32 rows."""
        st = self.fp8_strip.get(name) if qx.shape[0] <= 32 else None
        if st is not None:
            return ops.fp8_gemm_resident(qx, st, slabs=True)
        return ops.scaled_mm_fp8_slabs(qx, getattr(self, name).weight)

    def enable_resident_layouts(self, m: int = 32) -> None:
        """Strip-major copies of the qkv, o and down weights for the resident / stream kernels at <= 32 rows (same K
        partition as the round-2 kernel, so the fp32 slabs -- and everything downstream -- are bit-identical; round 3:
        8.2 -> 7.1 us and 11.4 -> 10.5 us, profiles/r3_resident_bench.txt; round 4, single-pass stream kernel: qkv 6.65,
        down 9.84, o_proj 5.58 -> 4.98 us, profiles/r4_gemm_lab.txt)."""
        self.strip = {}
        if m > 64 or self.has_bias or switch("APHRO_DECODE_NO_RESIDENT"):
            return
        # gate_up_proj: the NON-interleaved [gate | up] matrix, for layers whose SiluAndMul does not ride in the GEMM epilogue
        # (K-sliced gate_up of a TP shard: slabs -> silu_and_mul_pack(slabs=...))
        for name in ("qkv_proj", "o_proj", "down_proj", "gate_up_proj"):
            lin = getattr(self, name, None)
            fp = lin.fast_params() if lin is not None else None
            if fp is None:
                continue
            qw, qz, sc, zo = fp
            n, k, g = lin.out_features, lin.in_features, sc.shape[0]
            rks = ops.wna16_resident_ksplit(m, n, k, g)
            if name == "gate_up_proj":
                if self.tp > 1 and rks > 1:
                    self.strip[name] = ops.wna16_strip_relayout(qw, m, g)
                continue
            # (round 6) no copy the step never reads: under TP the row-parallel projections run the round-2 kernel on
            # [K/8, N] (their output is all-reduced, not handed on as slabs), and so does o_proj in front of a sparse MLP --
            # 14.7 MB per layer of a 70B TP-8 shard (down_proj) that used to sit in HBM unread
            if (name in ("o_proj", "down_proj") and self.tp > 1) or (name == "o_proj" and self.is_moe):
                continue
            # <= 32 rows: only where the resident plan keeps the round-2 kernel's K slices (the consumers were tuned to those
            # slab counts); 33..64 rows (two 32-row halves): opt-in, see enable_fused_silu
            if m > 32 and switch("APHRO_DECODE_ROW_HALVES") != "1":
                continue
            if rks > 0 and (m > 32 or rks == ops.wna16_ksplit(m, n, k, g)) and (k // 8) * n * 4 >= 2 ** 23:
                self.strip[name] = ops.wna16_strip_relayout(qw, m, g)

    def _gemm_slabs(self, name, packed, m, k):
        """fp32 split-K slabs of projection ``name`` on packed activations: the resident kernel on its strip-major copy at
        <= 32 rows, else the round-2 kernel."""
        lin = getattr(self, name)
        qw, qz, sc, zo = lin.fast_params()
        st = self.strip.get(name)
        #  the strip-major copy serves <= 32 rows; 33..64 rows only with the row halves (opt-in, or a layer that
        # keeps one copy) AND a plan the stream kernel is instantiated for (an 8192 x 8192 o_proj plans to {4, 8, 1, 0}: no
        # stream form) -- otherwise the round-2 kernel on the [K/8, N] layout, as before round 4
        if st is not None and m > 32 and not (self._row_halves(m)
                                              and ops.wna16_resident_ksplit(m, lin.out_features, lin.in_features, sc.shape[0]) > 0):
            st = None
        if st is not None:
            return ops.wna16_gemm_resident(packed, m, k, st, qz, sc, zo, mode="slabs", strip_layout=True)
        if getattr(lin, "qweight_strip_major", False):     # (one copy, no stream plan for this M: fused_decode_ok says no --
            qw = ops.wna16_strip_unrelayout(qw, 32, sc.shape[0])    #  kept correct for a direct caller: [K/8, N] rebuilt)
        return ops.wna16_gemm_packed(packed, m, k, qw, qz, sc, zo, partials=True)

    def _packed_weights(self, name: str) -> Optional[torch.Tensor]:
        """The int4 weight tensor the decode GEMM of projection ``name`` streams (the strip-major copy where the layer has
        one, the interleaved gate_up where enabled): what an all-reduce + norm launch in front of it prefetches."""
        if name == "gate_up_proj":
            if self.gate_up_strip is not None:
                return self.gate_up_strip
            if self.gate_up_interleaved is not None:
                return self.gate_up_interleaved[0]
        st = self.strip.get(name)
        if st is not None:
            return st
        fp = getattr(self, name).fast_params()
        return fp[0] if fp is not None else None

    def fused_decode_ok(self, m: int) -> bool:
        """Decode fast path (7 launches per layer instead of 17): W4A16 linears in the
        K-packed layout, shapes (per TP shard) served by the packed-activation kernel.  With
        TP > 1 the two row-parallel projections reduce their split-K slabs locally, all-reduce
        the [M, hidden] result over the TP group and hand it to the fused norm as a tensor."""
        if m > 64 or self.has_bias:
            return False
        for lin in self.linears():
            fp = lin.fast_params()
            if fp is None or ops.wna16_ksplit(m, lin.out_features, lin.in_features, fp[2].shape[0]) <= 0:
                return False
            # one resident copy (enable_one_copy): 33..64 rows need the stream kernel's row-halves plan on that copy
            if m > 32 and getattr(lin, "qweight_strip_major", False) \
                    and ops.wna16_resident_ksplit(m, lin.out_features, lin.in_features, fp[2].shape[0]) <= 0:
                return False
        return True

    def linears(self):
        """The dense quantised projections of this layer (a sparse MLP's experts live in ``experts``)."""
        return (self.qkv_proj, self.o_proj) if self.is_moe else \
            (self.qkv_proj, self.o_proj, self.gate_up_proj, self.down_proj)

    def moe_block(self, normed: torch.Tensor, router_logits: Optional[torch.Tensor] = None,
                  defer_combine: bool = False, defer_all_reduce: bool = False):
        """router (fp16 library GEMM, [M, E], unless the norm kernel already produced the logits) + fused experts
        (+ TP all-reduce inside FusedMoE)."""
        if router_logits is None:
            router_logits = torch.matmul(normed, self.moe_gate.t())
        return self.experts(normed, router_logits, defer_combine=defer_combine,
                            defer_all_reduce=defer_all_reduce and self.tp > 1)

    def forward_decode_fused(self, positions, x, slabs, residual, first, kv_cache, attn_metadata, cos_sin,
                             cos_sin_tok=None, next_weights=None):
        """x: row-major input (first layer, or the all-reduced down_proj output of the previous
        layer when TP > 1) or None; slabs: fp32 split-K slabs of the previous down_proj (TP == 1).
        Returns (x, slabs) of this layer's down_proj in the same convention."""
        eps = self.cfg.rms_norm_eps
        m = positions.shape[0]
        h = self.cfg.hidden_size
        if isinstance(x, DeferredCombine):        # the previous layer's sparse MLP: combine inside this norm launch
            packed, _ = ops.fused_add_rms_norm_pack_combine(x.slabs, x.inv, x.topk_weights, residual, not first,
                                                            self.input_layernorm, eps)
            qkv_slabs, _ = self._gemm_slabs("qkv_proj", packed, m, h)
        elif isinstance(x, DeferredAllReduce):    # TP: the previous layer's last all-reduce runs inside this norm launch
            packed, _ = x.finish(residual, self.input_layernorm, eps, prefetch=self._packed_weights("qkv_proj"))
            qkv_slabs, _ = self._gemm_slabs("qkv_proj", packed, m, h)
        else:
            packed, _ = ops.fused_add_rms_norm_pack(x if slabs is None else None, slabs, residual,
                                                    not first, self.input_layernorm, eps)
            qkv_slabs, _ = self._gemm_slabs("qkv_proj", packed, m, h)
        from .attention.paged_attn import PagedAttention
        key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads, self.head_dim)
        if self.head_dim == 128 and self.fuse_rope_attention:
            # rotary embedding + cache write run inside the attention kernel
            attn_packed, _ = ops.paged_attention_rope_packed(
                qkv_slabs, None if cos_sin_tok is not None else positions,
                cos_sin_tok if cos_sin_tok is not None else cos_sin, attn_metadata.slot_mapping,
                key_cache, value_cache,
                self.num_heads, self.num_kv_heads, self.attn.scale, attn_metadata.block_tables,
                attn_metadata.seq_lens_tensor, value_cache.shape[3], attn_metadata.max_decode_seq_len,
                None, self.attn.kv_cache_dtype, self.k_scale, self.v_scale)
        else:
            q = ops.rope_cache(None, qkv_slabs, None if cos_sin_tok is not None else positions,
                               cos_sin_tok if cos_sin_tok is not None else cos_sin, True, key_cache, value_cache,
                               attn_metadata.slot_mapping, self.num_heads, self.num_kv_heads, self.head_dim,
                               self.attn.kv_cache_dtype, self.k_scale, self.v_scale)
            attn_packed, _ = ops.paged_attention_packed(
                q.view(m, self.num_heads, self.head_dim), key_cache, value_cache, self.num_kv_heads,
                self.attn.scale, attn_metadata.block_tables, attn_metadata.seq_lens_tensor,
                value_cache.shape[3], attn_metadata.max_decode_seq_len, None, self.attn.kv_cache_dtype,
                self.k_scale, self.v_scale)
        qw, qz, sc, zo = self.o_proj.fast_params()
        if self.is_moe:
            # sparse MLP: the norm hands row-major activations to the router and the expert gather
            if self.tp > 1:
                o = ops.wna16_gemm_packed(attn_packed, m, self.q_size, qw, qz, sc, zo, partials=False)
                if self.moe_gate.shape[0] <= 16 and not switch("APHRO_MOE_NO_NORM_ROUTER"):
                    # all-reduce + residual add + norm + the router's logits: one launch of the peer-access kernel where the
                    # communicator serves the shape (csrc/custom_all_reduce.hip, ROUTER form), else all-reduce, then norm + router
                    dar = defer_all_reduce(o)
                    if dar is not None:
                        normed, logits = dar.finish_router(residual, self.post_attention_layernorm, eps, self.moe_gate)
                    else:
                        o = tensor_model_parallel_all_reduce(o)
                        normed, logits = ops.fused_add_rms_norm_router(o, None, residual, True,
                                                                       self.post_attention_layernorm, eps, self.moe_gate)
                    return self.moe_block(normed, logits, defer_all_reduce=True), None
                o = tensor_model_parallel_all_reduce(o)
                _, normed = ops.fused_add_rms_norm_pack(o, None, residual, True, self.post_attention_layernorm,
                                                        eps, pack=False, want_out=True)
            else:
                o_slabs, _ = ops.wna16_gemm_packed(attn_packed, m, self.q_size, qw, qz, sc, zo, partials=True)
                if self.moe_gate.shape[0] <= 16 and not switch("APHRO_MOE_NO_NORM_ROUTER"):
                    # the router's logits come out of the norm launch (no [M, E] library GEMM launch)
                    normed, logits = ops.fused_add_rms_norm_router(None, o_slabs, residual, True,
                                                                   self.post_attention_layernorm, eps, self.moe_gate)
                    return self.moe_block(normed, logits,
                                          defer_combine=not switch("APHRO_MOE_NO_DEFERRED_COMBINE")), None
                _, normed = ops.fused_add_rms_norm_pack(None, o_slabs, residual, True,
                                                        self.post_attention_layernorm, eps, pack=False,
                                                        want_out=True)
            return self.moe_block(normed, defer_all_reduce=True), None
        if self.tp > 1:   # row-parallel: local reduce, all-reduce over the TP group, then the norm
            o = ops.wna16_gemm_packed(attn_packed, m, self.q_size, qw, qz, sc, zo, partials=False)
            # (with enable_all_reduce_overlap: the all-reduce runs on a side stream while this stream pulls the
            #  gate_up weights through the Infinity Cache -- distributed/overlap.py)
            gu = self.gate_up_interleaved if self.gate_up_interleaved is not None else self.gate_up_proj.fast_params()
            # all-reduce + residual add + RMSNorm + pack as ONE launch of the peer-access kernel where it applies
            # (csrc/custom_all_reduce.hip; same bits)
            dar = defer_all_reduce(o)
            if dar is not None:
                packed2, _ = dar.finish(residual, self.post_attention_layernorm, eps,
                                        prefetch=None if self.is_moe else self._packed_weights("gate_up_proj"))
            else:
                o = tensor_model_parallel_all_reduce(o, prefetch=gu[:3])
                packed2, _ = ops.fused_add_rms_norm_pack(o, None, residual, True,
                                                         self.post_attention_layernorm, eps)
        else:
            o_slabs, _ = self._gemm_slabs("o_proj", attn_packed, m, self.q_size)
            packed2, _ = ops.fused_add_rms_norm_pack(None, o_slabs, residual, True,
                                                     self.post_attention_layernorm, eps)
        # 33..64 rows: the MLP weights go through the one-pass 32x32x16 MFMA kernel (wna16_gemm_mid.hip: 26.6 vs 37.6 us on
        # gate_up at 64 rows) -- same packed activations in, same packed activations / fp32 slabs out
        mid = 32 < m <= 64 and not switch("APHRO_DECODE_NO_MID")
        # 33..64 rows: the one-pass 32x32x16 MFMA kernel; APHRO_DECODE_ROW_HALVES=1: the stream kernel on two 32-row halves
        # where the layer has the strip-major copies (measured slower on the one-GPU shapes, see enable_fused_silu)
        halves = self._row_halves(m)
        # one resident copy: the one-pass kernel reads the strip-major words in place (strip_m), so the MLP weights keep it and
        # only an explicit APHRO_DECODE_ROW_HALVES=1 moves them to the stream kernel's halves
        gu_one = self.one_copy and getattr(self.gate_up_proj, "qweight_strip_major", False)
        dn_one = self.one_copy and getattr(self.down_proj, "qweight_strip_major", False)
        asked_halves = halves and switch("APHRO_DECODE_ROW_HALVES") == "1"
        if self.gate_up_interleaved is not None:
            # SiluAndMul + pack run in the GEMM epilogue (interleaved gate/up columns)
            qw, qz, sc, zo = self.gate_up_interleaved
            gu_mid = mid and ops.wna16_gemm_mid_ksplit(m, qw.shape[1], h, sc.shape[0]) == 1 and qw.shape[1] % 256 == 0
            if halves and self.gate_up_strip is not None and ops.wna16_resident_ksplit(m, qw.shape[1], h, sc.shape[0]) == 1 \
                    and not (gu_one and gu_mid and not asked_halves):
                act_packed = ops.wna16_gemm_resident(packed2, m, h, self.gate_up_strip, qz, sc, zo, mode="silu",
                                                     strip_layout=True)
            elif gu_mid:
                act_packed = ops.wna16_gemm_mid_silu_pack(packed2, m, h, qw, qz, sc, zo, strip_m=32 if gu_one else 0)
            elif self.gate_up_strip is not None and m <= 32 and ops.wna16_resident_ksplit(m, qw.shape[1], h, sc.shape[0]) == 1:
                act_packed = ops.wna16_gemm_resident(packed2, m, h, self.gate_up_strip, qz, sc, zo, mode="silu",
                                                     strip_layout=True)
            else:
                if gu_one:                                         # (not reached through fused_decode_ok; kept correct)
                    qw = ops.wna16_strip_unrelayout(qw, 32, sc.shape[0])
                act_packed = ops.wna16_gemm_silu_pack(packed2, m, h, qw, qz, sc, zo)
        else:
            qw, qz, sc, zo = self.gate_up_proj.fast_params()
            if (ops.wna16_ksplit(m, qw.shape[1], h, sc.shape[0]) > 1 or "gate_up_proj" in self.strip) \
                    and not switch("APHRO_DECODE_NO_SILU_SLABS"):
                # K-sliced gate_up (TP shards: 8192 x 7168 at 64 rows): the slab reduce rides in the SiluAndMul + pack launch
                # (GEMM + splitk_reduce + silu_and_mul_pack -> GEMM + one consumer, same bits); the GEMM is the stream kernel
                # on a strip-major copy where the layer has one
                gu_slabs, _ = self._gemm_slabs("gate_up_proj", packed2, m, h)
                act_packed = ops.silu_and_mul_pack(None, slabs=gu_slabs, dtype=sc.dtype)
            else:
                gate_up = ops.wna16_gemm_packed(packed2, m, h, qw, qz, sc, zo, partials=False)
                act_packed = ops.silu_and_mul_pack(gate_up)
        qw, qz, sc, zo = self.down_proj.fast_params()
        if self.tp > 1:
            d = ops.wna16_gemm_packed(act_packed, m, self.down_proj.in_features, qw, qz, sc, zo, partials=False)
            dar = defer_all_reduce(d)      # the next norm launch (next layer / final norm) runs it
            if dar is not None:
                return dar, None
            return tensor_model_parallel_all_reduce(d, prefetch=next_weights), None
        kd = self.down_proj.in_features
        if mid and not ((asked_halves if dn_one else halves) and "down_proj" in self.strip) \
                and qw.shape[1] * kd >= 2 ** 25 \
                and ops.wna16_gemm_mid_ksplit(m, qw.shape[1], kd, sc.shape[0]) > 0:
            down_slabs, _ = ops.wna16_gemm_mid_packed(act_packed, m, kd, qw, qz, sc, zo, partials=True, strip_m=32 if dn_one else 0)
        else:
            down_slabs, _ = self._gemm_slabs("down_proj", act_packed, m, kd)
        return None, down_slabs

    # -- FP8 W8A8 (per-token dynamic or static per-tensor activations) decode fast path ----------
    def fused_decode_fp8_ok(self, m: int) -> bool:
        """9 launches per layer instead of 15+: every activation quantisation rides in the kernel
        that produces the activations, the GEMMs hand their raw fp32 split-K slabs to the consumer
        (which dequantises with the per-token x per-channel scales), rotary + cache write run
        inside the attention kernel.  The activation scheme -- dynamic per token, or the checkpoint's
        static per-tensor input_scale -- must be the same for the four projections of the layer."""
        from .quantization.fp8 import CompressedTensorsW8A8Fp8Method, CDNA4Fp8LinearMethod
        if m > 64 or self.head_dim != 128 or not self.fuse_rope_attention or self.is_moe or self.has_bias:
            return False
        lins = self.linears()
        static = [getattr(lin, "input_scale", None) is not None for lin in lins]
        if any(static) != all(static):
            return False
        for lin in lins:
            qm = lin.quant_method
            if isinstance(qm, CDNA4Fp8LinearMethod):
                # Fp8Config checkpoints (per-tensor weight scale): the static activation scheme only -- the dynamic one
                # is ONE scale over the whole tensor (a cross-workgroup absmax: the op-by-op path)
                if qm.use_marlin or lin.input_scale is None or lin.weight.dtype != torch.float8_e4m3fn \
                        or lin.weight_scale.numel() != 1:
                    return False
            elif not isinstance(qm, CompressedTensorsW8A8Fp8Method):
                return False
            if lin.input_scale is not None and (lin.input_scale.numel() != 1 or lin.input_scale.dtype != torch.float32):
                return False
            if ops.fp8_gemm_ksplit(m, lin.out_features, lin.in_features) <= 0:
                return False
        return True

    def _channel_scale(self, lin) -> torch.Tensor:
        ws = lin.weight_scale
        if ws.numel() == lin.out_features:
            return ws
        cached = getattr(lin, "_weight_scale_channel", None)
        if cached is None:
            cached = ws.reshape(1).expand(lin.out_features).contiguous()
            lin._weight_scale_channel = cached
        return cached

    def forward_decode_fused_fp8(self, positions, x, prev, residual, first, kv_cache, attn_metadata, cos_sin,
                                 cos_sin_tok=None):
        """x: row-major input or None; prev = (slabs, a_scales, b_scales) of the previous layer's
        down_proj (TP == 1).  Returns (x, prev) of this layer's down_proj in the same convention."""
        eps = self.cfg.rms_norm_eps
        m = positions.shape[0]
        # static scheme: each projection's input_scale ([1]) goes to the kernel that quantises its input
        s_qkv, s_o = self.qkv_proj.input_scale, self.o_proj.input_scale
        s_gu, s_dn = self.gate_up_proj.input_scale, self.down_proj.input_scale
        if isinstance(x, DeferredAllReduce):      # TP: the previous layer's down_proj all-reduce runs inside this norm + quant launch
            qx, sx, _ = x.finish_quant_fp8(residual, self.input_layernorm, eps, static_scale=s_qkv)
        elif prev is None:
            qx, sx, _ = ops.fused_add_rms_norm_quant_fp8(x, None, None, None, residual, not first,
                                                         self.input_layernorm, eps, static_scale=s_qkv)
        else:
            qx, sx, _ = ops.fused_add_rms_norm_quant_fp8(None, prev[0], prev[1], prev[2], residual, True,
                                                         self.input_layernorm, eps, static_scale=s_qkv)
        qkv_slabs = self._fp8_slabs("qkv_proj", qx)
        from .attention.paged_attn import PagedAttention
        key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads, self.head_dim)
        # static scheme: the attention launch writes the o_proj input as e4m3 itself, and gate_up + SiluAndMul + the
        # down_proj input quantisation are one launch where the streaming kernel tiles the shape (7 launches per layer)
        fuse_static = s_o is not None and not switch("APHRO_FP8_NO_STATIC_FUSION")
        # round 6, dynamic per-token scheme at <= 32 rows (TP 1): the attention launch leaves its 16-bit output plus one absmax
        # partial per (token, kv-head), and the o_proj GEMM makes the per-token scale from the partials and quantises its A
        # fragments on load (ops.fp8_gemm_resident_aq) -- scaled_fp8_quant's launch is gone, its bits are not
        o_st = self.fp8_strip.get("o_proj") if m <= 32 else None
        aq_o = s_o is None and self.tp == 1 and o_st is not None and self.num_heads // self.num_kv_heads <= 16 and self.q_size % 64 == 0 \
            and ops.fp8_gemm_resident_aq_supported(m, self.o_proj.out_features, self.q_size, self.num_kv_heads) \
            and not switch("APHRO_FP8_NO_LAUNCH_DIET")
        attn_res = ops.paged_attention_rope_scaled(
            qkv_slabs, sx, self._channel_scale(self.qkv_proj),
            None if cos_sin_tok is not None else positions,
            cos_sin_tok if cos_sin_tok is not None else cos_sin, attn_metadata.slot_mapping,
            key_cache, value_cache, self.num_heads, self.num_kv_heads, self.attn.scale,
            attn_metadata.block_tables, attn_metadata.seq_lens_tensor, value_cache.shape[3],
            attn_metadata.max_decode_seq_len, None, self.attn.kv_cache_dtype, self.k_scale, self.v_scale,
            out_q8_scale=s_o if fuse_static else None, want_out=False, want_absmax=aq_o, out_pairs=aq_o)
        o_slabs = None
        if fuse_static:
            qa, sa = attn_res[1], s_o
            act_dtype = cos_sin.dtype if cos_sin_tok is None else cos_sin_tok.dtype
        elif aq_o:
            attn_out, attn_absmax = attn_res            # (pair-major: every A load of the GEMM is lane-linear)
            act_dtype = attn_out.dtype
            o_slabs, sa = ops.fp8_gemm_resident_aq(attn_out, attn_absmax, o_st, a_pairs=True)
        else:
            attn_out = attn_res
            act_dtype = attn_out.dtype
            qa, sa = ops.scaled_fp8_quant(attn_out.view(m, self.q_size), s_o, use_per_token_if_dynamic=True)
        if self.tp > 1:
            o = ops.cutlass_scaled_mm(qa, self.o_proj.weight, out_dtype=act_dtype, scale_a=sa,
                                      scale_b=self.o_proj.weight_scale)
            # all-reduce + residual add + norm + the gate_up input quantisation: one launch of the peer-access kernel
            # (csrc/custom_all_reduce.hip, Q8 form) where the communicator serves the shape, else the two launches
            deferred = defer_all_reduce(o)
            if deferred is not None:
                qh, sh, _ = deferred.finish_quant_fp8(residual, self.post_attention_layernorm, eps, static_scale=s_gu)
            else:
                o = tensor_model_parallel_all_reduce(o)
                qh, sh, _ = ops.fused_add_rms_norm_quant_fp8(o, None, None, None, residual, True,
                                                             self.post_attention_layernorm, eps, static_scale=s_gu)
        else:
            if o_slabs is None:
                o_slabs = self._fp8_slabs("o_proj", qa)
            qh, sh, _ = ops.fused_add_rms_norm_quant_fp8(None, o_slabs, sa, self.o_proj.weight_scale, residual,
                                                         True, self.post_attention_layernorm, eps, static_scale=s_gu)
        # gate_up: SiluAndMul in the resident kernel's epilogue on the interleaved strip copy (round 6).  Static scheme: the
        # epilogue writes the down projection's e4m3 input itself; dynamic scheme: 16-bit activation + one absmax partial per
        # (token, column strip), and the down GEMM quantises on load like o_proj above -- silu_and_mul_quant's launch is gone
        gu_il = self.fp8_gate_up_il if m <= 32 else None
        d_st = self.fp8_strip.get("down_proj") if m <= 32 else None
        down_slabs = None
        if gu_il is not None and s_dn is not None and fuse_static:
            qd = ops.fp8_gemm_resident_silu(qh, gu_il, sh, self.gate_up_proj.weight_scale, act_dtype, static_out_scale=s_dn)
            sd = s_dn
        elif gu_il is not None and s_dn is None and self.tp == 1 and d_st is not None and self.down_proj.in_features % 64 == 0 \
                and ops.fp8_gemm_resident_aq_supported(
                m, self.down_proj.out_features, self.down_proj.in_features, ops.fp8_gemm_resident_strips(m, gu_il.shape[0], gu_il.shape[1])):
            act, act_absmax = ops.fp8_gemm_resident_silu(qh, gu_il, sh, self.gate_up_proj.weight_scale, act_dtype, act_pairs=True)
            down_slabs, sd = ops.fp8_gemm_resident_aq(act, act_absmax, d_st, a_pairs=True)
        elif fuse_static and ops.fp8_gemm_silu_quant_supported(m, self.gate_up_proj.out_features,
                                                               self.gate_up_proj.in_features):
            qd = ops.fp8_gemm_silu_quant(qh, self.gate_up_proj.weight, sh, self.gate_up_proj.weight_scale, s_dn, act_dtype)
            sd = s_dn
        else:
            gu_strip = self.fp8_strip.get("gate_up_proj") if m <= 32 else None
            if gu_strip is not None and ops.fp8_gemm_resident_ksplit(m, gu_strip.shape[0], gu_strip.shape[1]) == 1:
                gate_up = ops.fp8_gemm_resident(qh, gu_strip, sh, self.gate_up_proj.weight_scale, out_dtype=act_dtype)
            else:
                gate_up = ops.cutlass_scaled_mm(qh, self.gate_up_proj.weight, out_dtype=act_dtype, scale_a=sh,
                                                scale_b=self.gate_up_proj.weight_scale)
            qd, sd, _ = ops.silu_and_mul_quant_fp8(gate_up, static_scale=s_dn)
        if self.tp > 1:
            d = ops.cutlass_scaled_mm(qd, self.down_proj.weight, out_dtype=act_dtype, scale_a=sd,
                                      scale_b=self.down_proj.weight_scale)
            deferred = defer_all_reduce(d)          # (the next layer's input norm, or the model's last norm, finishes it)
            return (deferred if deferred is not None else tensor_model_parallel_all_reduce(d)), None
        if down_slabs is None:
            down_slabs = self._fp8_slabs("down_proj", qd)
        return None, (down_slabs, sd, self.down_proj.weight_scale)

    # -- FP8 W8A8 prefill: the activation quantisations ride in the kernels that produce the activations ---------------
    def fused_prefill_fp8_ok(self) -> bool:
        """Prompt-sized batches of an FP8 W8A8 layer (compressed-tensors per-token / static, or Fp8Config static): the
        decode fast path's fused kernels at M = prompt tokens -- fused_add_rms_norm + quant and SiluAndMul + quant in one
        launch each instead of norm, quant, SiluAndMul, quant (profiles/r5_prefill_e2e_kernels.txt: the four per-token
        quantisation passes were 9.2 of 82 ms of an 8192-token prompt).  Same bits as the op-by-op path."""
        from .quantization.fp8 import CompressedTensorsW8A8Fp8Method, CDNA4Fp8LinearMethod
        if self.is_moe or self.has_bias or switch("APHRO_PREFILL_NO_FUSED_FP8"):
            return False
        lins = self.linears()
        static = [getattr(lin, "input_scale", None) is not None for lin in lins]
        if any(static) != all(static):
            return False
        for lin in lins:
            qm = lin.quant_method
            if isinstance(qm, CDNA4Fp8LinearMethod):
                if qm.use_marlin or lin.input_scale is None or lin.weight.dtype != torch.float8_e4m3fn \
                        or lin.weight_scale.numel() != 1:
                    return False
            elif not isinstance(qm, CompressedTensorsW8A8Fp8Method):
                return False
            if lin.input_scale is not None and (lin.input_scale.numel() != 1 or lin.input_scale.dtype != torch.float32):
                return False
            if getattr(lin, "bias", None) is not None:
                return False
        return True

    def forward_prefill_fp8(self, positions, hidden, residual, first, kv_cache, attn_metadata, cos_sin):
        """One decoder layer on a prompt-sized batch, FP8 W8A8: returns the down_proj output (the next norm adds it to
        ``residual``, updated in place)."""
        eps = self.cfg.rms_norm_eps
        act_dtype = hidden.dtype
        s_qkv, s_o = self.qkv_proj.input_scale, self.o_proj.input_scale
        s_gu, s_dn = self.gate_up_proj.input_scale, self.down_proj.input_scale
        mm = lambda a, sa, lin: ops.cutlass_scaled_mm(a, lin.weight, out_dtype=act_dtype, scale_a=sa, scale_b=lin.weight_scale)
        qx, sx, _ = ops.fused_add_rms_norm_quant_fp8(hidden, None, None, None, residual, not first, self.input_layernorm,
                                                     eps, static_scale=s_qkv)
        qkv = mm(qx, sx, self.qkv_proj)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        ops.rotary_embedding(positions, q, k, self.head_dim, cos_sin, True)
        attn_out = self.attn.forward(q, k, v, kv_cache, attn_metadata, self.k_scale, self.v_scale)
        qa, sa = ops.scaled_fp8_quant(attn_out.view(attn_out.shape[0], self.q_size), s_o, use_per_token_if_dynamic=True)
        o = mm(qa, sa, self.o_proj)
        if self.tp > 1:
            o = tensor_model_parallel_all_reduce(o)
        qh, sh, _ = ops.fused_add_rms_norm_quant_fp8(o, None, None, None, residual, True, self.post_attention_layernorm,
                                                     eps, static_scale=s_gu)
        gate_up = mm(qh, sh, self.gate_up_proj)
        qd, sd, _ = ops.silu_and_mul_quant_fp8(gate_up, static_scale=s_dn)
        d = mm(qd, sd, self.down_proj)
        if self.tp > 1:
            d = tensor_model_parallel_all_reduce(d)
        return d

    def forward(self, positions, hidden, residual, kv_cache, attn_metadata, cos_sin):
        eps = self.cfg.rms_norm_eps
        if residual is None:
            residual = hidden
            normed = torch.empty_like(hidden)
            ops.rms_norm(normed, hidden, self.input_layernorm, eps)
            hidden = normed
        else:
            ops.fused_add_rms_norm(hidden, residual, self.input_layernorm, eps)
        qkv = self.qkv_proj(hidden)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        ops.rotary_embedding(positions, q, k, self.head_dim, cos_sin, True)
        attn_out = self.attn.forward(q, k, v, kv_cache, attn_metadata,
                                     self.k_scale, self.v_scale)
        hidden = self._row_parallel(self.o_proj, attn_out)
        ops.fused_add_rms_norm(hidden, residual, self.post_attention_layernorm, eps)
        if self.is_moe:
            return self.moe_block(hidden), residual
        il = self.gate_up_interleaved is not None and not self.gate_up_keep_original
        if il and hidden.shape[0] > 64 and not self.has_bias and not switch("APHRO_PREFILL_NO_SILU_EPILOGUE") \
                and not switch("APHRO_WNA16_NO_LARGE"):
            # prompt-sized batches on the interleaved copy: SiluAndMul in the GEMM's epilogue (same bits, no [M, 2 I] round trip)
            qw, qz, sc, zo = self.gate_up_interleaved
            if ops.wna16_gemm_large_silu_supported(hidden.shape[0], qw.shape[1], hidden.shape[1], sc.shape[0]) \
                    and hidden.dtype == sc.dtype:
                if getattr(self.gate_up_proj, "qweight_strip_major", False):      # one resident copy: the strip-major words
                    act = ops.wna16_gemm_large_strip(hidden, qw, qz, sc, zo, silu=True)
                else:
                    act = ops.wna16_gemm_large_silu(hidden, qw, qz, sc, zo)
                hidden = self.down_proj(act)
                if self.tp > 1:
                    hidden = tensor_model_parallel_all_reduce(hidden)
                return hidden, residual
        gate_up = self.gate_up_proj(hidden)
        act = torch.empty(gate_up.shape[0], gate_up.shape[1] // 2, dtype=gate_up.dtype,
                          device=gate_up.device)
        # ONE copy of the gate_up weights: once SiluAndMul rides in the decode GEMM's epilogue the parameters hold the
        # interleaved (gate_j, up_j) column order, and the op-by-op / prompt-sized path pairs the columns in the activation
        ops.silu_and_mul(act, gate_up, interleaved=self.gate_up_interleaved is not None and not self.gate_up_keep_original)
        return self._row_parallel(self.down_proj, act), residual

    def _row_parallel(self, lin: QuantLinear, x: torch.Tensor) -> torch.Tensor:
        """RowParallelLinear.forward (linear.py:1136-1150): this rank's partial product, the all-reduce over the TP group,
        then the bias -- once, not per rank."""
        if self.tp == 1:
            return lin(x)
        out = tensor_model_parallel_all_reduce(lin(x, add_bias=False))
        return out if lin.bias is None else out + lin.bias


class LlamaForCausalLM(nn.Module):
    def __init__(self, cfg: LlamaConfig, quant_config: Optional[QuantizationConfig],
                 dtype: torch.dtype = torch.float16, kv_cache_dtype: str = "auto"):
        super().__init__()
        self.cfg = cfg
        self.dtype = dtype
        self.kv_cache_dtype = kv_cache_dtype
        self.embed_tokens = nn.Parameter(
            torch.empty(cfg.vocab_size, cfg.hidden_size, dtype=dtype), requires_grad=False)
        self.layers = nn.ModuleList([
            LlamaDecoderLayer(cfg, quant_config, dtype, kv_cache_dtype, i)
            for i in range(cfg.num_hidden_layers)])
        self.norm = nn.Parameter(torch.ones(cfg.hidden_size, dtype=dtype),
                                 requires_grad=False)
        tp = get_tensor_model_parallel_world_size()
        # ParallelLMHead (vocab_parallel_embedding.py: DEFAULT_VOCAB_PADDING_SIZE = 64): the vocabulary is padded to a
        # multiple of 64 (and of the TP size) and the PADDED size is sharded; padding rows are zero and their logits
        # are sliced away after the gather, so a checkpoint with added tokens (vocab % tp != 0) loses no token.
        self.vocab_padded = padded_vocab_size(cfg.vocab_size, tp)
        self.lm_head = nn.Parameter(
            torch.zeros(self.vocab_padded // tp, cfg.hidden_size, dtype=dtype),
            requires_grad=False)
        self.cos_sin = None
        self.use_fused_decode = True

    # -- synthetic weights in the real formats -----------------------------------
    @torch.no_grad()
    def init_synthetic(self, device, seed: int = 0):
        g = torch.Generator(device=device)
        g.manual_seed(seed + 1000 * get_tensor_model_parallel_rank())
        self.to(device)
        cfg = self.cfg

        def randn_(p, std):
            p.copy_((torch.randn(p.shape, generator=g, device=device,
                                 dtype=torch.float32) * std).to(p.dtype))

        # replicated parameters are identical on every TP rank; sharded ones draw from the rank's stream
        g_rep = torch.Generator(device=device)
        g_rep.manual_seed(seed)
        self.embed_tokens.copy_((torch.randn(self.embed_tokens.shape, generator=g_rep, device=device,
                                             dtype=torch.float32)).to(self.embed_tokens.dtype))
        randn_(self.lm_head, 1.0 / math.sqrt(cfg.hidden_size))
        rows = self.lm_head.shape[0]
        valid = max(0, min(rows, cfg.vocab_size - get_tensor_model_parallel_rank() * rows))
        if valid < rows:
            self.lm_head[valid:].zero_()
        for layer in self.layers:
            for lin in layer.linears():
                _init_linear(lin, g, device)
            if layer.is_moe:
                layer.moe_gate.copy_((torch.randn(layer.moe_gate.shape, generator=g_rep, device=device)
                                      / math.sqrt(cfg.hidden_size)).to(layer.moe_gate.dtype))
                _init_experts(layer.experts, g, device)
        self.cos_sin = _rope_cache(cfg.head_dim, cfg.max_position_embeddings,
                                   cfg.rope_theta, self.dtype, device, cfg.rope_scaling)
        self.process_weights_after_loading()
        return self

    def process_weights_after_loading(self):
        """Every quant method's post-load hook (repack / requantise), once the parameters are on
        the device (model_loader/loader.py:396-408)."""
        for layer in self.layers:
            for lin in layer.linears():
                if lin.quant_method is not None:
                    lin.quant_method.process_weights_after_loading(lin)
            if layer.is_moe:
                layer.experts.quant_method.process_weights_after_loading(layer.experts)
        return self

    def weight_bytes_per_layer(self, active_expert_fraction: float = 1.0) -> int:
        """Algorithmic bytes of the linears of one layer; for a sparse MLP the experts'
        weights count in proportion to the experts a step actually routes to."""
        layer = self.layers[0]
        n = 0
        for lin in layer.linears():
            for name, p in lin.named_parameters():
                if name in ("g_idx", "input_scale"):
                    continue
                n += p.numel() * p.element_size()
        if layer.is_moe:   # every expert's weights (at batch 32, top-2 of 8 touches all of them) + the router
            ex = getattr(layer.experts, "experts_packed", None)
            tensors = (ex.w13 + ex.w2) if ex is not None else (layer.experts.w13_weight, layer.experts.w2_weight)
            n += int(sum(t.numel() * t.element_size() for t in tensors) * active_expert_fraction)
            n += layer.moe_gate.numel() * layer.moe_gate.element_size()
        return n

    def forward(self, input_ids, positions, kv_caches, attn_metadata):
        hidden = self.embed_tokens[input_ids]
        if (self.use_fused_decode and attn_metadata.num_prefill_tokens == 0
                and attn_metadata.num_decode_tokens > 0
                and all(l.fused_decode_ok(hidden.shape[0]) for l in self.layers)):
            residual = torch.empty_like(hidden)
            x, slabs = hidden, None
            # rotary table rows of this step's positions, gathered once for all layers
            cos_sin_tok = self.cos_sin.index_select(0, positions)
            tp = get_tensor_model_parallel_world_size()
            for i, layer in enumerate(self.layers):
                # TP: the down_proj all-reduce of this layer overlaps with a prefetch of the NEXT layer's qkv weights
                nxt = None
                if tp > 1 and i + 1 < len(self.layers):
                    fp = self.layers[i + 1].qkv_proj.fast_params()
                    nxt = fp[:3] if fp is not None else None
                x, slabs = layer.forward_decode_fused(positions, x, slabs, residual, i == 0,
                                                      kv_caches[i], attn_metadata, self.cos_sin, cos_sin_tok, nxt)
            if isinstance(x, DeferredCombine):
                _, out = ops.fused_add_rms_norm_pack_combine(x.slabs, x.inv, x.topk_weights, residual, True, self.norm,
                                                             self.cfg.rms_norm_eps, pack=False, want_out=True)
                return out
            if isinstance(x, DeferredAllReduce):
                _, out = x.finish(residual, self.norm, self.cfg.rms_norm_eps, pack=False, want_out=True)
                return out
            _, out = ops.fused_add_rms_norm_pack(x if slabs is None else None, slabs, residual, True, self.norm,
                                                 self.cfg.rms_norm_eps, pack=False, want_out=True)
            return out
        if (self.use_fused_decode and attn_metadata.num_prefill_tokens == 0
                and attn_metadata.num_decode_tokens > 0
                and all(l.fused_decode_fp8_ok(hidden.shape[0]) for l in self.layers)):
            residual = torch.empty_like(hidden)
            x, prev = hidden, None
            cos_sin_tok = self.cos_sin.index_select(0, positions)
            for i, layer in enumerate(self.layers):
                x, prev = layer.forward_decode_fused_fp8(positions, x, prev, residual, i == 0, kv_caches[i],
                                                         attn_metadata, self.cos_sin, cos_sin_tok)
            if isinstance(x, DeferredAllReduce):
                _, _, out = x.finish_quant_fp8(residual, self.norm, self.cfg.rms_norm_eps, want_out=True)
            elif prev is None:
                _, _, out = ops.fused_add_rms_norm_quant_fp8(x, None, None, None, residual, True, self.norm,
                                                             self.cfg.rms_norm_eps, want_out=True)
            else:
                _, _, out = ops.fused_add_rms_norm_quant_fp8(None, prev[0], prev[1], prev[2], residual, True,
                                                             self.norm, self.cfg.rms_norm_eps, want_out=True)
            return out
        if (attn_metadata.num_prefill_tokens > 0 and attn_metadata.num_decode_tokens == 0 and hidden.shape[0] > 64
                and all(l.fused_prefill_fp8_ok() for l in self.layers)):
            # prompt-sized FP8 W8A8 batches: norm + quant and SiluAndMul + quant fused (forward_prefill_fp8)
            residual = torch.empty_like(hidden)
            for i, layer in enumerate(self.layers):
                hidden = layer.forward_prefill_fp8(positions, hidden, residual, i == 0, kv_caches[i], attn_metadata,
                                                   self.cos_sin)
            ops.fused_add_rms_norm(hidden, residual, self.norm, self.cfg.rms_norm_eps)
            return hidden
        residual = None
        for i, layer in enumerate(self.layers):
            hidden, residual = layer(positions, hidden, residual, kv_caches[i],
                                     attn_metadata, self.cos_sin)
        ops.fused_add_rms_norm(hidden, residual, self.norm, self.cfg.rms_norm_eps)
        return hidden

    def compute_logits(self, hidden):
        logits = torch.matmul(hidden, self.lm_head.t())
        if get_tensor_model_parallel_world_size() > 1:
            from .distributed import tensor_model_parallel_all_gather
            logits = tensor_model_parallel_all_gather(logits, dim=-1)
        if logits.shape[-1] != self.cfg.vocab_size:
            logits = logits[..., :self.cfg.vocab_size]          # drop the padding columns (a strided view)
        return logits

    def greedy_tokens(self, hidden, out=None):
        """Greedy next tokens straight from the final hidden states: the LM head GEMM with the argmax folded in
        (csrc/lm_head.hip) where it is served -- one rank, <= 32 rows, 16-bit, K <= 4096 --, else logits + argmax.
        APHRO_NO_LM_HEAD_ARGMAX=1 keeps the two-step path."""
        import os
        if (hidden.is_cuda and hidden.dim() == 2 and get_tensor_model_parallel_world_size() == 1
                and not switch("APHRO_NO_LM_HEAD_ARGMAX") and hidden.stride(1) == 1
                and ops.lm_head_argmax_supported(hidden.shape[0], hidden.shape[1], self.cfg.vocab_size,
                                                 self.lm_head.stride(0), hidden.dtype)):
            return ops.lm_head_argmax(hidden, self.lm_head, self.cfg.vocab_size, out)
        return self.sample_greedy(self.compute_logits(hidden), out)

    def sample_greedy(self, logits, out=None):
        if logits.is_cuda and logits.dim() == 2 and logits.stride(1) == 1:
            return ops.argmax_rows(logits, out)
        return torch.argmax(logits, dim=-1)


@torch.no_grad()
def _init_experts(moe, g, device):
    """Random experts in the checkpoint layout of the method (GPTQ / AWQ int4, or FP8 with per-tensor scales)."""
    if type(moe.quant_method).__name__ == "Fp8MoEMethod":
        for name, p in list(moe.named_parameters()):
            if name.endswith("_weight"):
                k = p.shape[2]
                for e in range(p.shape[0]):      # unit-variance values in e4m3, scale 1 / sqrt(K)-ish below
                    p[e].copy_((torch.randn(p.shape[1:], generator=g, device=device) * 64.0).clamp(-448, 448).to(p.dtype))
            elif name.endswith("weight_scale"):
                k = moe.hidden_size if name.startswith("w13") else moe.intermediate_size_per_partition * moe.tp_size
                p.copy_(((torch.rand(p.shape, generator=g, device=device) * 0.1 + 0.95) / (64.0 * math.sqrt(k))).to(p.dtype))
            elif name.endswith("input_scale"):
                p.fill_(0.02)
        return
    gs = moe.quant_method.group_size
    for name, p in list(moe.named_parameters()):
        if name.endswith(("qweight", "qzeros")):
            for e in range(p.shape[0]):      # per expert: bounded temporaries at Mixtral size
                p[e].copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, p.shape[1:], generator=g, device=device,
                                         dtype=torch.int64).to(torch.int32))
        elif name.endswith("scales"):
            k = moe.hidden_size if name.startswith("w13") else moe.intermediate_size_per_partition * moe.tp_size
            p.copy_(((torch.rand(p.shape, generator=g, device=device) * 0.5 + 0.75)
                     * (1.0 / (4.6 * math.sqrt(k)))).to(p.dtype))
        elif name.endswith("g_idx"):
            k0 = moe.tp_rank * p.shape[1] if name.startswith("w2") else 0
            p.copy_(((torch.arange(p.shape[1], device=device) + k0) // gs).to(p.dtype).expand_as(p))


def _init_linear(lin: QuantLinear, g, device):
    k = lin.in_features
    qc = lin.quant_config
    if qc is None:
        lin.weight.copy_((torch.randn(lin.weight.shape, generator=g, device=device)
                          / math.sqrt(k)).to(lin.weight.dtype))
        return
    if isinstance(qc, (GPTQConfig, AWQConfig)):
        # random nibbles are a valid quantised weight; (q - z) ~ U[-8, 7]
        for name in ("qweight", "qzeros"):
            p = getattr(lin, name)
            p.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, p.shape, generator=g,
                                  device=device, dtype=torch.int64).to(torch.int32))
        s = lin.scales
        s.copy_(((torch.rand(s.shape, generator=g, device=device) * 0.5 + 0.75)
                 * (1.0 / (4.6 * math.sqrt(k)))).to(s.dtype))
        return
    if isinstance(qc, Fp8Config):
        w = torch.randn(lin.weight.shape, generator=g, device=device) / math.sqrt(k)
        amax = w.abs().max()
        scale = (amax / 448.0).float()
        lin.weight.copy_((w / scale).clamp(-448, 448).to(torch.float8_e4m3fn))
        lin.weight_scale.fill_(scale.item())
        if getattr(lin, "input_scale", None) is not None:
            lin.input_scale.fill_(8.0 / 448.0)
        return
    if isinstance(qc, CompressedTensorsW8A8Fp8Config):
        w = torch.randn(lin.weight.shape, generator=g, device=device) / math.sqrt(k)
        if qc.strategy == "channel":
            scale = (w.abs().amax(dim=1, keepdim=True) / 448.0).float()
            lin.weight.copy_((w / scale).clamp(-448, 448).to(torch.float8_e4m3fn))
            lin.weight_scale.copy_(scale)
        else:
            scale = (w.abs().max() / 448.0).float()
            lin.weight.copy_((w / scale).clamp(-448, 448).to(torch.float8_e4m3fn))
            lin.weight_scale.fill_(scale.item())
        if getattr(lin, "input_scale", None) is not None:
            lin.input_scale.fill_(8.0 / 448.0)
        return
    raise ValueError(f"no synthetic init for {type(qc).__name__}")


def make_decode_metadata(batch: int, ctx_len, block_size: int, device,
                         blocks_per_seq: Optional[int] = None, seed: int = 0):
    """Decode-step metadata as CommonMetadataBuilder.build produces it
    (attention/backends/utils.py:191-274): every sequence has ``ctx_len``
    tokens *including* the one being generated; block tables are a random
    permutation of the block pool, slot = last token's slot."""
    if isinstance(ctx_len, int):
        seq_lens = [ctx_len] * batch
    else:
        seq_lens = list(ctx_len)
    max_len = max(seq_lens)
    bps = blocks_per_seq or (max_len + block_size - 1) // block_size
    total_blocks = batch * bps
    gen = torch.Generator().manual_seed(seed)
    perm = torch.randperm(total_blocks, generator=gen).to(torch.int32)
    block_tables = perm.view(batch, bps)
    slots = []
    for i, L in enumerate(seq_lens):
        last = L - 1
        slots.append(int(block_tables[i, last // block_size]) * block_size + last % block_size)
    meta = MI355XAttentionMetadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=batch,
        slot_mapping=torch.tensor(slots, dtype=torch.int64, device=device),
        seq_lens=None,
        seq_lens_tensor=torch.tensor(seq_lens, dtype=torch.int32, device=device),
        max_query_len=None, max_prefill_seq_len=0, max_decode_seq_len=max_len,
        query_start_loc=None, seq_start_loc=None, context_lens_tensor=None,
        block_tables=block_tables.to(device), use_cuda_graph=False)
    positions = torch.tensor([L - 1 for L in seq_lens], dtype=torch.int64, device=device)
    return meta, positions, total_blocks


def make_kv_caches(cfg: LlamaConfig, num_blocks: int, block_size: int, dtype,
                   kv_cache_dtype: str, device, num_layers: Optional[int] = None,
                   fill: bool = True, seed: int = 0):
    """worker/cache_engine.py:66-86 + common/utils.py:686-740 (uniform fill)."""
    tp = get_tensor_model_parallel_world_size()
    hkv = max(1, cfg.num_key_value_heads // tp)
    cache_dtype = dtype if kv_cache_dtype == "auto" else torch.uint8
    shape = (2, num_blocks, block_size * hkv * cfg.head_dim)
    caches = []
    g = torch.Generator(device=device).manual_seed(seed)
    for _ in range(num_layers or cfg.num_hidden_layers):
        if not fill:
            caches.append(torch.zeros(shape, dtype=cache_dtype, device=device))
        elif cache_dtype == torch.uint8:
            # random e4m3 bytes without NaN patterns (0x7f / 0xff)
            # (|x| <= 3.75 as e4m3, finite as e5m2)
            c = torch.randint(0, 0x48, shape, generator=g, device=device, dtype=torch.int16)
            sign = torch.randint(0, 2, shape, generator=g, device=device, dtype=torch.int16) << 7
            caches.append((c | sign).to(torch.uint8))
        else:
            c = torch.rand(shape, generator=g, device=device, dtype=torch.float32)
            caches.append(((c * 2 - 1) * cfg.head_dim ** -0.5).to(cache_dtype))
    return caches
