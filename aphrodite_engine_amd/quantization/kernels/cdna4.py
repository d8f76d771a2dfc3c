"""CDNA4LinearKernel -- the MPLinearKernel the MI355X registers where Machete /
Marlin sit on NVIDIA (reference: quantization/kernels/marlin.py:18-132 is the
contract being honoured: can_implement / process_weights_after_loading /
apply_weights).  Weight layout after loading is our own K-packed exllama
order [K/8, N] -- no Marlin tile layout, no workspace locks.

Accepted checkpoint parameter layouts (those ``gptq_marlin.py`` and
``compressed_tensors_wNa16.py`` hand to an MPLinearKernel):
  w_q   int32 [K/8, N] packed along K (GPTQ order), uint4b8 or uint4+zp -- or [K/4, N], uint8b128 (round 3: the 8-bit
        GPTQ kernels of csrc/wnx_gemm.hip behind the same seam, as Marlin serves both widths: kernels/marlin.py:27-31)
  w_s   [G, N]
  w_zp  int32 [G, N/8] packed along N, plain column order (optional)
  g_idx int32 [K] (optional, act-order)
or the compressed-tensors ``pack_quantized`` orientation (``compressed_tensors_wNa16.py:97-135``:
``weight_packed`` int32 [N, K/8] with ``input_dim=1, output_dim=0, packed_dim=1``, ``weight_scale``
[N, G]): parameters that carry ``input_dim`` / ``output_dim`` attributes are brought to the
[K.., N] orientation first, as ``permute_param_layout_`` does for Marlin
(``quantization/kernels/marlin.py:90-109``).
"""
from typing import Optional, Tuple

import torch

from ... import _custom_ops as ops
from ...scalar_type import scalar_types
from .MPLinearKernel import MPLinearKernel, MPLinearLayerConfig


class CDNA4LinearKernel(MPLinearKernel):
    SUPPORTED_TYPES = (scalar_types.uint4b8, scalar_types.uint4, scalar_types.uint8b128)

    @classmethod
    def get_min_capability(cls) -> int:
        return 95  # gfx950

    @staticmethod
    def _type_key(t):
        """(bits, bias, signed, integer) of a ScalarType -- ours (scalar_type.py) or the reference's (the C++ class of
        aphrodite._core_ext / its Python mock, _core_ext.py:28-171: no common base class, so compare by value; through the
        plugin the reference's own MPLinearLayerConfig arrives here with ITS scalar_types.uint4b8)."""
        try:
            is_int = bool(t.is_integer()) if hasattr(t, "is_integer") else True
            signed = bool(t.is_signed()) if hasattr(t, "is_signed") else bool(getattr(t, "signed", False))
            return int(t.size_bits), int(getattr(t, "bias", 0) or 0), signed, is_int
        except Exception:
            return None

    @classmethod
    def can_implement(cls, c: MPLinearLayerConfig) -> Tuple[bool, Optional[str]]:
        wt = cls._type_key(c.weight_type)
        if wt not in tuple(cls._type_key(x) for x in cls.SUPPORTED_TYPES):
            return False, f"Quant type ({c.weight_type}) not supported by CDNA4 kernel"
        if c.zero_points != (wt == cls._type_key(scalar_types.uint4)):
            return False, "zero points must accompany uint4 (and only uint4)"
        k, n = c.partition_weight_shape
        if wt[0] == 8 and c.group_size != -1 and c.group_size % 4 != 0:
            return False, f"group size {c.group_size} must be a multiple of 4 for 8-bit weights"
        gs = c.group_size if c.group_size != -1 else c.full_weight_shape[0]
        if gs % 32 != 0 or k % gs != 0:
            return False, f"group size {gs} must be a multiple of 32 dividing K={k}"
        if k % 32 != 0 or n % 16 != 0:
            return False, f"K={k} must be a multiple of 32 and N={n} of 16"
        if c.has_g_idx and k != c.full_weight_shape[0]:
            # act-order rows of one K shard reference groups of the whole matrix in uneven numbers
            return False, "act-order (g_idx) with a K-sharded (row-parallel) weight is not supported"
        if c.act_type not in (torch.float16, torch.bfloat16):
            return False, f"activation dtype {c.act_type} not supported"
        return True, None

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        c = self.config
        w_q, w_s, w_zp, w_gidx = self._get_weight_params(layer)
        device = w_q.device
        k, n = c.partition_weight_shape
        perm = torch.empty(0, dtype=torch.int32, device=device)
        if c.has_g_idx and w_gidx is not None and w_gidx.numel() > 0:
            perm = torch.argsort(w_gidx).to(torch.int32)
        def kn(x):  # [out, in..] (compressed-tensors) -> [in.., out]
            if getattr(x, "output_dim", 1) == 0:
                return x.data.t().contiguous()
            return x.data.contiguous()

        layer._cdna4_bits = self._type_key(c.weight_type)[0]
        if layer._cdna4_bits == 8:
            # uint8b128: the checkpoint's sequential [K/4, N] words are what gptq_gemm(bit=8) reads (act-order rows made
            # sequential by gptq_shuffle, as GPTQLinearMethod does); the zero point 128 of every group in GPTQ's
            # stored-minus-one convention is 127 per byte
            def seq8(x):
                w = kn(x)
                if perm.numel() > 0:
                    ops.gptq_shuffle(w, perm, 8)
                return w
            self._transform_param(layer, self.w_q_name, seq8)
            self._transform_param(layer, self.w_s_name, kn)
            groups = getattr(layer, self.w_s_name).shape[0]
            layer.register_buffer("_cdna4_zp", torch.full((groups, n // 4), 0x7f7f7f7f, dtype=torch.int32, device=device),
                                  persistent=False)
            layer._cdna4_perm = perm if perm.numel() > 0 else None
            return

        self._transform_param(
            layer, self.w_q_name,
            lambda x: ops.gptq_marlin_repack(kn(x), perm, k, n, 4))
        self._transform_param(layer, self.w_s_name, kn)
        if c.zero_points:
            self._transform_param(layer, self.w_zp_name, kn)
        else:
            # symmetric uint4b8: zero point 8 for every column, stored once
            groups = getattr(layer, self.w_s_name).shape[0]
            zp = torch.full((groups, n // 8), 0x88888888 - (1 << 32), dtype=torch.int64,
                            device=device).to(torch.int32)
            layer.register_buffer("_cdna4_zp", zp, persistent=False)
        layer._cdna4_perm = perm if perm.numel() > 0 else None

    def apply_weights(self, layer: torch.nn.Module, x: torch.Tensor,
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        c = self.config
        w_q, w_s, w_zp, _ = self._get_weight_params(layer)
        if not c.zero_points:
            w_zp = layer._cdna4_zp
        x2 = x.reshape(-1, x.shape[-1])
        if getattr(layer, "_cdna4_bits", 4) == 8:
            perm = layer._cdna4_perm
            out = ops.gptq_gemm(x2, w_q, w_zp, w_s, perm if perm is not None else torch.empty(0, dtype=torch.int32, device=x.device),
                                True, 8)
            if bias is not None:
                out.add_(bias)
            return out.reshape(x.shape[:-1] + (c.partition_weight_shape[1], ))
        out = ops.wna16_gemm(x2, w_q, w_zp, w_s, layer._cdna4_perm, 0)
        if bias is not None:
            out.add_(bias)
        return out.reshape(x.shape[:-1] + (c.partition_weight_shape[1], ))
