"""KV-cache scale handling -- mirror of aphrodite/quantization/kv_cache.py:8-75
and attention/layer.py:52-76: per-tensor k_scale / v_scale python floats."""
import torch
from torch import nn

from .base_config import QuantizeMethodBase


class BaseKVCacheMethod(QuantizeMethodBase):
    def __init__(self, quant_config):
        self.quant_config = quant_config

    def create_weights(self, layer: nn.Module):
        layer.k_scale = nn.Parameter(torch.tensor(-1.0), requires_grad=False)
        layer.v_scale = nn.Parameter(torch.tensor(-1.0), requires_grad=False)

    def apply(self, layer: nn.Module) -> torch.Tensor:
        raise RuntimeError(f"{self.__class__.__name__}.apply should not be called.")

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        # kv_cache.py:37-75 (gfx950 is OCP: no fnuz doubling)
        if layer.kv_cache_dtype != "auto":
            if layer.k_scale > 0.0 and layer.v_scale > 0.0:
                k_scale = layer.k_scale.to("cpu").tolist()
                v_scale = layer.v_scale.to("cpu").tolist()
            elif layer.k_scale < 0.0 and layer.v_scale < 0.0:
                k_scale = v_scale = 1.0
            else:
                assert layer.k_scale > 0.0
                scale_to_duplicate = max(layer.k_scale, layer.v_scale)
                k_scale = v_scale = scale_to_duplicate.to("cpu").tolist()
            if not isinstance(k_scale, float) or not isinstance(v_scale, float):
                raise ValueError("Only support per-tensor scaling factor for fp8 KV cache")
            layer._k_scale = k_scale
            layer._v_scale = v_scale
        del layer.k_scale
        del layer.v_scale


_REF_SUBCLASS = {}


def make_kv_cache_method(quant_config) -> BaseKVCacheMethod:
    """What a config's ``get_quant_method`` returns for an Attention layer.  Under the plugin the caller is the
    REFERENCE's ``Attention.__init__``, which asserts ``isinstance(quant_method, BaseKVCacheMethod)`` against ITS class
    (attention/layer.py:61-64): when ``aphrodite.quantization.kv_cache`` is importable the instance derives from both
    (ours first in the MRO: same behaviour, and the assertion holds); standalone it is plainly ours."""
    try:
        from aphrodite.quantization.kv_cache import BaseKVCacheMethod as ref_cls
    except Exception:                # the reference package is not installed / not importable here
        return BaseKVCacheMethod(quant_config)
    cls = _REF_SUBCLASS.get(ref_cls)
    if cls is None:
        cls = type("BaseKVCacheMethod", (BaseKVCacheMethod, ref_cls), {})
        _REF_SUBCLASS[ref_cls] = cls
    return cls(quant_config)
