"""Helpers shared by the quantisation configs: which layers a checkpoint left unquantised.
Behaviour of aphrodite/quantization/compressed_tensors/utils.py:113-171 (``should_ignore_layer``,
``re:`` targets) and quantization/utils/quant_utils.py ``is_layer_skipped`` (exact names)."""
import re
from typing import List, Optional

# fused module -> the projections it is stored as on disk
FUSED_LAYER_SHARDS = {"qkv_proj": ["q_proj", "k_proj", "v_proj"], "gate_up_proj": ["gate_proj", "up_proj"]}


def name_matches(name: str, target: str) -> bool:
    if target.startswith("re:"):
        return re.match(target[3:], name) is not None
    return name == target


def layer_is_ignored(prefix: Optional[str], ignore: List[str]) -> bool:
    """A fused layer is ignored iff all of its on-disk shards are; a mix is an error."""
    if prefix is None or not ignore:
        return False
    proj = prefix.split(".")[-1]
    names = [prefix.replace(proj, s) for s in FUSED_LAYER_SHARDS[proj]] if proj in FUSED_LAYER_SHARDS else [prefix]
    verdicts = {any(name_matches(n, t) for t in ignore) for n in names}
    if len(verdicts) != 1:
        raise ValueError(f"Found a different quantization schemes for {FUSED_LAYER_SHARDS[proj]} in {prefix}. "
                         "All shards of a fused layer must use the same scheme.")
    return verdicts.pop()


def layer_kind(layer) -> str:
    """Which family a layer asking for its quant method belongs to, by class NAME along the MRO (the
    reference dispatches with isinstance on LinearBase / FusedMoE / Attention / ParallelLMHead --
    fp8.py:79-92, gptq.py:75-80 -- which must work here without importing the reference):
    "moe", "attention", "embedding" (vocab embedding / lm_head), else "linear"."""
    names = {c.__name__ for c in type(layer).__mro__}
    if "FusedMoE" in names:
        return "moe"
    if "Attention" in names:
        return "attention"
    if names & {"VocabParallelEmbedding", "ParallelLMHead"}:
        return "embedding"
    return "linear"


def unquantized_linear_method():
    """What an ignored linear layer gets: the reference's UnquantizedLinearMethod when the reference is
    importable (its LinearBase needs a method object), else None (our QuantLinear keeps a 16-bit weight)."""
    try:
        from aphrodite.modeling.layers.linear import UnquantizedLinearMethod
        return UnquantizedLinearMethod()
    except Exception:
        return None
