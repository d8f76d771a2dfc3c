"""On-disk format ingestion: Hugging Face checkpoint directories in the GPTQ
(AutoGPTQ v1), AWQ, AutoFP8 / ``fp8`` and compressed-tensors (``float-quantized``,
``pack-quantized``) formats -> the parameters of the quantised Llama skeleton, sharded for
tensor parallelism, plus FP8 KV-cache scales.

What is restated from the reference (behaviour, not code):
  * the per-parameter shard arithmetic of ``MergedColumnParallelLinear`` / ``QKVParallelLinear`` /
    ``RowParallelLinear`` (modeling/layers/linear.py:452-595, 815-988, 1072-1112): logical output
    slices, packing along the sharded dimension (``packed_dim`` / ``pack_factor``), KV-head
    replication when TP > num_kv_heads (:701-712), one scalar per fused shard
    (``adjust_scalar_to_fused_array`` :68-88), weights already fused on disk;
  * the Llama name mapping (modeling/models/llama.py:480-542): q/k/v -> qkv_proj, gate/up ->
    gate_up_proj, skipped rotary buffers, skipped GPTQ biases, tied embeddings;
  * quantisation-config discovery (modeling/model_loader/weight_utils.py:118-195) and the
    dtype / capability gates of the loader (model_loader/loader.py:90-112);
  * KV-cache scales from the checkpoint (``k_scale`` / ``v_scale`` / deprecated ``kv_scale``,
    compressed-tensors ``{k,v}_proj.output_scale``: weight_utils.py:632-680,
    compressed_tensors/utils.py:226-239; rules of quantization/kv_cache.py:37-75) or from a
    ``quantization_param_path`` JSON (weight_utils.py:504-541, quantization/schema.py).
gfx950 is OCP e4m3: none of the reference's fnuz scale doubling (llama.py:557-562) applies.

The shard arithmetic is expressed once, as data: every parallel linear owns a ``ShardPlan`` (a
table of logical output slices + the rank's place in them) and ONE generic copy routine reads the
parameter's metadata (``input_dim`` / ``output_dim`` / ``packed_dim`` / ``pack_factor`` /
``needs_scalar_to_array``) against it."""
import glob
import json
import os
import re
from dataclasses import dataclass
from typing import Any, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import torch

from .distributed import get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size

ShardId = Union[str, int, None]

DEVICE_CAPABILITY = 95   # gfx950 reports (9, 5)


# --------------------------------------------------------------------------------------------------
# shard plans
# --------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class OutSlice:
    """One logical matrix of a fused column-parallel layer."""
    key: ShardId        # "q" / "k" / "v" or 0 / 1 -- the shard id the model's name mapping hands over
    total: int          # its full output width in the checkpoint
    local: int          # the width this rank holds
    src_index: int      # which ``local``-wide slice of the checkpoint tensor this rank takes


@dataclass(frozen=True)
class ShardPlan:
    """How one parallel linear's parameters are cut.  ``outs`` is empty for a row-parallel layer
    (the input dimension is cut instead: slice ``rank`` of ``param.shape[input_dim]``)."""
    outs: Tuple[OutSlice, ...]
    rank: int
    world: int

    @property
    def row_parallel(self) -> bool:
        return not self.outs

    def find(self, key: ShardId) -> int:
        for i, o in enumerate(self.outs):
            if o.key == key:
                return i
        raise ValueError(f"Unknown shard id {key!r} (this layer has {[o.key for o in self.outs]})")


def qkv_plan(total_heads: int, total_kv_heads: int, head_size: int, rank: Optional[int] = None,
             world: Optional[int] = None) -> ShardPlan:
    """Heads split over the ranks; when there are fewer KV heads than ranks each KV head is
    replicated on world / kv_heads consecutive ranks (linear.py:701-712)."""
    rank = get_tensor_model_parallel_rank() if rank is None else rank
    world = get_tensor_model_parallel_world_size() if world is None else world
    if total_heads % world != 0:
        raise ValueError(f"{total_heads} attention heads are not divisible by tensor parallel size {world}")
    if world >= total_kv_heads:
        if world % total_kv_heads != 0:
            raise ValueError(f"tensor parallel size {world} is not a multiple of {total_kv_heads} KV heads")
        kv_local, replicas = 1, world // total_kv_heads
    else:
        if total_kv_heads % world != 0:
            raise ValueError(f"{total_kv_heads} KV heads are not divisible by tensor parallel size {world}")
        kv_local, replicas = total_kv_heads // world, 1
    q = OutSlice("q", total_heads * head_size, total_heads // world * head_size, rank)
    k = OutSlice("k", total_kv_heads * head_size, kv_local * head_size, rank // replicas)
    v = OutSlice("v", total_kv_heads * head_size, kv_local * head_size, rank // replicas)
    return ShardPlan((q, k, v), rank, world)


def merged_plan(output_sizes: Sequence[int], rank: Optional[int] = None, world: Optional[int] = None) -> ShardPlan:
    rank = get_tensor_model_parallel_rank() if rank is None else rank
    world = get_tensor_model_parallel_world_size() if world is None else world
    for n in output_sizes:
        if n % world != 0:
            raise ValueError(f"output size {n} is not divisible by tensor parallel size {world}")
    return ShardPlan(tuple(OutSlice(i, n, n // world, rank) for i, n in enumerate(output_sizes)), rank, world)


def row_plan(rank: Optional[int] = None, world: Optional[int] = None) -> ShardPlan:
    rank = get_tensor_model_parallel_rank() if rank is None else rank
    world = get_tensor_model_parallel_world_size() if world is None else world
    return ShardPlan((), rank, world)


# --------------------------------------------------------------------------------------------------
# the one copy routine
# --------------------------------------------------------------------------------------------------
def _as_scalar(t: torch.Tensor) -> torch.Tensor:
    # AutoFP8 scales are 0-dim, compressed-tensors scales have shape [1] (linear.py:82-86)
    if t.dim() == 0:
        return t
    if t.numel() != 1:
        raise ValueError(f"expected one scale per logical matrix, got shape {tuple(t.shape)}")
    return t.reshape(())


def _checked_copy(dst: torch.Tensor, src: torch.Tensor, what: str) -> None:
    if dst.shape != src.shape:
        raise ValueError(f"{what}: checkpoint tensor {tuple(src.shape)} does not fit parameter slice "
                         f"{tuple(dst.shape)}")
    dst.copy_(src)


def load_sharded(plan: ShardPlan, param: torch.nn.Parameter, loaded: torch.Tensor, shard_id: ShardId = None,
                 what: str = "") -> None:
    """Copy this rank's part of ``loaded`` (one checkpoint tensor) into ``param``."""
    data = param.data
    out_dim = getattr(param, "output_dim", None)
    in_dim = getattr(param, "input_dim", None)
    scalar_array = getattr(param, "needs_scalar_to_array", False)

    if plan.row_parallel:
        if shard_id is not None:
            raise ValueError(f"{what}: a row-parallel layer has no output shards (got {shard_id!r})")
        if in_dim is not None:
            size = data.shape[in_dim]
            loaded = loaded.narrow(in_dim, plan.rank * size, size)
        if loaded.dim() == 0:
            loaded = loaded.reshape(1)
        _checked_copy(data, loaded, what)
        return

    pack = getattr(param, "pack_factor", 1) if (out_dim is not None
                                                and getattr(param, "packed_dim", None) == out_dim) else 1

    def units(n: int) -> int:          # logical columns -> storage elements along out_dim
        if n % pack != 0:
            raise ValueError(f"{what}: width {n} is not a multiple of the pack factor {pack}")
        return n // pack

    if shard_id is None:               # tensor already fused on disk (qkv_proj / gate_up_proj)
        if out_dim is None:
            if scalar_array:
                data[0].copy_(_as_scalar(loaded))
            else:
                _checked_copy(data, loaded, what)
            return
        start = 0
        for o in plan.outs:
            load_sharded(plan, param, loaded.narrow(out_dim, units(start), units(o.total)), o.key, what)
            start += o.total
        return

    idx = plan.find(shard_id)
    if out_dim is None:
        if scalar_array:
            data[idx].copy_(_as_scalar(loaded))
        else:                           # replicated metadata (e.g. g_idx of a column-parallel layer)
            _checked_copy(data, loaded, what)
        return
    o = plan.outs[idx]
    dst_off = units(sum(p.local for p in plan.outs[:idx]))
    size = units(o.local)
    _checked_copy(data.narrow(out_dim, dst_off, size), loaded.narrow(out_dim, o.src_index * size, size), what)


def make_weight_loader(plan: ShardPlan):
    """The callable a quant method stores on its parameters (``weight_loader`` attribute)."""
    def weight_loader(param, loaded_weight, loaded_shard_id: ShardId = None):
        load_sharded(plan, param, loaded_weight, loaded_shard_id)
    weight_loader.plan = plan
    return weight_loader


def default_weight_loader(param: torch.nn.Parameter, loaded: torch.Tensor) -> None:
    """Replicated parameters (norms, embeddings): shapes must agree (weight_utils.py:576-591)."""
    if param.numel() == 1 and loaded.numel() == 1:
        param.data.fill_(loaded.item())
        return
    _checked_copy(param.data, loaded, "replicated parameter")


# --------------------------------------------------------------------------------------------------
# checkpoint files
# --------------------------------------------------------------------------------------------------
def read_hf_config(model_dir: str) -> Dict[str, Any]:
    path = os.path.join(model_dir, "config.json")
    if not os.path.isfile(path):
        raise FileNotFoundError(f"{path} not found")
    with open(path) as f:
        return json.load(f)


def llama_config_from_hf(hf: Dict[str, Any]):
    from .model import LlamaConfig
    # what this skeleton does not model is refused, never loaded approximately
    rs = hf.get("rope_scaling")
    if rs and rs.get("rope_type", rs.get("type")) not in ("llama3", "linear", "dynamic", "yarn"):
        raise NotImplementedError(f"rope_scaling={rs!r}: the llama3, linear, dynamic and yarn schemes are implemented")
    if rs and hf.get("original_max_position_embeddings"):       # models/llama.py:196-200
        rs = dict(rs, original_max_position_embeddings=hf["original_max_position_embeddings"])
    if hf.get("sliding_window"):
        # (the attention backend serves the window -- MI355XAttentionImpl, the metadata builder's trimmed block tables --
        #  under the reference's own model class; this skeleton's standalone metadata has no window)
        raise NotImplementedError("sliding-window checkpoints run through the reference's model class on this package's "
                                  "attention backend, not through the fused step")
    if hf.get("hidden_act", "silu") != "silu":
        raise NotImplementedError(f"hidden_act={hf['hidden_act']!r}: only SiluAndMul MLPs are implemented")
    heads = hf["num_attention_heads"]
    head_dim = hf.get("head_dim") or hf["hidden_size"] // heads
    if head_dim * heads != hf["hidden_size"]:
        raise ValueError("head_dim * num_attention_heads != hidden_size is not supported by this skeleton")
    return LlamaConfig(hidden_size=hf["hidden_size"], intermediate_size=hf["intermediate_size"],
                       num_hidden_layers=hf["num_hidden_layers"], num_attention_heads=heads,
                       num_key_value_heads=hf.get("num_key_value_heads", heads), vocab_size=hf["vocab_size"],
                       rms_norm_eps=hf.get("rms_norm_eps", 1e-6), rope_theta=hf.get("rope_theta", 10000.0),
                       max_position_embeddings=hf.get("max_position_embeddings", 8192),
                       rope_scaling=rs or None,
                       num_local_experts=hf.get("num_local_experts", 0) or 0,
                       num_experts_per_tok=hf.get("num_experts_per_tok", 2),
                       # models/llama.py:206-211: attention_bias, or the `bias` of the internlm / abacusai exports; mlp_bias
                       attention_bias=bool(hf.get("attention_bias", False) or hf.get("bias", False)),
                       mlp_bias=bool(hf.get("mlp_bias", False)))


def resolve_quant_config(model_dir: str, hf: Dict[str, Any], quantization: Optional[str] = None,
                         dtype: torch.dtype = torch.float16):
    """The checkpoint's QuantizationConfig (or None): from ``quantization_config`` /
    ``compression_config`` inside config.json, else from the method's own json next to the weights
    (``quantize_config.json`` for GPTQ); then the loader's dtype and capability gates."""
    from .quantization import get_quantization_config
    embedded = hf.get("quantization_config") or hf.get("compression_config")
    if embedded is not None:
        method = str(embedded.get("quant_method", "")).lower()
        if not method and ("config_groups" in embedded or "format" in embedded):
            method = "compressed-tensors"
        if quantization is not None and quantization != method:
            raise ValueError(f"Quantization method specified in the model config ({method}) does not match the "
                             f"quantization argument ({quantization}).")
        quantization = method
    if quantization is None:
        return None
    cls = get_quantization_config(quantization)
    if embedded is not None:
        cfg = cls.from_config(embedded)
    else:
        names = cls.get_config_filenames()
        if not names:
            cfg = cls()
        else:
            found = [p for p in glob.glob(os.path.join(model_dir, "*.json"))
                     if any(p.endswith(n) for n in names)]
            if len(found) == 0:
                raise ValueError(f"Cannot find the config file for {quantization}")
            if len(found) > 1:
                raise ValueError(f"Found multiple config files for {quantization}: {found}")
            with open(found[0]) as f:
                cfg = cls.from_config(json.load(f))
    if dtype not in cfg.get_supported_act_dtypes():
        raise ValueError(f"{dtype} is not supported for quantization method {quantization}. Supported dtypes: "
                         f"{cfg.get_supported_act_dtypes()}")
    if DEVICE_CAPABILITY < cfg.get_min_capability():
        raise ValueError(f"The quantization method {quantization} is not supported for the current GPU. Minimum "
                         f"capability: {cfg.get_min_capability()}. Current capability: {DEVICE_CAPABILITY}.")
    return cfg


def iter_safetensors(model_dir: str) -> Iterator[Tuple[str, torch.Tensor]]:
    """(name, tensor) over every ``*.safetensors`` file of the directory; with a
    ``model.safetensors.index.json`` only the files it lists (weight_utils.py:280-330)."""
    from safetensors import safe_open
    files = sorted(glob.glob(os.path.join(model_dir, "*.safetensors")))
    index = os.path.join(model_dir, "model.safetensors.index.json")
    if os.path.isfile(index):
        with open(index) as f:
            listed = set(json.load(f)["weight_map"].values())
        files = [p for p in files if os.path.basename(p) in listed]
    if not files:
        raise RuntimeError(f"Cannot find any model weights with `{model_dir}`")
    for path in files:
        with safe_open(path, framework="pt", device="cpu") as f:
            for name in f.keys():
                yield name, f.get_tensor(name)


# --------------------------------------------------------------------------------------------------
# Llama name mapping
# --------------------------------------------------------------------------------------------------
_FUSED = {"q_proj": ("qkv_proj", "q"), "k_proj": ("qkv_proj", "k"), "v_proj": ("qkv_proj", "v"),
          "gate_proj": ("gate_up_proj", 0), "up_proj": ("gate_up_proj", 1)}
_LAYER_RE = re.compile(r"^(?:model\.)?layers\.(\d+)\.(.+)$")


@dataclass(frozen=True)
class Target:
    kind: str                       # "linear" | "expert" | "param" | "kv_scale" | "skip"
    path: str = ""                  # attribute path inside LlamaForCausalLM
    attr: str = ""                  # parameter name inside the linear / "k" or "v" or "kv" for scales
    shard: ShardId = None
    layer: int = -1
    expert: int = -1


def map_llama_name(name: str, tie_word_embeddings: bool = False) -> Target:
    """Checkpoint tensor name -> where it goes (llama.py:480-542)."""
    if "rotary_emb.inv_freq" in name or "rotary_emb.cos_cached" in name or "rotary_emb.sin_cached" in name:
        return Target("skip")
    if name in ("model.embed_tokens.weight", "embed_tokens.weight"):
        return Target("param", "embed_tokens")
    if name in ("model.norm.weight", "norm.weight"):
        return Target("param", "norm")
    if name == "lm_head.weight":
        return Target("skip") if tie_word_embeddings else Target("param", "lm_head")
    m = _LAYER_RE.match(name)
    if m is None:
        raise KeyError(f"unexpected checkpoint tensor {name!r}")
    layer, rest = int(m.group(1)), m.group(2)
    if rest in ("input_layernorm.weight", "post_attention_layernorm.weight"):
        return Target("param", f"layers.{layer}.{rest[:-len('.weight')]}", layer=layer)
    # FP8 KV-cache scales: per-layer python floats on the attention module
    if rest in ("self_attn.k_scale", "self_attn.attn.k_scale", "self_attn.k_proj.output_scale"):
        return Target("kv_scale", attr="k", layer=layer)
    if rest in ("self_attn.v_scale", "self_attn.attn.v_scale", "self_attn.v_proj.output_scale"):
        return Target("kv_scale", attr="v", layer=layer)
    if rest in ("self_attn.kv_scale", "self_attn.attn.kv_scale"):   # deprecated spelling: one scale for both
        return Target("kv_scale", attr="kv", layer=layer)
    parts = rest.split(".")
    # Mixtral (modeling/models/mixtral.py:400-470): replicated router + experts.{e}.w{1,2,3}.<tensor>
    if rest == "block_sparse_moe.gate.weight":
        return Target("param", f"layers.{layer}.moe_gate", layer=layer)
    if len(parts) == 5 and parts[:2] == ["block_sparse_moe", "experts"] and parts[3] in ("w1", "w2", "w3"):
        fused = "w13_" if parts[3] in ("w1", "w3") else "w2_"
        return Target("expert", f"layers.{layer}.experts", fused + parts[4], parts[3], layer, int(parts[2]))
    if len(parts) == 3 and parts[0] in ("self_attn", "mlp"):
        _, proj, attr = parts
        if proj in _FUSED:
            fused, shard = _FUSED[proj]
            return Target("linear", f"layers.{layer}.{fused}", attr, shard, layer)
        if proj in ("qkv_proj", "gate_up_proj", "o_proj", "down_proj"):
            return Target("linear", f"layers.{layer}.{proj}", attr, None, layer)
    raise KeyError(f"unexpected checkpoint tensor {name!r}")


def _get_path(root: torch.nn.Module, path: str):
    obj = root
    for p in path.split("."):
        obj = obj[int(p)] if p.isdigit() else getattr(obj, p)
    return obj


def finalize_kv_scales(found: Dict[int, Dict[str, float]], num_layers: int, kv_cache_dtype: str
                       ) -> List[Tuple[float, float]]:
    """quantization/kv_cache.py:37-75: both scales given -> used; none -> 1.0; only one (or the
    deprecated kv_scale) -> the larger is used for both.  Ignored (1.0) for a 16-bit cache."""
    out = []
    for i in range(num_layers):
        got = found.get(i, {})
        k, v = got.get("k", got.get("kv", -1.0)), got.get("v", -1.0)
        if kv_cache_dtype == "auto":
            out.append((1.0, 1.0))
        elif k > 0.0 and v > 0.0:
            out.append((float(k), float(v)))
        elif k < 0.0 and v < 0.0:
            out.append((1.0, 1.0))
        else:
            s = float(max(k, v))
            out.append((s, s))
    return out


def load_llama_weights(model, weights: Iterable[Tuple[str, torch.Tensor]], tie_word_embeddings: bool = False
                       ) -> Dict[int, Dict[str, float]]:
    """Feed (name, tensor) pairs into the skeleton's parameters; returns the KV scales found."""
    rank = get_tensor_model_parallel_rank()
    kv: Dict[int, Dict[str, float]] = {}
    seen_lm_head = False
    for name, tensor in weights:
        tgt = map_llama_name(name, tie_word_embeddings)
        if tgt.kind == "skip":
            continue
        if tgt.kind == "kv_scale":
            kv.setdefault(tgt.layer, {})[tgt.attr] = float(tensor.reshape(-1)[0].item())
            continue
        if tgt.kind == "param":
            param = _get_path(model, tgt.path)
            if tgt.path == "lm_head":       # vocab-parallel rows of the PADDED vocabulary (ParallelLMHead)
                _load_vocab_rows(param, tensor, rank)
                seen_lm_head = True
                continue
            default_weight_loader(param, tensor.to(param.dtype))
            continue
        if tgt.kind == "expert":
            moe = _get_path(model, tgt.path)
            param = getattr(moe, tgt.attr, None)
            if not isinstance(param, torch.nn.Parameter):
                if tgt.attr.endswith("bias"):
                    continue
                raise KeyError(f"{name}: the expert layer has no parameter {tgt.attr!r}")
            moe.weight_loader(param, tensor, name, tgt.shard, tgt.expert)
            continue
        linear = _get_path(model, tgt.path)
        param = getattr(linear, tgt.attr, None)
        if not isinstance(param, torch.nn.Parameter):
            if tgt.attr == "bias":          # extra bias tensors of GPTQ exports (llama.py:516-518, 528-530)
                continue
            raise KeyError(f"{name}: layer {tgt.path} has no parameter {tgt.attr!r}")
        loader = getattr(param, "weight_loader", None)
        if loader is None:
            default_weight_loader(param, tensor)
        else:
            try:
                loader(param, tensor, tgt.shard)
            except ValueError as e:
                raise ValueError(f"{name}: {e}") from None
    if tie_word_embeddings or not seen_lm_head:
        _load_vocab_rows(model.lm_head, model.embed_tokens.data, rank)
    return kv


def _load_vocab_rows(param: torch.nn.Parameter, full: torch.Tensor, rank: int) -> None:
    """This rank's rows [rank * rows, (rank + 1) * rows) of the vocabulary padded to a multiple of 64 (and of tp):
    the rows the checkpoint holds are copied, the padding stays zero (VocabParallelEmbedding.weight_loader)."""
    rows = param.shape[0]
    start = rank * rows
    valid = max(0, min(rows, full.shape[0] - start))
    param.data.zero_()
    if valid:
        param.data[:valid].copy_(full.narrow(0, start, valid).to(param.dtype))


def read_kv_cache_scales(path: str, tp_rank: int, tp_size: int, num_hidden_layers: int,
                         model_type: Optional[str] = None) -> Dict[int, float]:
    """``quantization_param_path`` JSON -> {layer: scale} for this rank.  Schema of
    quantization/schema.py: {"model_type": ..., "kv_cache": {"dtype": "float8_e4m3fn",
    "scaling_factor": {rank: {layer: float}}}}.  Raises on a malformed file (the reference logs and
    falls back to 1.0, weight_utils.py:529-541; a silent fallback hides wrong outputs)."""
    with open(path) as f:
        doc = json.load(f)
    kvc = doc["kv_cache"]
    if kvc.get("dtype") != "float8_e4m3fn":
        raise ValueError(f"Loaded scaling factors intended for KV cache dtype = {kvc.get('dtype')} rather than "
                         "float8_e4m3fn!")
    if model_type is not None and doc.get("model_type") is not None and doc["model_type"] != model_type:
        raise ValueError(f"Model type is {model_type} but loaded scaling factors belonging to different model "
                         f"type {doc['model_type']}!")
    table = {int(r): {int(l): float(s) for l, s in m.items()} for r, m in kvc["scaling_factor"].items()}
    if len(table) != tp_size or any(r not in table for r in range(tp_size)):
        raise ValueError(f"Loaded dictionary has TP size {len(table)} but LLM engine is currently running with TP "
                         f"size {tp_size}.")
    mine = table[tp_rank]
    missing = [i for i in range(num_hidden_layers) if i not in mine]
    if missing or len(mine) != num_hidden_layers:
        raise ValueError(f"KV cache scales map for TP rank {tp_rank} is malformed. Expected {num_hidden_layers} "
                         f"layers, got {len(mine)}.")
    return mine


def load_model(model_dir: str, dtype: torch.dtype = torch.float16, kv_cache_dtype: str = "auto",
               device: Union[str, torch.device] = "cuda", quantization: Optional[str] = None,
               quantization_param_path: Optional[str] = None, process_weights: bool = True):
    """config.json + quantisation config + *.safetensors -> a ready LlamaForCausalLM on ``device``
    (the role of DefaultModelLoader.load_model, model_loader/loader.py:370-420)."""
    from .model import LlamaForCausalLM, _rope_cache
    hf = read_hf_config(model_dir)
    cfg = llama_config_from_hf(hf)
    qc = resolve_quant_config(model_dir, hf, quantization, dtype)
    model = LlamaForCausalLM(cfg, qc, dtype, kv_cache_dtype)
    kv = load_llama_weights(model, iter_safetensors(model_dir), bool(hf.get("tie_word_embeddings", False)))
    if quantization_param_path is not None:
        if kv_cache_dtype == "auto":
            raise ValueError("quantization_param_path needs an fp8 KV cache (kv_cache_dtype='fp8')")
        per_layer = read_kv_cache_scales(quantization_param_path, get_tensor_model_parallel_rank(),
                                         get_tensor_model_parallel_world_size(), cfg.num_hidden_layers,
                                         hf.get("model_type"))
        for i, s in per_layer.items():
            kv[i] = {"k": s, "v": s}
    for layer, (k, v) in zip(model.layers, finalize_kv_scales(kv, cfg.num_hidden_layers, kv_cache_dtype)):
        layer.k_scale, layer.v_scale = k, v
    model.to(device)
    model.cos_sin = _rope_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta, dtype, device,
                                cfg.rope_scaling)
    if process_weights:
        model.process_weights_after_loading()
    return model
