"""Model-level adoption of the decode fast path inside the REFERENCE engine.

Through the op / quant-method seams alone a reference ``LlamaDecoderLayer`` (modeling/models/llama.py:193-270) gets one
launch per reference op -- ``bench.py``'s ``value_ops_path``.  The fused step (``forward_decode_fused``: 7 launches per
layer, fragment-major activations threaded between the kernels) needs the MODEL to call it, and the reference has a seam
for exactly that: ``ModelRegistry.register_model(arch, cls)`` (modeling/models/__init__.py:193-199) -- the out-of-tree
model hook, consulted before the built-in table (``_try_load_model_cls`` :158-160).  ``MI355XLlamaForCausalLM`` is such
a class: constructed the way ``build_model`` constructs every model (model_loader/loader.py:144-157:
``model_class(config=hf_config, cache_config=..., quant_config=..., **extra)``), fed by ``load_weights`` with the
checkpoint's (name, tensor) pairs (llama.py:480-542), called by the model runner as
``model(input_ids, positions, kv_caches, attn_metadata, intermediate_tensors)`` (worker/model_runner.py:1497-1507) and
then ``compute_logits`` / ``sample`` (llama.py:433-450).  Inside, it is this package's ``LlamaForCausalLM``: prefill runs
op by op (prefill kernels), decode batches of <= 64 rows take the fused path, both under the reference's own HIP-graph
capture (everything is launched on the current stream).

``plugin.register()`` registers it for ``LlamaForCausalLM`` and ``MistralForCausalLM`` (same decoder, llama.py:548) BY
DEFAULT since round 5 (``APHRODITE_MI355X_FUSED_MODEL=0`` opts out): one number for a maintainer, the fused one.  What this
class does not serve -- LoRA, sliding-window / biased projections / non-llama3 rope scaling
(``loader.llama_config_from_hf``), a quantization config that is not this package's, float32 -- falls back to the
reference's own class of the architecture (``unsupported_reason`` / ``__new__``), i.e. to the op-by-op path on the same
kernels; pipeline parallelism raises where it is asked for.

The attention metadata it receives is the reference's ``ROCmFlashAttentionMetadata`` (or ``MI355XAttentionMetadata``): the
fields read -- ``num_prefill_tokens``, ``num_decode_tokens``, ``slot_mapping``, ``prefill_metadata`` /
``decode_metadata``, ``seq_lens``, ``seq_lens_tensor``, ``block_tables``, ``max_decode_seq_len``, ``query_start_loc``,
``seq_start_loc``, ``context_lens_tensor``, ``max_query_len``, ``max_prefill_seq_len`` -- carry the same names and meaning
in both (attention/backends/rocm_flash_attn.py:62-152; pinned by tests/golden/attn_builder_cases)."""
from typing import Iterable, List, Optional, Tuple

import torch
from torch import nn

from . import loader as L
from .model import LlamaForCausalLM, _rope_cache


class MI355XLlamaForCausalLM(nn.Module):
    # the reference's loader and LoRA manager look these up on the class (llama.py:338-366)
    packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"], "gate_up_proj": ["gate_proj", "up_proj"]}
    supported_lora_modules: List[str] = []
    embedding_modules = {}
    embedding_padding_modules: List[str] = []
    # ``build_model`` asks ``supports_lora(model_class)`` BEFORE it constructs anything (model_loader/loader.py:115-131,
    # interfaces.py:136-142: the flag plus the four attributes above) and raises "does not support LoRA" for a class
    # without it -- so the REGISTERED subclasses carry the flag (register_with_reference) and ``lora_config`` reaches
    # ``__new__``, which hands a LoRA-enabled engine to the reference's own class.  The base class, constructed directly,
    # refuses LoRA and does not DEFINE the flag (the protocol check is hasattr, not truth).

    # the architecture this class was registered for and the reference's registry (set per registration by
    # register_with_reference): a configuration the fused step does not serve falls back to the registry's BUILT-IN class
    # of that architecture, resolved lazily (importing the reference's llama module while plugins load would be circular)
    _fallback_arch = None
    _fallback_registry = None

    @classmethod
    def unsupported_reason(cls, config, cache_config=None, quant_config=None, lora_config=None, **extra) -> Optional[str]:
        """None when the fused step serves this model; else why not (the engine then gets the reference's own model class on
        top of this package's ops and quant methods -- the op-by-op path -- instead of an exception)."""
        if lora_config is not None:
            return "LoRA adapters"
        # the engine's parallel layout: this package's layers shard by its own TP state, which must BE the
        # reference's before a fused model is built under TP > 1; pipeline stages are not served at all
        from . import distributed as D
        sizes = D.reference_parallel_sizes()
        if sizes is not None:
            tp, pp = sizes
            if pp > 1:
                return f"pipeline parallel size {pp}"
            if tp != D.get_tensor_model_parallel_world_size() and not D.adopt_reference_parallel_state():
                return (f"tensor parallel size {tp}: this package's TP state ({D.get_tensor_model_parallel_world_size()}) "
                        "could not be matched to the engine's")
        hf = config.to_dict() if hasattr(config, "to_dict") else dict(vars(config))
        try:
            L.llama_config_from_hf(hf)
        except (NotImplementedError, ValueError, KeyError) as e:
            return str(e)
        from .quantization.base_config import QuantizationConfig as OurBase
        if quant_config is not None and not isinstance(quant_config, OurBase):
            return f"quantization config {type(quant_config).__name__} is not one of this package's"
        explicit = extra.get("dtype") or getattr(extra.get("model_config"), "dtype", None)
        if isinstance(explicit, str):
            explicit = getattr(torch, explicit, None)
        if explicit is None and torch.get_default_dtype() != torch.float32:
            explicit = torch.get_default_dtype()
        if explicit is None and cls._fallback_registry is not None:
            # constructed THROUGH the registry = by the reference's loader, which builds every model under
            # set_default_torch_dtype(model_config.dtype) (model_loader/loader.py:384-390) and passes neither dtype nor
            # model_config: a float32 default there is the engine's choice (--dtype float32: its runner and KV cache are
            # float32), not "unset" -- the fused path has no float32 form
            explicit = torch.float32
        if explicit is not None and explicit not in (torch.float16, torch.bfloat16):
            return f"dtype {explicit}"
        return None

    def __new__(cls, *args, **kwargs):
        # registered by default for the dense Llama-family architectures (plugin.register): a checkpoint the fused step does
        # not serve must still load -- hand it to the reference's own class (an object that is not an instance of `cls`, so
        # this class's __init__ does not run)
        if cls._fallback_arch is not None and cls._fallback_registry is not None:
            config = kwargs.get("config", args[0] if args else None)
            rest = {k: v for k, v in kwargs.items() if k != "config"}
            try:
                reason = cls.unsupported_reason(config, **rest) if config is not None else None
            except Exception as e:      # noqa: BLE001 -- a probe must never be the reason a model does not load
                reason = f"probe failed: {e!r}"
            if reason is not None:
                fb = cls._fallback_registry._get_model(cls._fallback_arch)      # the built-in table, not the OOT one
                if fb is not None:
                    import logging
                    logging.getLogger(__name__).warning("MI355X fused model not used (%s): the reference's %s runs on the "
                                                        "MI355X ops instead", reason, getattr(fb, "__name__", fb))
                    return fb(*args, **kwargs)
        return super().__new__(cls)

    def __init__(self, config, cache_config=None, quant_config=None, lora_config=None, **extra) -> None:
        super().__init__()
        if lora_config is not None:
            raise NotImplementedError("MI355XLlamaForCausalLM: LoRA is outside the hot path (SURVEY 8 out of scope)")
        hf = config.to_dict() if hasattr(config, "to_dict") else dict(vars(config))
        self.config = config
        self.cfg = L.llama_config_from_hf(hf)
        # The engine's RESOLVED dtype wins over the checkpoint's: the reference's loader builds the model under
        # set_default_torch_dtype(model_config.dtype) (modeling/model_loader/loader.py:384-390), and that dtype -- e.g.
        # --dtype half for a GPTQ checkpoint whose config says bfloat16 -- is what the KV cache and the model runner use.
        # hf.torch_dtype only decides when the default is still torch's own float32 (standalone construction).
        # (an engine that ASKED for float32 -- an explicit dtype argument, or a model_config.dtype the caller passes
        # through -- must not silently get the checkpoint's dtype while its runner and KV cache stay float32: the fused path
        # has no float32 form, so that case is refused.)
        explicit = extra.get("dtype") or getattr(extra.get("model_config"), "dtype", None)
        if isinstance(explicit, str):
            explicit = getattr(torch, explicit)
        if explicit is not None:
            dtype = explicit
        else:
            dtype = torch.get_default_dtype()
            if dtype == torch.float32:
                dtype = hf.get("torch_dtype") or getattr(config, "torch_dtype", None) or dtype
        if isinstance(dtype, str):
            dtype = getattr(torch, dtype)
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"MI355XLlamaForCausalLM runs in float16 or bfloat16; got {dtype} (pass --dtype half / bfloat16)")
        kv_cache_dtype = getattr(cache_config, "cache_dtype", "auto") if cache_config is not None else "auto"
        self.tie_word_embeddings = bool(hf.get("tie_word_embeddings", False))
        self.kv_cache_dtype = kv_cache_dtype
        # (quant_config is OUR config class: the plugin has swapped QUANTIZATION_METHODS before the loader resolves it)
        self.inner = LlamaForCausalLM(self.cfg, quant_config, dtype, kv_cache_dtype)
        self._kv_scales = {}
        self._ready = False
        self._sampler = None

    # -- checkpoint ingestion (DefaultModelLoader calls model.load_weights(iterator), loader.py:396-408) --------------
    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> None:
        self._kv_scales = L.load_llama_weights(self.inner, weights, self.tie_word_embeddings)

    def _finish(self, device: torch.device) -> None:
        """After the loader's ``process_weights_after_loading`` pass over the modules (loader.py:402-408: every module
        with a ``quant_method`` -- our linears have one): rotary table, KV scales, the load-time relayouts of the fused
        step.  Lazily at the first forward: that is the first point where the parameters are on the device AND
        post-processed."""
        inner, cfg = self.inner, self.cfg
        for layer, (k, v) in zip(inner.layers, L.finalize_kv_scales(self._kv_scales, cfg.num_hidden_layers, self.kv_cache_dtype)):
            layer.k_scale, layer.v_scale = k, v
        if inner.cos_sin is None or inner.cos_sin.device != device:
            inner.cos_sin = _rope_cache(cfg.head_dim, cfg.max_position_embeddings, cfg.rope_theta, inner.dtype, device,
                                        cfg.rope_scaling)
        inner.process_weights_after_loading()       # no-op for the layers the loader's own pass has already processed
        for layer in inner.layers:
            # SiluAndMul in the gate_up epilogue; ONE copy of the matrix (prompt-sized batches pair the interleaved columns in
            # their SiluAndMul): the footprint bench.py measures
            layer.enable_fused_silu(32, keep_original=False)
            # ... of EVERY int4 matrix: the strip-major decode copy stays, the [K/8, N] words go (TP 1 dense layers; prompt-sized
            # GEMMs and the one-pass 33..64-row kernel address the strip-major pieces in place -- 1.5-2 % of a 33..64-row step
            # for 3.5 GB of KV blocks on Llama-3-8B; APHRO_WEIGHTS_TWO_COPIES=1 keeps both): below
            layer.enable_fp8_strips(32)                                        # FP8 checkpoints: strip-major decode copies
        for layer in inner.layers:          # (after every layer has its decode copies: see bench.build_model)
            layer.enable_one_copy()
        if device.type == "cuda":
            # the released blocks go back to the driver (the worker's memory profiling does the same before it sizes the KV
            # cache, worker/worker.py determine_num_available_blocks): left in the caching allocator they attract the step's
            # small hot buffers, scattered over 3.5 GB of address space (+ 1.4 % per step, profiles/r6_one_copy.txt)
            torch.cuda.empty_cache()
        inner.use_fused_decode = True
        self._ready = True

    # -- the model runner's calls -----------------------------------------------------------------------------------
    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor, kv_caches: List[torch.Tensor], attn_metadata,
                intermediate_tensors=None) -> torch.Tensor:
        if intermediate_tensors is not None:
            raise NotImplementedError("MI355XLlamaForCausalLM: pipeline parallelism is not implemented")
        if not self._ready:
            self._finish(input_ids.device)
        return self.inner(input_ids, positions, kv_caches, attn_metadata)

    def compute_logits(self, hidden_states: torch.Tensor, sampling_metadata) -> Optional[torch.Tensor]:
        """LogitsProcessor.forward (modeling/layers/logits_processor.py:46-77): keep the rows that are sampled from, lm_head,
        gather over TP, drop the vocabulary padding."""
        idx = getattr(sampling_metadata, "selected_token_indices", None)
        if idx is not None:
            hidden_states = hidden_states.index_select(0, idx)
        logits = self.inner.compute_logits(hidden_states)
        scale = getattr(self.config, "logit_scale", 1.0)
        return logits if scale == 1.0 else logits * scale

    def sample(self, logits: torch.Tensor, sampling_metadata):
        if self._sampler is None:
            from aphrodite.modeling.layers.sampler import Sampler      # the reference's own sampler (llama.py:430, 444-450)
            self._sampler = Sampler()
        return self._sampler(logits, sampling_metadata)

    def make_empty_intermediate_tensors(self, batch_size: int, dtype: torch.dtype, device: torch.device):
        raise NotImplementedError("MI355XLlamaForCausalLM: pipeline parallelism is not implemented")


def register_with_reference(model_registry, archs=("LlamaForCausalLM", "MistralForCausalLM")) -> None:
    """ModelRegistry.register_model for the dense Llama-family architectures (idempotent: a dict assignment).  Each
    architecture gets its own subclass carrying the reference's built-in class as the fallback for configurations the
    fused step does not serve (sliding window, projection biases, LoRA, foreign quantization configs, float32)."""
    for arch in archs:
        has_builtin = callable(getattr(model_registry, "_get_model", None))
        cls = type(f"MI355X{arch}", (MI355XLlamaForCausalLM, ),
                   {"_fallback_arch": arch if has_builtin else None, "_fallback_registry": model_registry if has_builtin else None,
                    # only a class that CAN fall back advertises LoRA to the loader's pre-construction check
                    **({"supports_lora": True} if has_builtin else {})})
        model_registry.register_model(arch, cls)
