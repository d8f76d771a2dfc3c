"""The REFERENCE-side half of ``plugin.register()`` executed against the reference's own modules (VERDICT r2 missing #2 /
next-round 5): ``aphrodite/quantization/__init__.py`` (QUANTIZATION_METHODS, get_quantization_config),
``aphrodite/quantization/kernels/__init__.py`` (_POSSIBLE_KERNELS, choose_mp_linear_kernel), ``MPLinearKernel.py`` and
``scalar_type.py`` + ``_core_ext.py`` (its own ScalarType), and ``distributed/device_communicators/custom_all_reduce.py`` are
loaded BY PATH from /root/reference under stub parent packages (the package as a whole is not importable here: loguru,
msgspec ... are missing -- the technique of tests/golden/make_golden.py); every sibling module they import is replaced by a
stub that defines only the imported names.  Then the plugin entry point runs, twice.

Also: the reference's ``_custom_ops.py`` wrappers of the SURVEY 8a ops are parsed (ast) and every ``torch.ops.<ns>.<op>(...)``
call they make is checked against the schemas ``torch_ops.register`` defines (op exists, positional arity matches); the
call list is committed as tests/golden/ref_custom_ops_calls.json so the GPU box (no /root/reference there) checks the same
list against the registered ops (tests/test_schema_gpu.py).

Needs /root/reference (this container); skipped where it is absent."""
import ast
import importlib.util
import json
import os
import sys
import types

import pytest
import torch

REF = os.environ.get("APHRODITE_REFERENCE", "/root/reference")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_custom_ops_calls.json")
HOT_OPS = ["paged_attention_v1", "paged_attention_v2", "paged_attention_rocm", "reshape_and_cache", "reshape_and_cache_flash",
           "copy_blocks", "swap_blocks", "convert_fp8", "gptq_gemm", "gptq_shuffle", "awq_gemm", "awq_dequantize",
           "scaled_fp8_quant", "cutlass_scaled_mm", "cutlass_scaled_mm_supports_fp8", "fp8_marlin_gemm", "gptq_marlin_gemm",
           "gptq_marlin_repack", "rms_norm", "fused_add_rms_norm", "silu_and_mul", "rotary_embedding",
           "advance_step_flashattn", "topk_softmax", "moe_align_block_size"]

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "aphrodite")), reason="reference checkout not present")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(name, relpath):
    pkg = relpath.endswith("__init__.py")
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath),
                                                  submodule_search_locations=[] if pkg else None)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _stub_imports_of(relpath, keep=()):
    """A stub module for every ``from aphrodite.x.y import A, B`` of the file (unless already loaded / in keep), defining
    A and B as empty classes."""
    tree = ast.parse(open(os.path.join(REF, relpath)).read())
    for node in tree.body:
        if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("aphrodite") \
                and node.module not in keep and node.module not in sys.modules:
            _stub(node.module, **{a.name: type(a.name, (), {}) for a in node.names})


@pytest.fixture()
def reference_modules():
    saved = {k: v for k, v in sys.modules.items() if k == "aphrodite" or k.startswith("aphrodite.") or k == "loguru"}
    for k in list(saved):
        del sys.modules[k]

    class _Log:
        def __getattr__(self, _):
            return lambda *a, **k: None
    _stub("loguru", logger=_Log())
    _stub("aphrodite")
    _stub("aphrodite.common")
    _stub("aphrodite.common.envs", APHRODITE_PLUGINS=None)
    _stub("aphrodite.common.utils", cuda_device_count_stateless=lambda: 0, is_hip=lambda: True)

    class _Platform:           # gfx950 reports (9, 5)
        @staticmethod
        def get_device_capability(device_id=0):
            return (9, 5)
        is_rocm = staticmethod(lambda: True)
        is_tpu = staticmethod(lambda: False)
        is_cuda_alike = staticmethod(lambda: True)
    _stub("aphrodite.platforms", current_platform=_Platform())
    # The reference's ScalarType is the C++ class of aphrodite._core_C (kernels/core/scalar_type.hpp:12-260, bound in
    # kernels/core/torch_bindings.cpp); its Python fallback in _core_ext.py is a typing mock (min / max / __str__ raise,
    # is_signed() returns None).  A stand-in with the C++ class's fields and methods takes its place -- NOT our own
    # aphrodite_engine_amd.scalar_type class, so that the kernel's by-value type check is what gets exercised -- and the
    # reference's real aphrodite/scalar_type.py builds ``scalar_types`` out of it.
    import enum

    class NanRepr(enum.Enum):
        NONE, IEEE_754, EXTD_RANGE_MAX_MIN = 0, 1, 2

    class RefScalarType:
        def __init__(self, exponent, mantissa, bias, signed, finite_values_only=False, nan_repr=1):
            self.exponent, self.mantissa, self.bias, self.signed = exponent, mantissa, bias, signed
            self._finite_values_only, self.nan_repr = finite_values_only, nan_repr
        size_bits = property(lambda self: self.exponent + self.mantissa + int(self.signed))     # scalar_type.hpp:88-90
        is_signed = lambda self: self.signed
        is_integer = lambda self: self.exponent == 0
        is_floating_point = lambda self: self.exponent > 0
        has_bias = lambda self: self.bias != 0
        int_ = classmethod(lambda cls, size_bits, bias: cls(0, size_bits - 1, bias or 0, True))   # :32-36
        uint = classmethod(lambda cls, size_bits, bias: cls(0, size_bits, bias or 0, False))      # :38-41
        float_IEEE754 = classmethod(lambda cls, e, m: cls(e, m, 0, True))
        float_ = classmethod(lambda cls, e, m, finite_values_only, nan_repr: cls(e, m, 0, True, finite_values_only, nan_repr))

        def __str__(self):                                                                        # :171-208
            if self.is_integer():
                return ("int" if self.signed else "uint") + str(self.size_bits) + (f"b{self.bias}" if self.bias else "")
            return f"float{self.size_bits}_e{self.exponent}m{self.mantissa}"
        __repr__ = __str__
    core = _stub("aphrodite._core_ext", ScalarType=RefScalarType, NanRepr=NanRepr)
    st = _load("aphrodite.scalar_type", "aphrodite/scalar_type.py")
    # quantization package: the real __init__ over stub method modules
    _stub_imports_of("aphrodite/quantization/__init__.py")
    ref_q = _load("aphrodite.quantization", "aphrodite/quantization/__init__.py")
    _stub("aphrodite.quantization.utils", replace_parameter=lambda *a, **k: None)
    _load("aphrodite.quantization.kernels.MPLinearKernel", "aphrodite/quantization/kernels/MPLinearKernel.py")
    mpk = sys.modules["aphrodite.quantization.kernels.MPLinearKernel"]

    def _other(name):          # Machete / Marlin stand-ins: present in the list, unable to implement anything here
        return type(name, (mpk.MPLinearKernel, ), {
            "get_min_capability": classmethod(lambda cls: 80),
            "can_implement": classmethod(lambda cls, c: (False, "stub")),
            "process_weights_after_loading": lambda self, layer: None,
            "apply_weights": lambda self, layer, x, bias=None: None})
    _stub("aphrodite.quantization.kernels.machete", MacheteLinearKernel=_other("MacheteLinearKernel"))
    _stub("aphrodite.quantization.kernels.marlin", MarlinLinearKernel=_other("MarlinLinearKernel"))
    ref_k = _load("aphrodite.quantization.kernels", "aphrodite/quantization/kernels/__init__.py")
    # custom all-reduce: the real class file over stubs of what it imports

    class _NoOps:
        def __getattr__(self, name):
            raise AttributeError(name)         # `ops.meta_size()` fails -> custom_ar = False, as on the reference's ROCm build
    sys.modules["aphrodite"]._custom_ops = _NoOps()
    sys.modules["aphrodite._custom_ops"] = sys.modules["aphrodite"]._custom_ops
    _stub("aphrodite.distributed")
    _stub("aphrodite.distributed.device_communicators")
    _stub("aphrodite.distributed.device_communicators.custom_all_reduce_utils", gpu_p2p_access_check=lambda a, b: True)
    _stub("aphrodite.distributed.parallel_state", in_the_same_node_as=lambda pg, source_rank=0: [True])
    ref_ca = _load("aphrodite.distributed.device_communicators.custom_all_reduce",
                   "aphrodite/distributed/device_communicators/custom_all_reduce.py")
    try:
        yield types.SimpleNamespace(q=ref_q, k=ref_k, ca=ref_ca, scalar_types=st.scalar_types, mpk=mpk, core=core)
    finally:
        for k in [k for k in sys.modules if k == "aphrodite" or k.startswith("aphrodite.") or k == "loguru"]:
            del sys.modules[k]
        sys.modules.update(saved)


@needs_ref
def test_plugin_register_against_the_reference_modules(reference_modules):
    ref = reference_modules
    from aphrodite_engine_amd import plugin
    from aphrodite_engine_amd import quantization as ours_q
    from aphrodite_engine_amd.distributed.custom_all_reduce import CustomAllreduce
    from aphrodite_engine_amd.quantization.kernels import CDNA4LinearKernel
    before = dict(ref.q.QUANTIZATION_METHODS)
    assert ref.ca.custom_ar is False                    # the ROCm build's state: the reference class disables itself
    assert not issubclass(before["gptq"], ours_q.QUANTIZATION_METHODS["gptq"])
    plugin.register()
    plugin.register()                                   # every worker process loads plugins again: idempotent
    # (1) quantization methods: ours under the reference's names, the reference's own lookup function returns them, the
    # methods we do not implement are untouched, the iteration order of the dict (config.py walks it) is unchanged
    for name, cls in ours_q.QUANTIZATION_METHODS.items():
        assert ref.q.QUANTIZATION_METHODS[name] is cls
        assert ref.q.get_quantization_config(name) is cls
    assert list(ref.q.QUANTIZATION_METHODS)[:len(before)] == list(before)
    for name in ("aqlm", "gguf", "bitsandbytes", "marlin"):
        assert ref.q.QUANTIZATION_METHODS[name] is before[name]
    with pytest.raises(ValueError):
        ref.q.get_quantization_config("no-such-method")
    # (2) mixed-precision kernels: first in the reference's list, exactly once; the reference's own chooser picks it for a
    # GPTQ uint4b8 g128 layer described with the REFERENCE's MPLinearLayerConfig and ScalarType, at capability 95 and at
    # the platform's (stubbed: (9, 5)) capability
    assert ref.k._POSSIBLE_KERNELS[0] is CDNA4LinearKernel and ref.k._POSSIBLE_KERNELS.count(CDNA4LinearKernel) == 1
    assert len(ref.k._POSSIBLE_KERNELS) == 3
    cfg = ref.mpk.MPLinearLayerConfig(full_weight_shape=(4096, 28672), partition_weight_shape=(4096, 28672),
                                      weight_type=ref.scalar_types.uint4b8, act_type=torch.float16, group_size=128,
                                      zero_points=False, has_g_idx=False)
    assert ref.k.choose_mp_linear_kernel(cfg, 95) is CDNA4LinearKernel
    assert ref.k.choose_mp_linear_kernel(cfg) is CDNA4LinearKernel
    awq_cfg = ref.mpk.MPLinearLayerConfig(full_weight_shape=(8192, 7168), partition_weight_shape=(8192, 7168),
                                          weight_type=ref.scalar_types.uint4, act_type=torch.bfloat16, group_size=128,
                                          zero_points=True, has_g_idx=False)
    assert ref.k.choose_mp_linear_kernel(awq_cfg, 95) is CDNA4LinearKernel
    # what we do not serve falls through to the reference's kernels (stubs here: "cannot implement") with its own error
    cfg8 = ref.mpk.MPLinearLayerConfig(full_weight_shape=(4096, 4096), partition_weight_shape=(4096, 4096),
                                       weight_type=ref.scalar_types.uint8b128, act_type=torch.float16, group_size=128,
                                       zero_points=False, has_g_idx=False)
    assert ref.k.choose_mp_linear_kernel(cfg8, 95) is CDNA4LinearKernel       # 8-bit symmetric (GPTQ-Marlin / wNa16 8-bit)
    cfg48 = ref.mpk.MPLinearLayerConfig(full_weight_shape=(4096, 4096), partition_weight_shape=(4096, 4096),
                                        weight_type=ref.scalar_types.uint4b8, act_type=torch.float16, group_size=48,
                                        zero_points=False, has_g_idx=False)
    with pytest.raises(ValueError, match="CDNA4LinearKernel cannot implement"):
        ref.k.choose_mp_linear_kernel(cfg48, 95)
    with pytest.raises(ValueError, match="requires capability 95"):
        ref.k.choose_mp_linear_kernel(cfg, 90)          # an MI300 (9, 4) keeps the reference's kernels
    os.environ["APHRODITE_DISABLED_KERNELS"] = "CDNA4LinearKernel"
    try:
        with pytest.raises(ValueError, match="disabled by environment variable"):
            ref.k.choose_mp_linear_kernel(cfg, 95)
    finally:
        del os.environ["APHRODITE_DISABLED_KERNELS"]
    # (3) the all-reduce class symbol GroupCoordinator instantiates (parallel_state.py:186-196)
    assert ref.ca.CustomAllreduce is CustomAllreduce


def _wrapper_calls():
    """{wrapper name: [[namespace, op, n_positional_args], ...]} for the hot-path wrappers of the reference's _custom_ops.py."""
    src = open(os.path.join(REF, "aphrodite/_custom_ops.py")).read()
    tree = ast.parse(src)
    out = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in HOT_OPS:
            calls = []
            for sub in ast.walk(node):
                if isinstance(sub, ast.Call) and isinstance(sub.func, ast.Attribute) and isinstance(sub.func.value, ast.Attribute) \
                        and isinstance(sub.func.value.value, ast.Attribute) and isinstance(sub.func.value.value.value, ast.Name) \
                        and sub.func.value.value.value.id == "torch" and sub.func.value.value.attr == "ops":
                    assert not sub.keywords, f"{node.name}: keyword arguments in a torch.ops call"
                    calls.append([sub.func.value.attr, sub.func.attr, len(sub.args)])
            out[node.name] = calls
    return out


@needs_ref
def test_reference_custom_ops_wrappers_bind_to_our_schemas():
    """Every torch.ops call the reference's own wrappers make for the SURVEY 8a ops lands on a registered op whose schema
    takes exactly that many arguments; the call list is what tests/golden/ref_custom_ops_calls.json holds."""
    from aphrodite_engine_amd import torch_ops
    calls = _wrapper_calls()
    assert set(calls) >= {"paged_attention_v1", "paged_attention_rocm", "reshape_and_cache", "gptq_gemm", "gptq_shuffle",
                          "awq_gemm", "scaled_fp8_quant", "cutlass_scaled_mm"}
    committed = json.load(open(GOLDEN))
    assert committed == calls, "tests/golden/ref_custom_ops_calls.json is stale: regenerate with `python tests/test_reference_binding_cpu.py`"
    schemas = torch_ops.schema_arity()
    missing = []
    for wrapper, cl in calls.items():
        for ns, op, nargs in cl:
            key = (ns, op)
            if key not in schemas:
                missing.append(f"{wrapper}: torch.ops.{ns}.{op} is not registered")
            elif schemas[key] != nargs:
                missing.append(f"{wrapper}: torch.ops.{ns}.{op} takes {schemas[key]} arguments, the wrapper passes {nargs}")
    assert not missing, "\n".join(missing)


if __name__ == "__main__":   # regenerate the committed call list (build container only)
    json.dump(_wrapper_calls(), open(GOLDEN, "w"), indent=1, sort_keys=True)
    print("wrote", GOLDEN)


@needs_ref
def test_fused_model_registers_through_the_reference_model_registry(monkeypatch):
    """plugin.register() (default; APHRODITE_MI355X_FUSED_MODEL=0 opts out) hands MI355XLlamaForCausalLM to the reference's OWN ModelRegistry
    (modeling/models/__init__.py loaded by path), whose lookup then resolves the Llama architectures to it; the class is
    constructed with the keyword arguments ``build_model`` passes (model_loader/loader.py:144-157) and exposes the methods
    the model runner calls."""
    saved = {k: v for k, v in sys.modules.items() if k == "aphrodite" or k.startswith("aphrodite.") or k == "loguru"}
    for k in list(saved):
        del sys.modules[k]
    try:
        class _Log:
            def __getattr__(self, _):
                return lambda *a, **k: None
        _stub("loguru", logger=_Log())
        _stub("aphrodite")
        _stub("aphrodite.common")
        _stub("aphrodite.common.utils", is_hip=lambda: True)
        _stub("aphrodite.modeling")
        # quantization seams must exist for register() to get as far as the model hook
        _stub("aphrodite.quantization", QUANTIZATION_METHODS={})
        _stub("aphrodite.quantization.kernels", _POSSIBLE_KERNELS=[])
        reg = _load("aphrodite.modeling.models", "aphrodite/modeling/models/__init__.py")
        from aphrodite_engine_amd import plugin
        from aphrodite_engine_amd.reference_model import MI355XLlamaForCausalLM
        builtin = reg._MODELS["LlamaForCausalLM"]
        monkeypatch.setenv("APHRODITE_MI355X_FUSED_MODEL", "0")
        plugin.register()                                                   # opted out: the registry keeps its own
        assert "LlamaForCausalLM" not in reg._OOT_MODELS
        monkeypatch.delenv("APHRODITE_MI355X_FUSED_MODEL")
        # a stand-in for the reference's built-in class (its real module needs the whole engine): the fallback target
        class RefLlama:
            def __init__(self, *a, **k):
                self.args, self.kwargs = a, k
        monkeypatch.setattr(reg.ModelRegistry, "_get_model", staticmethod(lambda arch: RefLlama), raising=False)
        plugin.register()
        plugin.register()
        cls = reg.ModelRegistry._try_load_model_cls("LlamaForCausalLM")
        assert issubclass(cls, MI355XLlamaForCausalLM) and issubclass(reg.ModelRegistry._try_load_model_cls("MistralForCausalLM"), MI355XLlamaForCausalLM)
        assert reg._MODELS["LlamaForCausalLM"] == builtin                   # the built-in table is untouched
        assert "LlamaForCausalLM" in reg.ModelRegistry.get_supported_archs()
        # constructed the way build_model does, from a HF-style config object
        from aphrodite_engine_amd.quantization.gptq import GPTQConfig
        hf = types.SimpleNamespace(hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=2, vocab_size=1024, rms_norm_eps=1e-5, rope_theta=10000.0,
                                   max_position_embeddings=2048, torch_dtype=torch.float16, tie_word_embeddings=False)
        cache = types.SimpleNamespace(cache_dtype="auto")
        m = MI355XLlamaForCausalLM(config=hf, cache_config=cache, quant_config=GPTQConfig(4, 128, False))
        for name in ("forward", "load_weights", "compute_logits", "sample"):
            assert callable(getattr(m, name))
        assert m.inner.layers[0].qkv_proj.qweight.shape == (512 // 8, (4 + 2 * 2) * 128)
        with pytest.raises(NotImplementedError):
            MI355XLlamaForCausalLM(config=hf, cache_config=cache, quant_config=None, lora_config=object())
        # ... while the REGISTERED class hands what the fused step does not serve to the reference's built-in class.
        # The reference's loader constructs every model under set_default_torch_dtype(model_config.dtype)
        # (model_loader/loader.py:384-390): --dtype half here
        torch.set_default_dtype(torch.float16)
        assert isinstance(cls(config=hf, cache_config=cache, quant_config=GPTQConfig(4, 128, False)), MI355XLlamaForCausalLM)
        for bad_kw, bad_hf in ((dict(lora_config=object()), hf),
                               ({}, types.SimpleNamespace(**{**vars(hf), "sliding_window": 4096})),
                               ({}, types.SimpleNamespace(**{**vars(hf), "hidden_act": "gelu"})),
                               (dict(quant_config=object()), hf)):
            kw = dict(cache_config=cache, quant_config=GPTQConfig(4, 128, False))
            kw.update(bad_kw)
            ref = cls(config=bad_hf, **kw)
            assert isinstance(ref, RefLlama) and ref.kwargs["config"] is bad_hf
        # projection biases (config.attention_bias, models/llama.py:206-211) are served by this class (op-by-op layers)
        biased = cls(config=types.SimpleNamespace(**{**vars(hf), "attention_bias": True}), cache_config=cache,
                     quant_config=GPTQConfig(4, 128, False))
        assert isinstance(biased, MI355XLlamaForCausalLM) and biased.inner.layers[0].qkv_proj.bias is not None
        assert biased.inner.layers[0].has_bias and biased.inner.layers[0].down_proj.bias is None
        # an engine run with --dtype float32 builds the model under a float32 default and passes no dtype:
        # through the registry that is the engine's choice, not "unset" -- the reference's class gets it
        torch.set_default_dtype(torch.float32)
        assert isinstance(cls(config=hf, cache_config=cache, quant_config=GPTQConfig(4, 128, False)), RefLlama)
        torch.set_default_dtype(torch.float16)
        # build_model asks supports_lora(model_class) BEFORE constructing (loader.py:115-131) -- with the
        # reference's own interfaces.py: the registered class passes the check, so lora_config reaches __new__ and falls back
        _stub("aphrodite.common.config", LoRAConfig=type("LoRAConfig", (), {}), MultiModalConfig=type("MultiModalConfig", (), {}),
              SchedulerConfig=type("SchedulerConfig", (), {}))
        itf = _load("aphrodite.modeling.models.interfaces", "aphrodite/modeling/models/interfaces.py")
        assert itf.supports_lora(cls) and not itf.supports_lora(MI355XLlamaForCausalLM)
        lora = sys.modules["aphrodite.common.config"].LoRAConfig()
        assert isinstance(cls(config=hf, cache_config=cache, quant_config=GPTQConfig(4, 128, False), lora_config=lora), RefLlama)
        assert isinstance(cls(config=hf, cache_config=cache, quant_config=GPTQConfig(4, 128, False), lora_config=None),
                          MI355XLlamaForCausalLM)
        # under a TP > 1 engine the fused model takes group, rank and size from the reference's
        # GroupCoordinator (parallel_state.py:875-889) -- every rank builds ITS shard; pipeline stages fall back
        from aphrodite_engine_amd import distributed as D
        grp = types.SimpleNamespace(world_size=2, rank_in_group=1, device_group=object(), ca_comm=None)
        pp = types.SimpleNamespace(world_size=1)
        _stub("aphrodite.distributed")
        ps = _stub("aphrodite.distributed.parallel_state", get_tp_group=lambda: grp, get_pp_group=lambda: pp)
        sys.modules["aphrodite.distributed"].parallel_state = ps
        try:
            assert D.reference_parallel_sizes() == (2, 1)
            m2 = cls(config=hf, cache_config=cache, quant_config=GPTQConfig(4, 128, False))
            assert isinstance(m2, MI355XLlamaForCausalLM)
            assert (D.get_tensor_model_parallel_world_size(), D.get_tensor_model_parallel_rank()) == (2, 1)
            assert m2.inner.layers[0].qkv_proj.qweight.shape == (512 // 8, (4 + 2 * 2) * 128 // 2)      # this rank's heads
            assert m2.inner.layers[0].num_kv_heads == 1
            pp.world_size = 2
            assert isinstance(cls(config=hf, cache_config=cache, quant_config=GPTQConfig(4, 128, False)), RefLlama)
            # a TP state of this package that is something ELSE is never overwritten: fall back instead
            pp.world_size, grp.world_size = 1, 4
            assert isinstance(cls(config=hf, cache_config=cache, quant_config=GPTQConfig(4, 128, False)), RefLlama)
            assert D.get_tensor_model_parallel_world_size() == 2
        finally:
            D.destroy_tensor_parallel()
    finally:
        torch.set_default_dtype(torch.float32)
        for k in [k for k in sys.modules if k == "aphrodite" or k.startswith("aphrodite.") or k == "loguru"]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_fused_moe_methods_accept_the_reference_layers_calls():
    """The reference's FusedMoE drives its quant method with these keywords (modeling/layers/fused_moe/layer.py:211-217
    create_weights, :437-446 apply) and owns no ``tp_rank`` / ``orig_dtype``: every MoE method the plugin's configs return
    must accept exactly that -- a TypeError here is a broken drop-in."""
    import inspect
    from aphrodite_engine_amd.moe import CompressedTensorsMoEMethod, Fp8MoEMethod, Wna16MoEMethod
    apply_kw = ("layer", "x", "router_logits", "top_k", "renormalize", "use_grouped_topk", "topk_group", "num_expert_group",
                "custom_routing_function")
    create_kw = ("layer", "num_experts", "hidden_size", "intermediate_size", "params_dtype", "weight_loader")
    for cls in (Wna16MoEMethod, Fp8MoEMethod, CompressedTensorsMoEMethod):
        sig = inspect.signature(cls.apply)
        assert all(k in sig.parameters for k in apply_kw), (cls.__name__, list(sig.parameters))
        sig = inspect.signature(cls.create_weights)
        has_var_kw = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in sig.parameters.values())
        assert all(k in sig.parameters or has_var_kw for k in create_kw), cls.__name__
    # a bare module standing in for the reference's layer: no tp_rank, no orig_dtype
    import torch
    from aphrodite_engine_amd.quantization.fp8 import Fp8Config

    class Layer(torch.nn.Module):
        tp_size = 1
        intermediate_size_per_partition = 256
    layer = Layer()
    m = Fp8Config(True, "dynamic").get_quant_method(type("FusedMoE", (torch.nn.Module, ), {})(), "x.experts")
    m.create_weights(layer=layer, num_experts=2, hidden_size=256, intermediate_size=256, params_dtype=torch.float16,
                     weight_loader=lambda *a, **k: None)
    assert layer.orig_dtype == torch.float16 and layer.w13_weight.shape == (2, 512, 256)
    # per-tensor scales are recognised by the reference's loader through this attribute (fused_moe/layer.py:340-361)
    assert layer.w13_weight_scale.quant_method == "tensor" and layer.w2_weight_scale.quant_method == "tensor"


@needs_ref
def test_kv_cache_method_satisfies_the_reference_attention_layer(reference_modules):
    """attention/layer.py:61-64: the reference's Attention asserts ``isinstance(quant_method, BaseKVCacheMethod)`` against
    ITS OWN class for whatever the (plugin-registered) config returns.  With the reference's real kv_cache.py /
    base_config.py loaded, our FP8 and compressed-tensors configs must return something that passes, and behaves like
    ours (k_scale / v_scale parameters -> python floats)."""
    import torch
    sys.modules["aphrodite.common.utils"].print_warning_once = lambda *a, **k: None
    _load("aphrodite.quantization.base_config", "aphrodite/quantization/base_config.py")
    ref_kv = _load("aphrodite.quantization.kv_cache", "aphrodite/quantization/kv_cache.py")
    from aphrodite_engine_amd.quantization.compressed_tensors import CompressedTensorsConfig
    from aphrodite_engine_amd.quantization.fp8 import Fp8Config
    from aphrodite_engine_amd.quantization.kv_cache import BaseKVCacheMethod as OursKV
    ct = CompressedTensorsConfig.from_config({"format": "float-quantized", "config_groups": {
        "g": {"targets": ["Linear"], "weights": {"num_bits": 8, "type": "float", "strategy": "channel"},
              "input_activations": {"num_bits": 8, "type": "float", "strategy": "token", "dynamic": True}}}})
    for cfg in (Fp8Config(True, "dynamic"), ct):
        attn = type("Attention", (torch.nn.Module, ), {})()
        attn.kv_cache_dtype = "fp8"
        m = cfg.get_quant_method(attn, prefix="model.layers.0.self_attn.attn")
        assert isinstance(m, ref_kv.BaseKVCacheMethod) and isinstance(m, OursKV)
        assert type(m).create_weights is OursKV.create_weights            # ours first in the MRO
        m.create_weights(attn)
        attn.k_scale.data.fill_(0.5)
        attn.v_scale.data.fill_(0.25)
        m.process_weights_after_loading(attn)
        assert (attn._k_scale, attn._v_scale) == (0.5, 0.25) and not hasattr(attn, "k_scale")


@needs_ref
@pytest.mark.parametrize("fmt", ["ct-w4a16-group", "ct-w4a16-channel", "fp8"])
def test_reference_fused_moe_layer_loads_through_our_expert_methods(reference_modules, fmt):
    """The reference's REAL FusedMoE (modeling/layers/fused_moe/layer.py, loaded by path) built with OUR config: its
    __init__ asks our config for the method and calls create_weights with its keywords, its weight_loader (is_transposed,
    quant_method = group / channel / tensor, weight_shape, TP cut of rank 1 of 2) fills our parameters.  The same tensors
    through our own FusedMoE must give identical parameters -- the two loaders are independent implementations."""
    import numpy as np
    import torch
    rank, world = 1, 2
    sys.modules["aphrodite.distributed"].get_tensor_model_parallel_rank = lambda: rank
    sys.modules["aphrodite.distributed"].get_tensor_model_parallel_world_size = lambda: world
    sys.modules["aphrodite.distributed"].tensor_model_parallel_all_reduce = lambda x: x
    _load("aphrodite.quantization.base_config", "aphrodite/quantization/base_config.py")
    _stub("aphrodite.modeling")
    _stub("aphrodite.modeling._custom_op", CustomOp=type("CustomOp", (torch.nn.Module, ), {}))
    from aphrodite_engine_amd.quantization.base_config import set_weight_attrs
    _stub("aphrodite.modeling.utils", set_weight_attrs=set_weight_attrs)
    _stub("aphrodite.modeling.layers")
    _stub("aphrodite.modeling.layers.fused_moe")
    ref_layer = _load("aphrodite.modeling.layers.fused_moe.layer", "aphrodite/modeling/layers/fused_moe/layer.py")
    from aphrodite_engine_amd.distributed import simulated_tensor_parallel
    from aphrodite_engine_amd.moe import FusedMoE as OurFusedMoE
    from aphrodite_engine_amd.quantization.compressed_tensors import CompressedTensorsConfig
    from aphrodite_engine_amd.quantization.fp8 import Fp8Config
    e, h, inter = 2, 256, 512
    rng = np.random.default_rng(3)
    if fmt == "fp8":
        qc = Fp8Config(True, "static")
    else:
        w = {"num_bits": 4, "type": "int", "symmetric": True, "strategy": fmt.rsplit("-", 1)[1]}
        if w["strategy"] == "group":
            w["group_size"] = 128
        qc = CompressedTensorsConfig.from_config({"format": "pack-quantized", "config_groups": {
            "g": {"targets": ["Linear"], "weights": w, "input_activations": None}}})
    ref_moe = ref_layer.FusedMoE(e, 2, h, inter, params_dtype=torch.float16, quant_config=qc, prefix="m.experts")
    with simulated_tensor_parallel(rank, world):
        our_moe = OurFusedMoE(e, 2, h, inter, params_dtype=torch.float16, quant_config=qc, prefix="m.experts")
    assert type(ref_moe.quant_method) is type(our_moe.quant_method)
    for x in range(e):
        for shard, n, k in (("w1", inter, h), ("w3", inter, h), ("w2", h, inter)):
            fused = "w13_" if shard != "w2" else "w2_"
            if fmt == "fp8":
                tensors = {"weight": torch.from_numpy(rng.integers(0, 120, (n, k)).astype(np.uint8)).view(torch.float8_e4m3fn),
                           "weight_scale": torch.tensor(0.01 * (1 + x) + 0.001 * len(shard)),
                           "input_scale": torch.tensor(0.5 + 0.1 * x)}
            else:
                g = 1 if fmt.endswith("channel") else k // 128
                tensors = {"weight_packed": torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (n, k // 8)).astype(np.int32)),
                           "weight_scale": torch.from_numpy(rng.random((n, g)).astype(np.float16)),
                           "weight_shape": torch.tensor([n, k])}
            for suffix, t_ in tensors.items():
                name = f"m.experts.{x}.{shard}.{suffix}"
                for moe_ in (ref_moe, our_moe):
                    moe_.weight_loader(getattr(moe_, fused + suffix), t_.clone(), name, shard, x)
    names = [n for n, _ in our_moe.named_parameters()]
    assert sorted(names) == sorted(n for n, _ in ref_moe.named_parameters())
    for n in names:
        a, b = getattr(ref_moe, n).data, getattr(our_moe, n).data
        assert a.shape == b.shape and a.dtype == b.dtype, n
        assert torch.equal(a.view(torch.uint8) if a.dtype == torch.float8_e4m3fn else a,
                           b.view(torch.uint8) if b.dtype == torch.float8_e4m3fn else b), n


@needs_ref
@pytest.mark.parametrize("fmt,static", [("gptq", False), ("awq", False), ("fp8", True), ("ct-fp8-channel", True),
                                        ("ct-fp8-tensor", False), ("ct-w4a16", False), ("ct-w8a16i", False), ("ct-w8a16", False)])
@pytest.mark.parametrize("rank,world", [(1, 2), (3, 4)])
def test_reference_linear_layers_load_through_our_linear_methods(reference_modules, tmp_path, fmt, static, rank, world):
    """The reference's REAL QKVParallelLinear / MergedColumnParallelLinear / RowParallelLinear (modeling/layers/linear.py
    and modeling/parameter.py loaded by path) built with OUR configs: their __init__ asks our config for the method, calls
    create_weights with their keywords and -- our class names being outside WEIGHT_LOADER_V2_SUPPORTED (:28-44) -- hand it
    their v1 weight_loader, which then cuts a synthetic checkpoint into our parameters using the metadata we put on them
    (input_dim / output_dim / packed_dim / pack_factor / needs_scalar_to_array).  The same checkpoint through OUR model
    and loader (an independent implementation: loader.py's shard plans) must leave identical parameters, at TP 2 and at
    TP 4 (2 KV heads: replicated)."""
    import torch
    from safetensors import safe_open
    from aphrodite_engine_amd import loader as L
    from aphrodite_engine_amd import model as M
    from aphrodite_engine_amd.distributed import simulated_tensor_parallel
    from tests import ckpt_util as CU
    dist = sys.modules["aphrodite.distributed"]
    dist.get_tensor_model_parallel_rank = lambda: rank
    dist.get_tensor_model_parallel_world_size = lambda: world
    dist.divide = lambda a, b: a // b if a % b == 0 else (_ for _ in ()).throw(AssertionError((a, b)))
    dist.get_current_tp_rank_partition_size = lambda total, r=None, w=None, multiple_of=1: total // (w or world)
    dist.get_current_tp_rank_partition_offset = lambda total, r=None, w=None, multiple_of=1: \
        (total // (w or world)) * (rank if r is None else r)
    for name in ("split_tensor_along_last_dim", "tensor_model_parallel_all_gather", "tensor_model_parallel_all_reduce"):
        setattr(dist, name, lambda x, *a, **k: x)
    _load("aphrodite.quantization.base_config", "aphrodite/quantization/base_config.py")
    from aphrodite_engine_amd.quantization.base_config import set_weight_attrs
    _stub("aphrodite.modeling")
    _stub("aphrodite.modeling.utils", set_weight_attrs=set_weight_attrs)
    _load("aphrodite.modeling.parameter", "aphrodite/modeling/parameter.py")
    _stub("aphrodite.modeling.layers")
    ref_lin = _load("aphrodite.modeling.layers.linear", "aphrodite/modeling/layers/linear.py")

    cfg = M.TINY
    CU.write_checkpoint(str(tmp_path), cfg, fmt, seed=17, fp8_static=static)
    with simulated_tensor_parallel(rank, world):
        ours = L.load_model(str(tmp_path), dtype=torch.float16, device="cpu", process_weights=False)
        qc = L.resolve_quant_config(str(tmp_path), L.read_hf_config(str(tmp_path)))
    hd = cfg.hidden_size // cfg.num_attention_heads
    kw = dict(bias=False, params_dtype=torch.float16, quant_config=qc)
    layers = []
    for li in range(cfg.num_hidden_layers):
        p = f"model.layers.{li}."
        layers.append({
            "qkv_proj": ref_lin.QKVParallelLinear(cfg.hidden_size, hd, cfg.num_attention_heads, cfg.num_key_value_heads,
                                                  prefix=p + "self_attn.qkv_proj", **kw),
            "o_proj": ref_lin.RowParallelLinear(cfg.num_attention_heads * hd, cfg.hidden_size,
                                                prefix=p + "self_attn.o_proj", **kw),
            "gate_up_proj": ref_lin.MergedColumnParallelLinear(cfg.hidden_size, [cfg.intermediate_size] * 2,
                                                               prefix=p + "mlp.gate_up_proj", **kw),
            "down_proj": ref_lin.RowParallelLinear(cfg.intermediate_size, cfg.hidden_size, prefix=p + "mlp.down_proj", **kw)})
    stacked = {"q_proj": ("qkv_proj", "q"), "k_proj": ("qkv_proj", "k"), "v_proj": ("qkv_proj", "v"),
               "gate_proj": ("gate_up_proj", 0), "up_proj": ("gate_up_proj", 1)}       # llama.py:480-490
    fed = 0
    for fname in sorted(os.listdir(tmp_path)):
        if not fname.endswith(".safetensors"):
            continue
        with safe_open(os.path.join(tmp_path, fname), "pt") as f:
            for name in f.keys():
                parts = name.split(".")
                if len(parts) != 6 or parts[3] not in ("self_attn", "mlp") or not parts[4].endswith("_proj"):
                    continue
                li, proj, attr = int(parts[2]), parts[4], parts[5]
                mod, shard = stacked.get(proj, (proj, None))
                param = getattr(layers[li][mod], attr, None)
                if param is None:
                    assert attr == "bias", name                  # AutoGPTQ's extra bias tensors (llama.py:516-518)
                    continue
                args = (param, f.get_tensor(name)) + ((shard, ) if shard is not None else ())
                param.weight_loader(*args)                       # exactly how the reference's load_weights calls it
                fed += 1
    assert fed > 0
    for li, group in enumerate(layers):
        for mod, ref_layer in group.items():
            ours_layer = getattr(ours.layers[li], mod)
            assert type(ref_layer.quant_method) is type(ours_layer.quant_method), (mod, type(ref_layer.quant_method))
            ref_params, our_params = dict(ref_layer.named_parameters()), dict(ours_layer.named_parameters())
            assert sorted(ref_params) == sorted(our_params), (mod, sorted(ref_params), sorted(our_params))
            for n, a in ref_params.items():
                b = our_params[n]
                assert a.shape == b.shape and a.dtype == b.dtype, (mod, n, a.shape, b.shape)
                av = a.data.view(torch.uint8) if a.dtype == torch.float8_e4m3fn else a.data
                bv = b.data.view(torch.uint8) if b.dtype == torch.float8_e4m3fn else b.data
                assert torch.equal(av, bv), (fmt, mod, n)


def test_moe_apply_runs_with_the_reference_layers_keywords(monkeypatch):
    """FusedMoE.forward's call (fused_moe/layer.py:437-446), keyword for keyword, through both apply bodies with the GPU
    pieces stubbed: the routing helper must receive the model's custom routing function, the expert kernels the aligned
    lists."""
    import types
    import torch
    from aphrodite_engine_amd import moe
    seen = {}

    def fake_route(x, logits, top_k, renorm, num_experts, want_inverse=False, custom_routing_function=None):
        seen["route"] = (top_k, renorm, num_experts, want_inverse, custom_routing_function)
        return ("w", "ids", "sorted", "experts", "post", "inv" if want_inverse else None)
    monkeypatch.setattr(moe, "route_and_align", fake_route)
    monkeypatch.setattr(moe, "fused_wna16_moe", lambda *a, **k: seen.update(int4=(a, k)) or "int4-out")
    monkeypatch.setattr(moe, "fused_fp8_moe", lambda *a, **k: seen.update(fp8=(a, k)) or "fp8-out")
    routing = lambda **k: None                                                    # noqa: E731
    x, logits = torch.zeros(3, 8), torch.zeros(3, 4)
    kw = dict(x=x, router_logits=logits, top_k=2, renormalize=True, use_grouped_topk=False, topk_group=None,
              num_expert_group=None, custom_routing_function=routing)
    layer = types.SimpleNamespace(experts_packed=types.SimpleNamespace(num_experts=4))
    assert moe.Wna16MoEMethod("gptq", 128).apply(layer=layer, **kw) == "int4-out"
    assert seen["route"] == (2, True, 4, True, routing)
    assert seen["int4"][1]["aligned"] == ("sorted", "experts", "post", "inv") and seen["int4"][1]["topk_ids"] == "ids"
    layer = types.SimpleNamespace(w13_weight=torch.zeros(4, 2, 2), w2_weight=torch.zeros(4, 2, 2), w13_weight_scale=None,
                                  w2_weight_scale=None, w13_input_scale=None, w2_input_scale=None)
    assert moe.Fp8MoEMethod(types.SimpleNamespace()).apply(layer=layer, **kw) == "fp8-out"
    assert seen["route"] == (2, True, 4, False, routing) and seen["fp8"][1]["aligned"] == ("sorted", "experts", "post")
    with pytest.raises(NotImplementedError):
        moe.Wna16MoEMethod("gptq", 128).apply(layer=layer, **{**kw, "use_grouped_topk": True})


@needs_ref
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8"])
def test_reference_attention_layer_over_our_backend_and_kv_method(reference_modules, kv_cache_dtype):
    """The reference's REAL Attention layer (attention/layer.py, loaded by path): built with our FP8 config and with
    ``get_attn_backend`` handing out MI355XAttentionBackend (what the selector's ROCm branch is pointed at, INTEGRATION
    §1.4).  Its __init__ asserts the KV method's type, lets it register k_scale / v_scale, and constructs our impl with
    its nine positional arguments; after "loading" the scales our method turns them into the floats the layer passes to
    ``impl.forward(..., k_scale, v_scale, attn_type=)``."""
    import enum
    import types
    import torch
    from aphrodite_engine_amd.attention.backend import MI355XAttentionBackend, MI355XAttentionImpl
    from aphrodite_engine_amd.quantization.fp8 import Fp8Config
    sys.modules["aphrodite.common.utils"].print_warning_once = lambda *a, **k: None
    _load("aphrodite.quantization.base_config", "aphrodite/quantization/base_config.py")
    _load("aphrodite.quantization.kv_cache", "aphrodite/quantization/kv_cache.py")
    AttentionType = enum.Enum("AttentionType", ["DECODER", "ENCODER", "ENCODER_DECODER"])
    _stub("aphrodite.attention", AttentionMetadata=object, AttentionType=AttentionType)
    picked = {}

    def get_attn_backend(head_size, sliding_window, dtype, kv_dtype, block_size, is_attention_free, is_blocksparse=False):
        picked.update(head_size=head_size, kv=kv_dtype, block=block_size)
        return MI355XAttentionBackend
    _stub("aphrodite.attention.selector", get_attn_backend=get_attn_backend)
    _stub("aphrodite.common.config", CacheConfig=object)
    ref_attn = _load("aphrodite.attention.layer", "aphrodite/attention/layer.py")
    cache_config = types.SimpleNamespace(cache_dtype=kv_cache_dtype, block_size=16, sliding_window=None,
                                         is_attention_free=False)
    layer = ref_attn.Attention(32, 128, 128 ** -0.5, num_kv_heads=8, cache_config=cache_config,
                               quant_config=Fp8Config(True, "dynamic"), prefix="model.layers.0.self_attn.attn")
    assert picked == {"head_size": 128, "kv": kv_cache_dtype, "block": 16}
    assert isinstance(layer.impl, MI355XAttentionImpl) and (layer.impl.num_heads, layer.impl.num_kv_heads) == (32, 8)
    assert layer.impl.kv_cache_dtype == kv_cache_dtype
    assert isinstance(layer.k_scale, torch.nn.Parameter) and isinstance(layer.v_scale, torch.nn.Parameter)
    layer.k_scale.data.fill_(0.02)                    # what default_weight_loader leaves from self_attn.k_scale
    layer.v_scale.data.fill_(0.03)
    layer.quant_method.process_weights_after_loading(layer)
    if kv_cache_dtype == "fp8":
        assert (layer._k_scale, layer._v_scale) == (pytest.approx(0.02), pytest.approx(0.03))
    else:
        assert (layer._k_scale, layer._v_scale) == (1.0, 1.0)      # ignored without an fp8 cache (kv_cache.py:37-75)
    assert not hasattr(layer, "k_scale")
    # the layer's forward hands everything to impl.forward, attn_type by keyword
    got = {}
    layer.impl.forward = lambda *a, **k: got.update(a=a, k=k) or "out"
    q = torch.zeros(1, 32 * 128)
    assert layer.forward(q, q[:, :1024], q[:, :1024], None, None) == "out"
    assert got["a"][5:] == (layer._k_scale, layer._v_scale) and got["k"] == {"attn_type": AttentionType.DECODER}
    # a Mistral-style engine config: cache_config.sliding_window reaches our impl the way ROCmFlashAttentionImpl keeps it
    # (attention/layer.py:44-47 -> rocm_flash_attn.py:321-322)
    windowed = ref_attn.Attention(32, 128, 128 ** -0.5, num_kv_heads=8,
                                  cache_config=types.SimpleNamespace(cache_dtype=kv_cache_dtype, block_size=16, sliding_window=4096,
                                                                     is_attention_free=False),
                                  quant_config=None, prefix="model.layers.1.self_attn.attn")
    assert isinstance(windowed.impl, MI355XAttentionImpl) and windowed.impl.sliding_window == (4096, 4096)
    assert layer.impl.sliding_window == (-1, -1)
