"""``aphrodite.general_plugins`` entry point (reference loader:
aphrodite/plugins/__init__.py:8-31, run in every worker process, must be
idempotent).  Install with

    [project.entry-points."aphrodite.general_plugins"]
    mi355x = "aphrodite_engine_amd.plugin:register"

It (1) registers the torch.library ops so ``aphrodite._custom_ops`` resolves to the
MI355X kernels, (2) swaps the quantization methods and prepends the CDNA4 mixed
precision kernel, (3) leaves attention to the op level: the reference's
``ROCM_FLASH`` backend (selector.py:210-220) calls ``ops.paged_attention_rocm`` /
``reshape_and_cache``, which are now ours, and (4) puts the xGMI peer-access
all-reduce where the reference's ROCm build has none: ``CustomAllreduce`` is swapped at the class
level (distributed/device_communicators/custom_all_reduce.py:41 -- the ``_C_custom_ar`` ops are
compiled out on ROCm, torch_bindings.cpp:506, so ``custom_ar`` is False there and the class
disables itself), because its signal memory must be an uncached allocation a torch tensor cannot be, and (5) registers
``MI355XLlamaForCausalLM`` for the dense Llama-family architectures through ``ModelRegistry.register_model`` so that the
decode step runs the fused fast path (reference_model.py; modeling/models/__init__.py:193-199;
``APHRODITE_MI355X_FUSED_MODEL=0`` opts out; configurations the fused step does not serve fall back to the reference's
own class).
"""


def register() -> None:
    from . import torch_ops
    from .switches import switch
    torch_ops.register()
    try:
        import aphrodite.quantization as ref_q
        import aphrodite.quantization.kernels as ref_k
    except Exception:  # the reference is not installed: ops only
        return
    from .quantization import register_with_reference as reg_methods
    from .quantization.kernels import register_with_reference as reg_kernels
    reg_methods(ref_q.QUANTIZATION_METHODS)
    reg_kernels(ref_k._POSSIBLE_KERNELS)
    import os
    if (switch("APHRODITE_MI355X_FUSED_MODEL") or "1") != "0":
        # model-level adoption of the fused decode step through the reference's out-of-tree model hook (reference_model.py):
        # ON by default since round 5 -- what the fused step does not serve falls back to the reference's own class
        try:
            from aphrodite.modeling.models import ModelRegistry
            from .reference_model import register_with_reference as reg_model
            reg_model(ModelRegistry)
        except Exception as exc:    # opted in and did not happen: say so, the op-level path still works
            import logging
            logging.getLogger(__name__).warning("the fused model was not registered (%r): the op-by-op path stays", exc)
    try:   # GroupCoordinator builds ``ca_comm = CustomAllreduce(group=cpu_group, device=...)`` (parallel_state.py:186-196)
        import aphrodite.distributed.device_communicators.custom_all_reduce as ref_ca
        from .distributed.custom_all_reduce import CustomAllreduce
        ref_ca.CustomAllreduce = CustomAllreduce
    except Exception as exc:
        import logging
        logging.getLogger(__name__).warning("custom all-reduce not swapped in (%r): the reference's own communicator stays", exc)
