"""``aphrodite.general_plugins`` entry point (reference loader:
aphrodite/plugins/__init__.py:8-31, run in every worker process, must be
idempotent).  Install with

    [project.entry-points."aphrodite.general_plugins"]
    mi355x = "aphrodite_engine_amd.plugin:register"

It (1) registers the torch.library ops so ``aphrodite._custom_ops`` resolves to the
MI355X kernels, (2) swaps the quantization methods and prepends the CDNA4 mixed
precision kernel, (3) leaves attention to the op level: the reference's
``ROCM_FLASH`` backend (selector.py:210-220) calls ``ops.paged_attention_rocm`` /
``reshape_and_cache``, which are now ours.
"""


def register() -> None:
    from . import torch_ops
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
