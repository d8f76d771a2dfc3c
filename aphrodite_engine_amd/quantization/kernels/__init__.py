"""Mixed-precision linear kernel registry -- mirror of
aphrodite/quantization/kernels/__init__.py:11-74 with the CDNA4 kernel in the
slot Machete/Marlin occupy on NVIDIA."""
import os
from typing import List, Optional, Type

from ...switches import switch
from .MPLinearKernel import MPLinearKernel, MPLinearLayerConfig
from .cdna4 import CDNA4LinearKernel

_POSSIBLE_KERNELS: List[Type[MPLinearKernel]] = [CDNA4LinearKernel]


def choose_mp_linear_kernel(config: MPLinearLayerConfig,
                            compute_capability: Optional[int] = None
                            ) -> Type[MPLinearKernel]:
    if compute_capability is None:
        compute_capability = 95  # gfx950 reports (9, 5)
    failure_reasons = []
    for kernel in _POSSIBLE_KERNELS:
        if kernel.__name__ in (switch("APHRODITE_DISABLED_KERNELS") or "").split(","):
            failure_reasons.append(f" {kernel.__name__} disabled by environment variable")
            continue
        if kernel.get_min_capability() > compute_capability:
            failure_reasons.append(
                f"{kernel.__name__} requires capability {kernel.get_min_capability()}, "
                f"current compute capability is {compute_capability}")
            continue
        can_implement, failure_reason = kernel.can_implement(config)
        if can_implement:
            return kernel
        failure_reasons.append(
            f" {kernel.__name__} cannot implement due to: {failure_reason}")
    raise ValueError("Failed to find a kernel that can implement the "
                     "WNA16 linear layer. Reasons: \n" + "\n".join(failure_reasons))


def register_with_reference(possible_kernels: list) -> None:
    """Prepend the CDNA4 kernel to the reference's own ``_POSSIBLE_KERNELS``
    (aphrodite/quantization/kernels/__init__.py:11-14); idempotent, meant to be
    called from an ``aphrodite.general_plugins`` entry point."""
    if CDNA4LinearKernel not in possible_kernels:
        possible_kernels.insert(0, CDNA4LinearKernel)
