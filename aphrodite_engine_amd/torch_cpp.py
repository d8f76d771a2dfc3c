"""Loader of the C++ ``TORCH_LIBRARY`` registration (csrc_torch/torch_bindings.cpp ->
lib/libaphrodite_mi355x_torch.so): ops registered from C++ with the reference's schemas
(kernels/torch_bindings.cpp), forwarding to the C ABI with no Python in the dispatch path.

    from aphrodite_engine_amd import torch_cpp
    torch_cpp.load()                       # registers torch.ops._C_mi355x.* and torch.ops._C_mi355x_cache_ops.*
    torch.ops._C_mi355x.gptq_gemm(...)

Built with ``make -C aphrodite_engine_amd/csrc_torch NS=_C`` the same file takes the reference extension's place."""
import os

import torch

NAMESPACE = "_C_mi355x"
_LOADED = False


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libaphrodite_mi355x_torch.so")


def load() -> str:
    """Idempotent; raises if the library has not been built (``__graft_entry__.build()``)."""
    global _LOADED
    path = library_path()
    if not _LOADED:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C aphrodite_engine_amd/csrc_torch` (or __graft_entry__.build())")
        torch.ops.load_library(path)
        _LOADED = True
    return path
