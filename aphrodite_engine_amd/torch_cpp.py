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


def core_library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libaphrodite_mi355x_core.so")


def scalar_type_class():
    """``torch.classes._core_C.ScalarType`` or None when no library of the process has registered it."""
    try:
        return torch.classes._core_C.ScalarType
    except Exception:
        return None


def ensure_scalar_type_class():
    """The torchbind class the quantised GEMM schemas name (kernels/core/torch_bindings.cpp:10-13).  Inside the reference its own
    ``_core_C`` extension registers it (aphrodite/_core_ext.py imports that before any plugin runs); standalone this package's
    csrc_torch/core_scalar_type.cpp does -- loaded only when the name does not resolve: a class can be registered once."""
    cls = scalar_type_class()
    if cls is None and os.path.exists(core_library_path()):
        torch.ops.load_library(core_library_path())
        cls = scalar_type_class()
    return cls


def load() -> str:
    """Idempotent; raises if the library has not been built (``__graft_entry__.build()``)."""
    global _LOADED
    path = library_path()
    if not _LOADED:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C aphrodite_engine_amd/csrc_torch` (or __graft_entry__.build())")
        have_class = ensure_scalar_type_class() is not None
        torch.ops.load_library(path)
        # gptq_marlin_gemm: the verbatim schema (ScalarType b_q_type) when the class exists, else `int b_q_type`
        import ctypes
        rc = ctypes.CDLL(path).aphro_torch_register_marlin(1 if have_class else 0)
        if rc < 0:
            raise RuntimeError("torch_cpp.load: registering gptq_marlin_gemm failed (see stderr)")
        _LOADED = True
    return path
