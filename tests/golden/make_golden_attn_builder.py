#!/usr/bin/env python3
"""Golden metadata from the REFERENCE's own CommonMetadataBuilder / CommonAttentionState
(aphrodite/attention/backends/utils.py:123-372 -- what ROCmFlashAttentionBackend.get_builder_cls / get_state_cls
return, rocm_flash_attn.py:41-47), executed from where it lies on the scenarios of tests/attn_builder_cases.py.
Run in the build container only:   python tests/golden/make_golden_attn_builder.py   -> tests/golden/attn_builder.json"""
import importlib.util
import json
import os
import sys
import types
from dataclasses import make_dataclass
from typing import Generic, TypeVar

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
import attn_builder_cases as cases  # noqa: E402
from make_golden import REF, _lift, _stub  # noqa: E402

T = TypeVar("T")


def load_reference_utils():
    class AttentionMetadata:
        pass

    class AttentionMetadataBuilder(Generic[T]):
        pass

    class AttentionState(Generic[T]):
        pass
    _stub("aphrodite")
    _stub("aphrodite.attention", AttentionMetadata=AttentionMetadata, AttentionMetadataBuilder=AttentionMetadataBuilder,
          AttentionState=AttentionState)
    import numpy.typing as npt
    from typing import List, Optional, Union
    ns = _lift("aphrodite/common/utils.py", {"TORCH_DTYPE_TO_NUMPY_DTYPE", "make_ndarray_with_pad", "make_tensor_with_pad",
                                             "async_tensor_h2d"},
               dict(torch=torch, np=np, npt=npt, List=List, Optional=Optional, Union=Union, T=T))
    _stub("aphrodite.common")
    _stub("aphrodite.common.utils", async_tensor_h2d=ns["async_tensor_h2d"], make_tensor_with_pad=ns["make_tensor_with_pad"])
    spec = importlib.util.spec_from_file_location("aphrodite.attention.backends.utils",
                                                  os.path.join(REF, "aphrodite/attention/backends/utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference_utils()
    Meta = make_dataclass("Meta", [(f, object) for f in cases.FIELDS])
    Meta.decode_metadata = property(lambda self: self)

    class Builder(ref.CommonMetadataBuilder):
        _metadata_cls = Meta
    out = {}
    for name, sc in cases.scenarios().items():
        ib = cases.make_input_builder(sc)
        meta = Builder(ib).build(sc["seq_lens"], sc["query_lens"], sc["pad"], sc["batch"])
        out[name] = cases.to_plain(meta)
    # CommonAttentionState: the capture-time metadata and the buffer refresh
    backend = types.SimpleNamespace(make_metadata=lambda **kw: Meta(**kw), get_name=lambda: "rocm-flash-attn")
    runner = types.SimpleNamespace(device="cpu", graph_block_tables=np.arange(48, dtype=np.int32).reshape(8, 6),
                                   max_seq_len_to_capture=96, attn_backend=backend)
    st = ref.CommonAttentionState(runner)
    with st.graph_capture(8):
        m = st.graph_capture_get_metadata_for_batch(4)
        out["state_capture_batch4"] = cases.to_plain(m)
        bufs = st.get_graph_input_buffers(m)
        out["state_buffer_keys"] = sorted(bufs.keys())
    with open(os.path.join(HERE, "attn_builder.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote attn_builder.json:", list(out))


if __name__ == "__main__":
    main()
