"""Every environment switch the PYTHON side of the package honours, in one table (the C library's are ``csrc/common.h``
``Knobs``; both lists are INTEGRATION.md "Switches").  They route a layer back to the previous generation of a kernel for
A/B measurements and bit-identity tests, or carry a deployment choice; none of them is needed for the fast path, which is
the default everywhere.  ``switch(NAME)`` is the only way the package reads one: a name that is not in the table is a bug."""
import os
from typing import Optional

SWITCHES = {
    # deployment
    "APHRODITE_MI355X_FUSED_MODEL": "0: plugin.register() leaves the reference's own Llama class in place (op-by-op path)",
    "APHRODITE_MI355X_LIB": "path of the C-ABI library (default: the in-tree build; the lab build for tools/)",
    "APHRODITE_MI355X_NO_STRIP_COPY": "1: op-level W4A16 layers keep one weight layout (no strip-major decode copy)",
    "APHRODITE_AWQ_NO_PREPACK": "1: AWQ weights stay in the checkpoint's nibble order (transposed per call)",
    "APHRODITE_DISABLED_KERNELS": "the reference's own list of MPLinearKernel names to skip (kernels/__init__.py:37-56)",
    "APHRODITE_CUSTOM_AR_CHECK_EVERY": "host-side error-word poll period of the peer-access all-reduce (calls)",
    "APHRO_WEIGHTS_TWO_COPIES": "1: DecoderLayer.enable_one_copy keeps the [K/8, N] originals beside the strip-major decode copies",
    "APHRO_AR_PREFETCH": "1: extra workgroups of the fused all-reduce + norm launch prefetch the next GEMM's weights",
    # A/B routing: the previous generation of a kernel / an unfused form (same bits unless stated)
    "APHRO_PA_ROCM_PARTITIONED": "_rocm_C::paged_attention always in its partitioned two-launch form",
    "APHRO_CA_NO_GATHER": "context_attention_fwd on the scalar-gather kernel instead of the tile machines",
    "APHRO_WNA16_NO_LARGE": "W4A16 at > 64 rows: dequantise + library GEMM (the reference's own rule) instead of the MFMA tile machine",
    "APHRO_WNA16_NO_MID": "W4A16 at 33..64 rows: the decode kernel instead of the 32x32x16 one-pass kernel",
    "APHRO_FP8_NO_LARGE": "W8A8 at > 64 rows: torch._scaled_mm instead of the MFMA tile machine",
    "APHRO_DECODE_ROW_HALVES": "1: 33..64-row decode batches on the stream kernel's two 32-row halves",
    "APHRO_DECODE_NO_RESIDENT": "fused int4 decode on the round-2 kernels (no strip-major copies)",
    "APHRO_DECODE_NO_FP8_RESIDENT": "fused FP8 decode on the round-3 kernels (no strip-major copies)",
    "APHRO_DECODE_NO_MID": "fused int4 decode at 33..64 rows without the one-pass kernel",
    "APHRO_DECODE_NO_SILU_SLABS": "K-sliced gate_up: separate slab reduce and SiluAndMul launches",
    "APHRO_FP8_NO_STATIC_FUSION": "static-scheme FP8 decode without the e4m3-writing attention / gate_up epilogues",
    "APHRO_FP8_NO_LAUNCH_DIET": "dynamic-scheme FP8 decode in its 9-launch form (round 5) instead of 7 launches (round 6)",
    "APHRO_PREFILL_NO_FUSED_FP8": "FP8 prefill with separate norm / SiluAndMul and quantisation launches",
    "APHRO_PREFILL_NO_SILU_EPILOGUE": "W4A16 prefill: SiluAndMul as its own launch",
    "APHRO_MOE_NO_NORM_ROUTER": "sparse MLP: router GEMM as its own launch",
    "APHRO_MOE_NO_DEFERRED_COMBINE": "sparse MLP: moe_combine as its own launch",
    "APHRO_MOE_NO_ROUTE_ALIGN": "sparse MLP: top-k softmax and block alignment as separate launches",
    "APHRO_NO_LM_HEAD_ARGMAX": "greedy decode: LM-head GEMM and argmax as separate launches",
    "APHRO_NO_FUSED_AR_NORM": "1: TP layers issue the all-reduce and the norm as two launches",
}


def switch(name: str) -> Optional[str]:
    """The environment value of a REGISTERED switch (None when unset)."""
    if name not in SWITCHES:
        raise KeyError(f"{name} is not a switch of this package (aphrodite_engine_amd/switches.py)")
    return os.environ.get(name)
