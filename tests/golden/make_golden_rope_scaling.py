"""Golden cos|sin tables of the reference's scaled rotary embeddings (linear, dynamic NTK, YaRN):
modeling/layers/rotary_embedding.py:205-287, 291-329, 332-430.  Run HERE (the reference is read at /root/reference, its
functions executed in place -- nothing of it is stored); the GPU box only sees tests/golden/rope_scaling.npz.

    python tests/golden/make_golden_rope_scaling.py
"""
import ast
import math
import os
import textwrap
import types

import numpy as np
import torch

REF = os.environ.get("APHRODITE_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
SRC = "aphrodite/modeling/layers/rotary_embedding.py"


def lift(cls, name, glb):
    src = open(os.path.join(REF, SRC)).read()
    for node in ast.parse(src).body:
        if cls is None and isinstance(node, ast.FunctionDef) and node.name == name:
            exec(compile(ast.get_source_segment(src, node), SRC, "exec"), glb)
            return glb[name]
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == name:
                    ns = dict(glb)
                    exec(compile(textwrap.dedent(ast.get_source_segment(src, sub, padded=True)), SRC, "exec"), ns)
                    return ns[name]
    raise KeyError((cls, name))


def main():
    from typing import Dict, List, Tuple, Union
    g = dict(torch=torch, math=math, Dict=Dict, List=List, Tuple=Tuple, Union=Union)
    for fn in ("_yarn_find_correction_dim", "_yarn_find_correction_range", "_yarn_linear_ramp_mask", "_yarn_get_mscale"):
        lift(None, fn, g)
    base_inv = lift("RotaryEmbedding", "_compute_inv_freq", g)
    out = {}
    cases = {}

    def obj(**kw):
        o = types.SimpleNamespace(**kw)
        o._compute_inv_freq = lambda b: base_inv(o, b)
        return o
    # linear (one factor): positions divided by the factor, factor x max_position rows
    lin_cache = lift("LinearScalingRotaryEmbedding", "_compute_cos_sin_cache", g)
    o = obj(rotary_dim=128, base=10000.0, max_position_embeddings=96, scaling_factors=[4.0])
    out["linear"] = lin_cache(o).numpy()
    cases["linear"] = dict(head_dim=128, max_pos=96, theta=10000.0, rope_scaling={"rope_type": "linear", "factor": 4.0})
    # dynamic NTK: the base grows with the factor
    dyn_cache = lift("DynamicNTKScalingRotaryEmbedding", "_compute_cos_sin_cache", g)
    o = obj(rotary_dim=64, base=10000.0, max_position_embeddings=128, scaling_factor=2.0)
    out["dynamic"] = dyn_cache(o).numpy()
    cases["dynamic"] = dict(head_dim=64, max_pos=128, theta=10000.0, rope_scaling={"type": "dynamic", "factor": 2.0})
    # YaRN: interpolated / extrapolated frequencies blended over a ramp, cos / sin scaled by mscale
    yarn_inv = lift("YaRNScalingRotaryEmbedding", "_compute_inv_freq", g)
    yarn_cache = lift("YaRNScalingRotaryEmbedding", "_compute_cos_sin_cache", g)
    for tag, extra in (("yarn", {}), ("yarn_kw", {"attn_factor": 0.9, "beta_fast": 16, "beta_slow": 2, "extrapolation_factor": 1})):
        kw = dict(extrapolation_factor=1, attn_factor=1, beta_fast=32, beta_slow=1)
        kw.update(extra)
        o = types.SimpleNamespace(rotary_dim=128, base=1000000.0, max_position_embeddings=64, scaling_factor=4.0, **kw)
        o.mscale = float(g["_yarn_get_mscale"](o.scaling_factor) * o.attn_factor)      # __init__, :395-397
        o._compute_inv_freq = (lambda oo: (lambda s: yarn_inv(oo, s)))(o)
        out[tag] = yarn_cache(o).numpy()
        cases[tag] = dict(head_dim=128, max_pos=4096, theta=1000000.0,
                          rope_scaling=dict({"rope_type": "yarn", "factor": 4.0, "original_max_position_embeddings": 64}, **extra))
    np.savez_compressed(os.path.join(OUT, "rope_scaling.npz"), **out)
    import json
    with open(os.path.join(OUT, "rope_scaling_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
