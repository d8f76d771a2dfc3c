"""Scenario definitions shared by tests/golden/make_golden_attn_builder.py (runs the REFERENCE's CommonMetadataBuilder /
CommonAttentionState on them) and tests/test_host_cpu.py (runs ours): stand-ins for the model runner's per-step data
(worker/model_runner.py ModelInputForGPUBuilder.InterDataForSeqGroup)."""
from types import SimpleNamespace

import numpy as np

BLOCK = 16


def _group(is_prompt, seq_ids, tokens, orig_seq_lens, seq_lens, query_lens, context_lens, block_tables,
           prefix_cache_hit=False, computed_block_nums=None, sliding_blocks=None):
    return SimpleNamespace(is_prompt=is_prompt, seq_ids=seq_ids, input_tokens=[[7] * t for t in tokens],
                           orig_seq_lens=orig_seq_lens, seq_lens=seq_lens, query_lens=query_lens,
                           context_lens=context_lens, block_tables=block_tables, prefix_cache_hit=prefix_cache_hit,
                           computed_block_nums=computed_block_nums or [],
                           curr_sliding_window_blocks=sliding_blocks or [10 ** 9] * len(seq_ids))


def scenarios():
    """name -> dict(groups, seq_lens, query_lens, pad, batch, chunked, sliding_window)."""
    out = {}
    bt = {0: [5, 9, 2], 1: [7], 2: [3, 11, 4, 8]}
    dec = [_group(False, [i], [1], [L], [L], [1], [L - 1], bt) for i, L in ((0, 40), (1, 3), (2, 64))]
    out["decode_eager"] = dict(groups=dec, seq_lens=[40, 3, 64], query_lens=[1, 1, 1], pad=-1, batch=3)
    out["decode_graph_pad"] = dict(groups=dec, seq_lens=[40, 3, 64, 1], query_lens=[1, 1, 1, 1], pad=1, batch=4)
    pbt = {0: [1, 2], 1: [6, 4, 0]}
    pre = [_group(True, [0], [20], [20], [20], [20], [0], pbt), _group(True, [1], [37], [37], [37], [37], [0], pbt)]
    out["prefill_only"] = dict(groups=pre, seq_lens=[20, 37], query_lens=[20, 37], pad=-1, batch=2)
    mixed = [_group(True, [0], [8], [24], [24], [8], [16], {0: [12, 13], 1: [5, 9, 2], 2: [7]})] + \
        [_group(False, [1], [1], [40], [40], [1], [39], {0: [12, 13], 1: [5, 9, 2], 2: [7]}),
         _group(False, [2], [1], [3], [3], [1], [2], {0: [12, 13], 1: [5, 9, 2], 2: [7]})]
    out["chunked_mixed"] = dict(groups=mixed, seq_lens=[24, 40, 3], query_lens=[8, 1, 1], pad=-1, batch=3, chunked=True)
    hit = [_group(True, [0], [10], [42], [42], [10], [32], {0: [3, 4, 5]}, prefix_cache_hit=True,
                  computed_block_nums=[3, 4])]
    out["prefix_cache_hit"] = dict(groups=hit, seq_lens=[42], query_lens=[10], pad=-1, batch=1)
    slide = [_group(True, [0], [12], [12], [12], [12], [0], {0: [9]})]
    out["sliding_window"] = dict(groups=slide, seq_lens=[12], query_lens=[12], pad=-1, batch=1, sliding_window=8)
    prof = [_group(True, [0], [6], [6], [6], [6], [0], None)]
    out["profile_run"] = dict(groups=prof, seq_lens=[6], query_lens=[6], pad=-1, batch=1)
    return out


def make_input_builder(sc, device="cpu"):
    runner = SimpleNamespace(device=device, pin_memory=False, graph_block_tables=np.zeros((8, 6), dtype=np.int32),
                             max_seq_len_to_capture=96)
    return SimpleNamespace(runner=runner, sliding_window=sc.get("sliding_window"), block_size=BLOCK,
                           scheduler_config=SimpleNamespace(use_v2_block_manager=True),
                           inter_data_list=sc["groups"], chunked_prefill_enabled=sc.get("chunked", False))


FIELDS = ("num_prefills", "num_prefill_tokens", "num_decode_tokens", "slot_mapping", "seq_lens", "seq_lens_tensor",
          "max_query_len", "max_prefill_seq_len", "max_decode_seq_len", "query_start_loc", "seq_start_loc",
          "context_lens_tensor", "block_tables", "use_cuda_graph")


def to_plain(meta):
    d = {}
    for f in FIELDS:
        v = getattr(meta, f)
        d[f] = v.tolist() if hasattr(v, "tolist") else v
    return d
