"""All-reduce overlapped with the quantized GEMMs' weight stream on HIP streams (BASELINE.json north_star; SURVEY 5
last bullet / 7 step 6).

The decode layer's dependency chain is strict -- o_proj -> ALL-REDUCE -> norm -> gate_up -> down -> ALL-REDUCE -> norm
-> qkv -- so no GEMM's *math* can run under an all-reduce.  What can: its WEIGHT TRAFFIC.  The [M, hidden] all-reduce
is latency bound (flag round trips over xGMI, ~10 us, no HBM traffic) while the next column-parallel projection is
HBM bound on weights that do not depend on the reduced activations.  So:

    main stream   ... row-parallel GEMM | fork | prefetch(next weights) ........ | join | norm | next GEMM (weights on die)
    side stream                         |      all-reduce over xGMI peers        |

``fork`` / ``join`` are event record + stream-wait pairs: graph-capturable (they become the fork / join edges of the
captured HIP graph) and free of host synchronisation.  ``prefetch`` (csrc/cache_ops.hip: aphro_prefetch) pulls the next
projection's packed int4 / fp8 weights through the memory-side Infinity Cache (256 MiB: a 70B TP8 shard's projection is
7-15 MB), so the GEMM after the join streams them from the die instead of from HBM.  Numerics are untouched: the
all-reduce kernel is the same, only its stream changes (bit-identical outputs, tests/test_custom_ar_gpu.py).

The reference issues its all-reduce on the compute stream (parallel_state.py:321-379; custom_all_reduce.cuh:445-449
launches on the current stream), i.e. serialised with the GEMMs."""
from typing import Iterable, Optional

import torch

from .. import _lib
from .._lib import check


class AllReduceOverlap:
    """One side stream + two events per device, reused for every all-reduce of the step (the chain is serial, so one
    fork/join pair at a time is live)."""

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.side = torch.cuda.Stream(device=device)
        self._fork = torch.cuda.Event()
        self._join = torch.cuda.Event()
        self.stats = {"all_reduces": 0, "prefetched_bytes": 0}

    def all_reduce(self, reduce_fn, x: torch.Tensor, prefetch: Optional[Iterable[torch.Tensor]] = None) -> torch.Tensor:
        """out = reduce_fn(x) on the side stream; ``prefetch`` tensors are streamed on the current stream meanwhile."""
        main = torch.cuda.current_stream(self.device)
        self._fork.record(main)
        self.side.wait_event(self._fork)
        with torch.cuda.stream(self.side):
            out = reduce_fn(x)
            if not torch.cuda.is_current_stream_capturing():
                # eager: the caching allocator must not recycle x / out for the other stream before this one is done
                # (under capture both live in the graph's private pool for the graph's lifetime)
                x.record_stream(self.side)
                out.record_stream(main)
            self._join.record(self.side)
        if prefetch is not None:
            lib = _lib.lib()
            for w in prefetch:
                if w is None or not w.is_cuda or w.numel() == 0:
                    continue
                nbytes = w.numel() * w.element_size()
                check(lib.aphro_prefetch(w.data_ptr(), nbytes, main.cuda_stream), "prefetch")
                self.stats["prefetched_bytes"] += nbytes
        main.wait_event(self._join)
        self.stats["all_reduces"] += 1
        return out
