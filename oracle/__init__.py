"""CPU oracle for the MI355X quantized-inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported, linked or
executed by the product path (``aphrodite_engine_amd``); only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it,
and only as the *checker*.

Each function restates -- in numpy / torch-CPU -- the algorithm of the reference
(PygmalionAI/aphrodite-engine @ 2025-01-17) and cites the reference file:line it
follows.  Pinning status (see DESIGN.md "Oracle pinning"):

* pack/unpack, quantize_weights, sort/permutation helpers, AWQ dequant, fp8
  scaled-quant, scaled-mm, paged-attention and reshape_and_cache restatements
  are pinned against golden vectors generated from the reference's own Python
  (``tests/golden/make_golden.py``), and paged attention / cache write also
  against the reference's own CPU kernels compiled into ``oracle/_ref``.
* ``gptq_shuffle`` / ``gptq_gemm`` / CUDA ``awq_gemm``: the reference holds no
  kernel-level test or fixture for them -- **parity
  unpinned** at kernel level; the restatement follows the CUDA source
  (q_gemm.cu, qdq_4.cuh) line by line and is cross-checked for internal
  consistency (shuffle o dequant == dequant).
"""
