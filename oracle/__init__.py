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
* ``gptq_shuffle`` / ``gptq_dequant`` / ``gptq_gemm``: pinned by the reference's OWN CUDA kernels
  (q_gemm.cu shuffle_4bit_kernel, make_sequential_4bit_kernel, reconstruct_exllama_4bit_kernel,
  reconstruct_gptq_kernel, gemm_half_q_half_gptq_4bit_kernel + qdq_4.cuh, matrix_view.cuh) compiled
  unmodified for the HOST against ``oracle/cuda_host_shim/`` (``oracle/Makefile`` ->
  ``oracle/_ref/libaphro_ref_gptq.so``; fixtures ``tests/golden/gptq_ref.npz`` from
  ``tests/golden/make_golden_gptq.py``): shuffle / act-order repack and both fp16 dequant kernels agree
  bit for bit, the M <= 50 exllama GEMM within its own fp16-accumulation noise (mean rel 5e-4).
* ``awq_gemm``: on ROCm the reference's op IS ``awq_gemm_triton``, which its own test pins against
  ``matmul(x, awq_dequantize_torch(...))`` (tests/kernels/test_awq_triton.py) -- the golden-pinned
  dequant + a matmul, as restated here.  (The CUDA PTX kernel is not compilable on a host.)
"""
