"""Tensor-parallel state + collectives for the hot path -- the slice of
aphrodite/distributed/{parallel_state,communication_op}.py the layers call
(tensor_model_parallel_all_reduce :9-12, all_gather :15-18; GroupCoordinator
.all_reduce parallel_state.py:321-379).

One process per GPU; the TP group is a torch.distributed group whose "nccl"
backend IS RCCL over xGMI on ROCm ("gloo" in the CPU tests).  The all-reduce
is issued on the current stream so it is captured into the decode HIP graph
like the reference's pynccl path (pynccl.py:102-118)."""
import contextlib
from typing import Optional

import torch
from ..switches import switch
import torch.distributed as dist

_TP_GROUP: Optional[dist.ProcessGroup] = None
_TP_RANK = 0
_TP_SIZE = 1
_CUSTOM_AR = None      # CustomAllreduce of the TP group (enable_custom_all_reduce)
_OVERLAP = None        # AllReduceOverlap (enable_all_reduce_overlap): all-reduce on a side stream + weight prefetch
_ADOPTED = False       # TP state taken from the reference engine's GroupCoordinator (adopt_reference_parallel_state)
_SIM_AR_US = None      # simulated TP (init_simulated_tensor_parallel): {bytes threshold: microseconds} of the stubbed all-reduce


def init_tensor_parallel(tp_size: int, backend: Optional[str] = None) -> None:
    """initialize_model_parallel (parallel_state.py:968-1040) for TP only:
    consecutive ranks form a TP group."""
    global _TP_GROUP, _TP_RANK, _TP_SIZE
    if tp_size <= 1:
        _TP_GROUP, _TP_RANK, _TP_SIZE = None, 0, 1
        return
    assert dist.is_initialized(), "torch.distributed must be initialised first"
    world = dist.get_world_size()
    assert world % tp_size == 0
    rank = dist.get_rank()
    for start in range(0, world, tp_size):
        ranks = list(range(start, start + tp_size))
        grp = dist.new_group(ranks, backend=backend)
        if rank in ranks:
            _TP_GROUP = grp
            _TP_RANK = rank - start
            _TP_SIZE = tp_size


def init_simulated_tensor_parallel(tp_size: int, all_reduce_us, rank: int = 0) -> None:
    """ONE rank of a TP group of ``tp_size`` on a one-GPU box (bench.py --sim-tp): every layer
    shards exactly as it would in the group (column / row parallel, heads, vocabulary), the rank runs its real per-GPU
    kernels on its real shard shapes, and each all-reduce is replaced by a launch that holds the stream for the latency
    measured for that message size (``all_reduce_us``: float, or {max bytes: us} picked by size) -- the data is left as
    this rank's partial sum, so the OUTPUT is meaningless, only the timing is.  The logits gather returns the local
    shard."""
    global _TP_GROUP, _TP_RANK, _TP_SIZE, _SIM_AR_US
    _TP_GROUP, _TP_RANK, _TP_SIZE = None, rank, tp_size
    _SIM_AR_US = all_reduce_us if isinstance(all_reduce_us, dict) else {1 << 62: float(all_reduce_us)}


def enable_loopback_all_reduce(device, max_size: int = 8192 * 1024):
    """Simulated TP only: replace the stream-holding stub by the REAL peer-access kernels on a loopback communicator
    (distributed/custom_all_reduce.py: LoopbackAllreduce) -- flags, scratch and ``tp_size`` reads per element against
    local memory.  The stub stays the fallback for sizes the kernel does not take."""
    global _CUSTOM_AR
    assert _SIM_AR_US is not None and _TP_SIZE > 1, "enable_loopback_all_reduce follows init_simulated_tensor_parallel"
    from .custom_all_reduce import LoopbackAllreduce
    _CUSTOM_AR = LoopbackAllreduce(_TP_SIZE, device, max_size)
    return _CUSTOM_AR


def reference_parallel_sizes():
    """(tensor-parallel size, pipeline-parallel size) of the REFERENCE engine this process runs inside
    (aphrodite/distributed/parallel_state.py:875-889, 1104-1111), or None when the reference is not importable or its
    groups are not initialised (standalone use of this package)."""
    try:
        from aphrodite.distributed import parallel_state as ps
    except Exception:
        return None
    try:
        tp = int(ps.get_tp_group().world_size)
    except Exception:
        return None
    try:
        pp = int(ps.get_pp_group().world_size)
    except Exception:
        pp = 1
    return tp, pp


def adopt_reference_parallel_state() -> bool:
    """Inside the reference engine the TP group is the reference's ``GroupCoordinator`` (parallel_state.py:131-231), built
    by ``initialize_model_parallel`` before any model class is constructed (worker/worker.py ``init_worker_distributed_
    environment`` -> ``load_model``).  This package's layers shard by ITS OWN ``_TP_*`` state, so a fused model built
    under a TP > 1 engine must take group, rank and size from there -- otherwise every rank builds the unsharded model
    while the worker allocates KV caches for ``num_kv_heads / tp`` heads.  Adopts ``device_group``
    (the RCCL group the reference's own all-reduce uses), ``rank_in_group``, ``world_size`` and -- when the plugin has
    swapped this package's ``CustomAllreduce`` in and it is enabled -- the group's ``ca_comm``.  Returns True when this
    package's TP state now equals the reference's (also when both are 1 or it already matched)."""
    global _TP_GROUP, _TP_RANK, _TP_SIZE, _CUSTOM_AR, _ADOPTED
    try:
        from aphrodite.distributed import parallel_state as ps
        grp = ps.get_tp_group()
    except Exception:
        return False
    size = int(grp.world_size)
    if size == 1:
        return _TP_SIZE == 1
    rank = int(grp.rank_in_group)
    device_group = getattr(grp, "device_group", None)
    if device_group is None:
        return False
    if _TP_SIZE == size and _TP_RANK == rank and _TP_GROUP is device_group:
        return True
    if _TP_SIZE != 1 or _SIM_AR_US is not None:
        return False            # this package's TP state was initialised to something else: do not overwrite it
    _TP_GROUP, _TP_RANK, _TP_SIZE, _ADOPTED = device_group, rank, size, True
    ca = getattr(grp, "ca_comm", None)
    from .custom_all_reduce import CustomAllreduce
    if isinstance(ca, CustomAllreduce) and not getattr(ca, "disabled", True):
        _CUSTOM_AR = ca
    return True


def destroy_tensor_parallel() -> None:
    global _TP_GROUP, _TP_RANK, _TP_SIZE, _CUSTOM_AR, _OVERLAP, _SIM_AR_US, _ADOPTED
    if _CUSTOM_AR is not None and not _ADOPTED:     # an adopted communicator belongs to the reference's GroupCoordinator
        _CUSTOM_AR.close()
    _TP_GROUP, _TP_RANK, _TP_SIZE, _CUSTOM_AR, _OVERLAP, _SIM_AR_US, _ADOPTED = None, 0, 1, None, None, None, False


def enable_custom_all_reduce(device, cpu_group: Optional[dist.ProcessGroup] = None, max_size: int = 8192 * 1024):
    """Attach the xGMI peer-access all-reduce to the TP group (GroupCoordinator.__init__,
    parallel_state.py:186-196: ``ca_comm``).  ``cpu_group``: a non-NCCL group over the same ranks for
    the one-off handle exchange (the reference's ``cpu_group``); made here (gloo) if omitted."""
    global _CUSTOM_AR
    if _TP_SIZE == 1:
        return None
    from .custom_all_reduce import CustomAllreduce
    if cpu_group is None:
        world = dist.get_world_size()
        rank = dist.get_rank()
        for start in range(0, world, _TP_SIZE):
            ranks = list(range(start, start + _TP_SIZE))
            grp = dist.new_group(ranks, backend="gloo")
            if rank in ranks:
                cpu_group = grp
    _CUSTOM_AR = CustomAllreduce(cpu_group, device, max_size)
    return _CUSTOM_AR


def get_custom_all_reduce():
    return _CUSTOM_AR


def enable_all_reduce_overlap(device, enabled: bool = True):
    """Run the TP all-reduces on a side stream, overlapped with a prefetch of the next projection's weights
    (distributed/overlap.py).  Graph-capturable; results are bit-identical to the serial path."""
    global _OVERLAP
    if not enabled or _TP_SIZE == 1:
        _OVERLAP = None
        return None
    from .overlap import AllReduceOverlap
    _OVERLAP = AllReduceOverlap(torch.device(device))
    return _OVERLAP


def get_all_reduce_overlap():
    return _OVERLAP


def choose_all_reduce_overlap(device, timed_run, margin: float = 0.03) -> dict:
    """The side-stream overlap (distributed/overlap.py: all-reduce on its own stream + a prefetch of the next projection's
    weights) is kept ONLY behind a measured win on the group it runs on: ``timed_run()`` -> seconds of a representative
    decode run under the CURRENT setting is called with the overlap off, then on; the overlap stays enabled iff it is
    faster by more than ``margin``.  On one GPU it measured 38-61 % slower, so nothing enables it unmeasured.
    Every rank must take the same decision: the two times are max-reduced over the TP group first.  Returns the record."""
    rec = {"margin": margin}
    times = {}
    for name, on in (("serial", False), ("overlap", True)):
        enable_all_reduce_overlap(device, enabled=on)
        try:
            times[name] = float(timed_run())
        except Exception as e:      # noqa: BLE001 -- a failed arm loses, it does not take the caller down
            rec[name + "_error"] = repr(e)[:200]
            times[name] = float("inf")
    if _TP_GROUP is not None and dist.is_initialized():
        t = torch.tensor([times["serial"], times["overlap"]], dtype=torch.float64,
                         device=device if dist.get_backend(_TP_GROUP) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_TP_GROUP)
        times = {"serial": float(t[0]), "overlap": float(t[1])}
    keep = times["overlap"] < times["serial"] * (1.0 - margin)
    enable_all_reduce_overlap(device, enabled=keep)
    rec.update(serial_s=times["serial"], overlap_s=times["overlap"], enabled=bool(keep))
    return rec


def all_reduce_self_check(device, sizes=(32 * 4096 * 2, 256 * 1024, 1024 * 1024, 4 * 1024 * 1024)) -> dict:
    """First contact with a real TP group, BEFORE anything is timed or captured: for every message size the decode path can
    issue, run the peer-access kernel (one- / two-shot as it would choose) on data whose sum is exact in f16 and compare
    with ``dist.all_reduce`` on the same group; the same for the fused all-reduce + residual add + RMSNorm against the
    two-op sequence.  A size that does not verify DISABLES the peer-access communicator (every all-reduce then goes to
    RCCL): the report says "peer path verified" / "fell back to RCCL" per size instead of a hang or a silently wrong sum
    (the kernels' waits are bounded, custom_all_reduce.hip).  Collective: every rank of the group calls it."""
    out = {}
    ca = _CUSTOM_AR
    if _TP_SIZE == 1 or _TP_GROUP is None:
        return {"skipped": "no tensor-parallel group"}
    from .. import _custom_ops as ops
    ok_all = True
    for nbytes in sizes:
        n = nbytes // 2
        # small integers, different on every rank: the sum over <= 8 ranks is exact in f16 whatever the order
        base = (torch.arange(n, device=device, dtype=torch.int32) * 7 + _TP_RANK * 13) % 31 - 15
        x = base.to(torch.float16)
        want = x.clone()
        dist.all_reduce(want, group=_TP_GROUP)
        row = {"rccl": "ok"}
        if ca is None or getattr(ca, "disabled", True):
            row["peer"] = "off (" + str(getattr(ca, "disabled_reason", "no communicator")) + ")"
        elif not ca.should_custom_ar(x):
            row["peer"] = "not eligible at this size: RCCL serves it"
        else:
            try:
                got = ca.custom_all_reduce(x)
                torch.cuda.synchronize(device)
                ca.check()
                same = got is not None and torch.equal(got, want)
                algo = "one-shot" if ops.should_one_shot(_TP_SIZE, nbytes) else "two-shot"
                row["peer"] = f"verified ({algo})" if same else f"MISMATCH ({algo})"
                ok_all = ok_all and same
            except Exception as e:      # noqa: BLE001
                row["peer"] = "error: " + repr(e)[:160]
                ok_all = False
        out[str(nbytes)] = row
    # the fused all-reduce + norm on the decode layer's own [32, 4096] sum
    if ok_all and ca is not None and not getattr(ca, "disabled", True):
        try:
            tokens, hidden = 32, 4096
            xm = (((torch.arange(tokens * hidden, device=device, dtype=torch.int32) * 5 + _TP_RANK * 11) % 29 - 14)
                  .to(torch.float16).view(tokens, hidden))
            res = torch.ones(tokens, hidden, dtype=torch.float16, device=device)
            w = torch.full((hidden, ), 0.5, dtype=torch.float16, device=device)
            if ca.fused_norm_eligible(xm):
                r1, r2 = res.clone(), res.clone()
                summed = xm.clone()
                dist.all_reduce(summed, group=_TP_GROUP)
                _, o_ref = ops.fused_add_rms_norm_pack(summed, None, r1, True, w, 1e-5, pack=False, want_out=True)
                got = ca.fused_add_rms_norm(xm, r2, True, w, 1e-5, pack=False, want_out=True)
                torch.cuda.synchronize(device)
                ca.check()
                same = got is not None and torch.equal(got[1], o_ref) and torch.equal(r1, r2)
                out["fused_all_reduce_norm"] = "verified" if same else "MISMATCH"
                ok_all = ok_all and same
        except Exception as e:      # noqa: BLE001
            out["fused_all_reduce_norm"] = "error: " + repr(e)[:160]
            ok_all = False
    # one decision for the whole group
    flag = torch.tensor([1 if ok_all else 0], dtype=torch.int32, device=device if dist.get_backend(_TP_GROUP) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=_TP_GROUP)
    if int(flag.item()) == 0 and ca is not None and not getattr(ca, "disabled", True):
        ca.disabled = True
        ca.disabled_reason = "self-check failed on at least one rank"
        out["decision"] = "fell back to RCCL for every size"
    else:
        out["decision"] = "peer path verified" if (ca is not None and not getattr(ca, "disabled", True)) else "RCCL (no peer-access communicator)"
    return out


@contextlib.contextmanager
def simulated_tensor_parallel(rank: int, size: int):
    """Pretend to be TP rank ``rank`` of ``size`` with no process group: for building one rank's
    parameter shards offline (checkpoint resharding, loader tests).  Collectives are not available."""
    global _TP_GROUP, _TP_RANK, _TP_SIZE
    saved = (_TP_GROUP, _TP_RANK, _TP_SIZE)
    _TP_GROUP, _TP_RANK, _TP_SIZE = None, rank, size
    try:
        yield
    finally:
        _TP_GROUP, _TP_RANK, _TP_SIZE = saved


def get_tensor_model_parallel_world_size() -> int:
    return _TP_SIZE


def get_tensor_model_parallel_rank() -> int:
    return _TP_RANK


def _all_reduce_serial(input_: torch.Tensor) -> torch.Tensor:
    # GroupCoordinator.all_reduce (parallel_state.py:321-379): the peer-access kernel when eligible
    # (out of place), else RCCL in place
    if _CUSTOM_AR is not None and input_.is_cuda:
        out = _CUSTOM_AR.custom_all_reduce(input_)
        if out is not None:
            return out
    dist.all_reduce(input_, group=_TP_GROUP)
    return input_


def tensor_model_parallel_all_reduce(input_: torch.Tensor, prefetch=None) -> torch.Tensor:
    """Sum over the TP group (communication_op.py:9-12).  ``prefetch``: tensors (the next projection's packed
    weights) to stream through the Infinity Cache while the all-reduce is in flight -- used only when
    enable_all_reduce_overlap() is on; otherwise ignored and the all-reduce runs on the current stream."""
    if _TP_SIZE == 1:
        return input_
    if _SIM_AR_US is not None:
        from .. import _lib
        if _CUSTOM_AR is not None and input_.is_cuda:          # loopback communicator: the real kernel on local memory
            out = _CUSTOM_AR.custom_all_reduce(input_)
            if out is not None:
                return out
        nbytes = input_.numel() * input_.element_size()
        us = next(v for k, v in sorted(_SIM_AR_US.items()) if nbytes <= k)
        _lib.check(_lib.lib().aphro_spin_us(float(us), torch.cuda.current_stream(input_.device).cuda_stream), "spin_us")
        return input_
    if _OVERLAP is not None and input_.is_cuda:
        return _OVERLAP.all_reduce(_all_reduce_serial, input_, prefetch)
    return _all_reduce_serial(input_)


def tensor_model_parallel_all_reduce_norm(input_: torch.Tensor, residual: Optional[torch.Tensor], has_residual: bool,
                                         weight: torch.Tensor, epsilon: float, pack: bool = True, want_out: bool = False,
                                         prefetch: Optional[torch.Tensor] = None):
    """The row-parallel linear's all-reduce (linear.py:1142-1143) and the fused_add_rms_norm [+ pack] that follows it in
    every decoder layer as ONE launch of the peer-access kernel (csrc/custom_all_reduce.hip) -- the bits of
    tensor_model_parallel_all_reduce -> ops.fused_add_rms_norm_pack.  Returns (packed, out), or None when the fused
    form does not apply (no peer-access communicator, overlap mode, ineligible size): the caller then issues the two ops."""
    if _TP_SIZE == 1 or _CUSTOM_AR is None or _OVERLAP is not None or not input_.is_cuda:
        return None
    return _CUSTOM_AR.fused_add_rms_norm(input_, residual, has_residual, weight, epsilon, pack=pack, want_out=want_out,
                                         prefetch=prefetch)


def tensor_model_parallel_all_reduce_norm_quant_fp8(input_: torch.Tensor, residual: Optional[torch.Tensor],
                                                   has_residual: bool, weight: torch.Tensor, epsilon: float,
                                                   want_out: bool = False, static_scale: Optional[torch.Tensor] = None):
    """tensor_model_parallel_all_reduce_norm for an FP8 W8A8 layer: the all-reduce, fused_add_rms_norm and the next
    linear's activation quantisation (quantization/fp8.py -> ops.scaled_fp8_quant) as ONE launch -- the bits of
    tensor_model_parallel_all_reduce -> ops.fused_add_rms_norm_quant_fp8.  Returns (q, scales, out) or None."""
    if _TP_SIZE == 1 or _CUSTOM_AR is None or _OVERLAP is not None or not input_.is_cuda:
        return None
    return _CUSTOM_AR.fused_add_rms_norm_quant_fp8(input_, residual, has_residual, weight, epsilon, want_out=want_out,
                                                   static_scale=static_scale)


def tensor_model_parallel_all_reduce_norm_router(input_: torch.Tensor, residual: Optional[torch.Tensor], has_residual: bool,
                                                weight: torch.Tensor, epsilon: float, router_weight: torch.Tensor):
    """tensor_model_parallel_all_reduce_norm for a sparse-MLP layer: the attention block's all-reduce, fused_add_rms_norm and
    the router's logits (MixtralMoE.gate, mixtral.py:60-110) as ONE launch -- the bits of tensor_model_parallel_all_reduce ->
    ops.fused_add_rms_norm_router.  Returns (normed, router_logits) or None."""
    if _TP_SIZE == 1 or _CUSTOM_AR is None or _OVERLAP is not None or not input_.is_cuda:
        return None
    return _CUSTOM_AR.fused_add_rms_norm_router(input_, residual, has_residual, weight, epsilon, router_weight)


class DeferredAllReduce:
    """A row-parallel projection's per-rank partial sums [tokens, hidden] whose all-reduce has NOT been issued: the norm
    that consumes them runs it in its own launch (``finish``)."""

    def __init__(self, partial: torch.Tensor):
        self.partial = partial

    def finish(self, residual, weight, epsilon, pack=True, want_out=False, prefetch=None):
        """``prefetch`` (opt-in, APHRO_AR_PREFETCH=1): the packed weights of the GEMM that consumes the norm -- streamed
        through the Infinity Cache by extra workgroups of the same launch.  Off by default: on the one-GPU loopback rig
        the longer launch cost more than the warmer weights returned (profiles/r5_ar_norm_fused.txt)."""
        import os
        if switch("APHRO_AR_PREFETCH") != "1":
            prefetch = None
        res = tensor_model_parallel_all_reduce_norm(self.partial, residual, True, weight, epsilon, pack=pack,
                                                    want_out=want_out, prefetch=prefetch)
        if res is None:
            # the communicator stopped serving the shape between defer and finish (disabled after a peer timeout, a
            # capture-time registration refused): the two launches the fused one stands for -- same bits
            from .. import _custom_ops as ops
            x = tensor_model_parallel_all_reduce(self.partial)
            res = ops.fused_add_rms_norm_pack(x, None, residual, True, weight, epsilon, pack=pack, want_out=want_out)
        return res

    def finish_router(self, residual, weight, epsilon, router_weight):
        """finish() for the norm in front of a sparse MLP: (normed rows, router logits) of ops.fused_add_rms_norm_router on the
        all-reduced partial, one launch."""
        res = tensor_model_parallel_all_reduce_norm_router(self.partial, residual, True, weight, epsilon, router_weight)
        if res is None:     # (as in finish: the two launches the fused one stands for -- same bits)
            from .. import _custom_ops as ops
            x = tensor_model_parallel_all_reduce(self.partial)
            res = ops.fused_add_rms_norm_router(x, None, residual, True, weight, epsilon, router_weight)
        return res

    def finish_quant_fp8(self, residual, weight, epsilon, want_out=False, static_scale=None):
        """finish() for a consumer that is an FP8 W8A8 linear: (q, scales, out) of ops.fused_add_rms_norm_quant_fp8 on the
        all-reduced partial, one launch."""
        res = tensor_model_parallel_all_reduce_norm_quant_fp8(self.partial, residual, True, weight, epsilon,
                                                              want_out=want_out, static_scale=static_scale)
        if res is None:     # (as in finish: the two launches the fused one stands for -- same bits)
            from .. import _custom_ops as ops
            x = tensor_model_parallel_all_reduce(self.partial)
            res = ops.fused_add_rms_norm_quant_fp8(x, None, None, None, residual, True, weight, epsilon, want_out=want_out,
                                                   static_scale=static_scale)
        return res


def defer_all_reduce(partial: torch.Tensor) -> Optional["DeferredAllReduce"]:
    """A DeferredAllReduce for ``partial`` when the fused all-reduce + norm launch serves it (peer-access communicator
    attached, decode-sized [tokens <= 64, hidden] f16 / bf16, no side-stream overlap; APHRO_NO_FUSED_AR_NORM=1 opts
    out), else None: the caller all-reduces now."""
    import os
    if (_TP_SIZE == 1 or _CUSTOM_AR is None or _OVERLAP is not None or not partial.is_cuda
            or switch("APHRO_NO_FUSED_AR_NORM") == "1" or _CUSTOM_AR.disabled
            or not _CUSTOM_AR.fused_norm_eligible(partial)):
        return None
    return DeferredAllReduce(partial)


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """communication_op.py:15-18 / parallel_state.py:381-416."""
    if _TP_SIZE == 1 or _SIM_AR_US is not None:
        return input_
    if dim < 0:
        dim += input_.dim()
    inp = input_.contiguous()
    if inp.dim() == 0:
        inp = inp.reshape(1)
    out = torch.empty((_TP_SIZE * inp.shape[0], ) + tuple(inp.shape[1:]),
                      dtype=inp.dtype, device=inp.device)
    dist.all_gather_into_tensor(out, inp, group=_TP_GROUP)
    out = out.view((_TP_SIZE, ) + tuple(inp.shape)).movedim(0, dim)
    shape = list(input_.shape)
    shape[dim] *= _TP_SIZE
    return out.reshape(shape)
