"""CustomAllreduce -- the tensor-parallel sum over xGMI peer access, host side.  Mirror of
aphrodite/distributed/device_communicators/custom_all_reduce.py:41-300 (same constructor, same
``should_custom_ar`` / ``custom_all_reduce`` / ``capture`` / ``register_buffer`` /
``register_graph_buffers`` / ``close`` surface and eligibility rules) over
``csrc/custom_all_reduce.hip`` instead of the ``_C_custom_ar`` ops the reference compiles out on ROCm.

Differences that are MI355X decisions, not omissions:
  * the ``meta`` buffer (signal area + two-shot scratch) is fine-grained uncached device memory wrapped in a tensor
    (``ops.custom_ar_alloc_meta``; ``torch.zeros`` memory is coarse-grained); IPC handles are taken with
    hipIpcGetMemHandle (``ops.ipc_handle_of``), which also works for that memory, instead of ``storage._share_cuda_()``;
    everything else goes through the ``_C_custom_ar`` op surface of ``_custom_ops.py`` (init_custom_ar, register_buffer,
    all_reduce_reg / _unreg, get_graph_buffer_ipc_meta, register_graph_buffers, dispose, meta_size);
  * every node pair of an MI355X box is one xGMI hop, so the reference's NVLink-topology probe
    (``is_full_nvlink``) reduces to "same node + peer access": the constructor all-gathers (hostname, boot id,
    physical device) over the CPU group and asks the runtime ``can_device_access_peer`` for every pair; if any
    rank is on another node or a pair lacks peer access the communicator stays ``disabled`` and the caller falls
    back to RCCL, exactly as the reference does (custom_all_reduce.py:73-77, 133-146);
  * a peer that never arrives raises ``RuntimeError`` from ``check()`` (bounded spin in the
    kernel) instead of hanging the GPU; eager calls poll the error word every
    ``APHRODITE_CUSTOM_AR_CHECK_EVERY`` calls (default 256; the poll is a blocking 4-byte read), graph
    replays are checked by whoever replays (bench.py, the tests)."""
import ctypes
import os
import socket
from contextlib import contextmanager
from typing import List, Optional, Tuple, Union

import torch
import torch.distributed as dist

from ..switches import switch
from .. import _lib
from .._lib import check

_DT = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}


def is_weak_contiguous(inp: torch.Tensor) -> bool:
    """custom_all_reduce.py:35-38: contiguous, or a dense view ending its storage."""
    return inp.is_contiguous() or (inp.untyped_storage().nbytes() - inp.storage_offset() * inp.element_size()
                                   == inp.numel() * inp.element_size())


class CustomAllreduce:
    _SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]

    def __init__(self, group: dist.ProcessGroup, device: Union[int, str, torch.device],
                 max_size: int = 8192 * 1024, ops=None) -> None:
        """``ops``: the provider of the ``_C_custom_ar`` op surface (default: ``aphrodite_engine_amd._custom_ops``; the tests
        pass a view whose schema ops go through the C++ registration, ``torch.ops._C_mi355x_custom_ar``)."""
        self._IS_CAPTURING = False
        self.disabled = True
        self._ptr = None
        self.group = group
        if dist.get_backend(group) == dist.Backend.NCCL:
            raise AssertionError("CustomAllreduce should be attached to a non-NCCL group.")
        rank = dist.get_rank(group=group)
        world_size = dist.get_world_size(group=group)
        if world_size == 1 or world_size not in self._SUPPORTED_WORLD_SIZES:
            return
        if isinstance(device, int):
            device = torch.device(f"cuda:{device}")
        elif isinstance(device, str):
            device = torch.device(device)
        self.device = device
        self.rank, self.world_size, self.max_size = rank, world_size, max_size
        self._calls = 0
        self._check_every = int((switch("APHRODITE_CUSTOM_AR_CHECK_EVERY") or "256"))
        ok, why = self._peers_eligible(device)
        self.full_nvlink = ok              # one xGMI hop between any two GPUs of the node
        if not ok:
            if rank == 0:
                import warnings
                warnings.warn(f"custom all-reduce is disabled ({why}); tensor parallelism falls back to RCCL")
            return
        if ops is None:
            from .. import _custom_ops as ops
        self._ops = ops
        with torch.cuda.device(device):
            # custom_all_reduce.py:101-120: meta = signal area + two-shot scratch, a staging buffer for unregistered
            # inputs, the table of registered buffers -- all through the _C_custom_ar op surface (_custom_ops.py)
            self.meta = ops.custom_ar_alloc_meta(ops.meta_size() + max_size, device)
            self.buffer = torch.empty(max_size, dtype=torch.uint8, device=device)
            self.rank_data = torch.empty(8 * 1024 * 1024, dtype=torch.uint8, device=device)
            handles, offsets = self._get_ipc_meta(self.meta)
            self._ptr = ops.init_custom_ar(self.meta, self.rank_data, handles, offsets, rank, self.full_nvlink)
            self.disabled = False
            self.register_buffer(self.buffer)

    # -- eligibility (custom_all_reduce.py:73-77, 112-146; in_the_same_node_as, _can_p2p) ------------
    def _peers_eligible(self, device: torch.device):
        try:
            boot = open("/proc/sys/kernel/random/boot_id").read().strip()
        except OSError:
            boot = ""
        visible = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
        try:
            ids = [int(v) for v in visible.split(",")] if visible else None
        except ValueError:
            # UUID-style entries ("GPU-..."): physical ordinals unknown, peers cannot be matched -> RCCL
            return False, f"device mask {visible!r} is not a list of ordinals"
        phys = ids[device.index] if ids is not None and device.index < len(ids) else device.index
        mine = (socket.gethostname(), boot, int(phys), int(device.index))
        everyone: List[Optional[tuple]] = [None] * self.world_size
        dist.all_gather_object(everyone, mine, group=self.group)
        if any((e[0], e[1]) != (mine[0], mine[1]) for e in everyone):
            return False, "the process group spans several nodes"
        local_ok = True
        if ids is None:            # peers' local indices are meaningful in this process only without a device mask
            for e in everyone:
                if e[3] != device.index and e[3] < torch.cuda.device_count():
                    local_ok &= bool(torch.cuda.can_device_access_peer(device.index, e[3]))
        verdicts: List[Optional[bool]] = [None] * self.world_size
        dist.all_gather_object(verdicts, local_ok, group=self.group)
        if not all(verdicts):
            return False, "a GPU pair lacks peer access"
        return True, ""

    # -- IPC plumbing (custom_all_reduce.py:206-245) -----------------------------------------------------
    def _get_ipc_meta(self, inp: torch.Tensor):
        return self._gather_ipc_meta(self._ops.ipc_handle_of(inp))

    def _gather_ipc_meta(self, shard_data):
        """Every rank's (handle, offset) in group-rank order."""
        everyone: List[Optional[tuple]] = [None] * self.world_size
        dist.all_gather_object(everyone, shard_data, group=self.group)
        return [e[0] for e in everyone], [e[1] for e in everyone]

    def register_buffer(self, inp: torch.Tensor) -> None:
        handles, offsets = self._get_ipc_meta(inp)
        self._ops.register_buffer(self._ptr, inp, handles, offsets)

    def register_graph_buffers(self) -> None:
        handle, offset = self._ops.get_graph_buffer_ipc_meta(self._ptr)
        handles, offsets = self._gather_ipc_meta((bytes(handle), offset))
        if any(len(o) != len(offset) for o in offsets):
            raise RuntimeError("custom all-reduce: ranks captured different numbers of graph buffers")
        self._ops.register_graph_buffers(self._ptr, handles, offsets)

    @contextmanager
    def capture(self):
        """Wrap HIP-graph captures: the inputs seen while capturing are registered at the end."""
        try:
            self._IS_CAPTURING = True
            yield
        finally:
            self._IS_CAPTURING = False
            if not self.disabled:
                self.register_graph_buffers()

    # -- the op ------------------------------------------------------------------------------------
    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        if self.disabled or inp.dtype not in _DT:
            return False
        inp_size = inp.numel() * inp.element_size()
        if inp_size % 16 != 0 or inp.data_ptr() % 16 != 0:
            return False
        if not is_weak_contiguous(inp):
            return False
        return inp_size < self.max_size

    def all_reduce_reg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty_like(inp)
        self._ops.all_reduce_reg(self._ptr, inp, out)
        return out

    def all_reduce_unreg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty_like(inp)
        self._ops.all_reduce_unreg(self._ptr, inp, self.buffer, out)
        return out

    def custom_all_reduce(self, input: torch.Tensor) -> Optional[torch.Tensor]:
        """None = not eligible, the caller falls back to RCCL (custom_all_reduce.py:268-289)."""
        if self.disabled or not self.should_custom_ar(input):
            return None
        if self._IS_CAPTURING:
            if torch.cuda.is_current_stream_capturing():
                return self.all_reduce_reg(input)
            return torch.empty_like(input)          # warm-up run before the capture: shape only
        out = self.all_reduce_unreg(input)
        self._calls += 1
        if self._check_every > 0 and self._calls % self._check_every == 0:
            self.check()                            # a timed-out barrier must not return garbage silently for long
        return out

    # -- all-reduce + residual add + RMSNorm (+ pack) in one launch -------------------------------------
    def fused_norm_eligible(self, inp: torch.Tensor) -> bool:
        return (self.should_custom_ar(inp) and inp.dim() == 2 and inp.shape[0] <= 64 and inp.shape[1] % 128 == 0
                and inp.shape[1] <= 16384 and inp.dtype in (torch.float16, torch.bfloat16) and inp.is_contiguous()
                and inp.numel() * inp.element_size() <= self.max_size)

    def fused_add_rms_norm(self, inp: torch.Tensor, residual: Optional[torch.Tensor], has_residual: bool,
                           weight: torch.Tensor, epsilon: float, pack: bool = True, want_out: bool = False,
                           prefetch: Optional[torch.Tensor] = None):
        """custom_all_reduce(inp) followed by ops.fused_add_rms_norm_pack(out, None, residual, ...) in ONE launch, same
        bits (csrc/custom_all_reduce.hip).  None = not eligible: the caller runs the two ops.  Returns (packed, out)."""
        if self.disabled or not self.fused_norm_eligible(inp):
            return None
        kw = dict(pack=pack, want_out=want_out, prefetch=prefetch)
        if self._IS_CAPTURING:
            if torch.cuda.is_current_stream_capturing():
                return self._ops.custom_ar_fused_add_rms_norm(self._ptr, inp, residual, has_residual, weight, epsilon, **kw)
            # warm-up run before the capture: shapes only (as custom_all_reduce does)
            return self._ops.fused_add_rms_norm_pack(torch.zeros_like(inp), None, None, False, weight, epsilon, pack=pack,
                                                     want_out=want_out)
        res = self._ops.custom_ar_fused_add_rms_norm(self._ptr, inp, residual, has_residual, weight, epsilon,
                                                     reg_buffer=self.buffer, **kw)
        self._calls += 1
        if self._check_every > 0 and self._calls % self._check_every == 0:
            self.check()
        return res

    def fused_add_rms_norm_quant_fp8(self, inp: torch.Tensor, residual: Optional[torch.Tensor], has_residual: bool,
                                     weight: torch.Tensor, epsilon: float, want_out: bool = False,
                                     static_scale: Optional[torch.Tensor] = None):
        """custom_all_reduce(inp) followed by ops.fused_add_rms_norm_quant_fp8(out, None, None, None, residual, ...) in
        ONE launch, same bits.  None = not eligible.  Returns (q, scales, out)."""
        if self.disabled or not self.fused_norm_eligible(inp):
            return None
        kw = dict(want_out=want_out, static_scale=static_scale)
        if self._IS_CAPTURING:
            if torch.cuda.is_current_stream_capturing():
                return self._ops.custom_ar_fused_add_rms_norm_quant_fp8(self._ptr, inp, residual, has_residual, weight,
                                                                        epsilon, **kw)
            # warm-up run before the capture: shapes only (the residual is left alone)
            return self._ops.fused_add_rms_norm_quant_fp8(torch.zeros_like(inp), None, None, None, torch.empty_like(inp),
                                                          False, weight, epsilon, **kw)
        res = self._ops.custom_ar_fused_add_rms_norm_quant_fp8(self._ptr, inp, residual, has_residual, weight, epsilon,
                                                               reg_buffer=self.buffer, **kw)
        self._calls += 1
        if self._check_every > 0 and self._calls % self._check_every == 0:
            self.check()
        return res

    def _router_form_pays(self, inp: torch.Tensor) -> bool:
        """The one-launch all-reduce + norm + router form against its two launches on the loopback rig (tools/ar_router_bench.py,
        us): 4 ranks [32, 4096] 7.4 / 8.5, 8 ranks [64, 8192] (two-shot) 10.4 / 14.0 -- but 8 ranks [32, 4096] in the one-shot
        form 9.6 / 8.9: every workgroup reads the row from eight peers AND the router rows.  Not used there."""
        one_shot = self._ops.custom_ar_fused_norm_one_shot(self.world_size, inp.shape[0], inp.shape[1], inp.element_size())
        return not (one_shot and self.world_size > 4)

    def fused_add_rms_norm_router(self, inp: torch.Tensor, residual: Optional[torch.Tensor], has_residual: bool,
                                  weight: torch.Tensor, epsilon: float, router_weight: torch.Tensor):
        """custom_all_reduce(inp) followed by ops.fused_add_rms_norm_router(out, None, residual, ...) in ONE launch, same
        bits.  None = not eligible.  Returns (normed [tokens, hidden], router_logits [tokens, E])."""
        if self.disabled or not self.fused_norm_eligible(inp) or router_weight.shape[0] > 16 \
                or not self._router_form_pays(inp):
            return None
        if self._IS_CAPTURING:
            if torch.cuda.is_current_stream_capturing():
                return self._ops.custom_ar_fused_add_rms_norm_router(self._ptr, inp, residual, has_residual, weight, epsilon,
                                                                     router_weight)
            # warm-up run before the capture: shapes only (the residual is left alone)
            return self._ops.fused_add_rms_norm_router(torch.zeros_like(inp), None, torch.empty_like(inp), False, weight,
                                                       epsilon, router_weight)
        res = self._ops.custom_ar_fused_add_rms_norm_router(self._ptr, inp, residual, has_residual, weight, epsilon,
                                                            router_weight, reg_buffer=self.buffer)
        self._calls += 1
        if self._check_every > 0 and self._calls % self._check_every == 0:
            self.check()
        return res

    def check(self) -> None:
        """Raise if one of this rank's barriers timed out since the last call."""
        if not self.disabled and self._ops.custom_ar_error(self._ptr):
            raise RuntimeError("custom all-reduce: a peer did not arrive at a barrier (results are invalid)")

    def close(self) -> None:
        if self._ptr:
            torch.cuda.synchronize(self.device)
            self._ops.dispose(self._ptr)
            self._ptr = None
            self.meta = None
            self.disabled = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LoopbackAllreduce:
    """ONE rank of a ``world_size`` TP group on a one-GPU box (bench.py --sim-tp): the CustomAllreduce surface over a
    loopback communicator (ops.init_custom_ar_loopback) -- the all-reduce and fused all-reduce + norm kernels run their
    real instruction stream, flag protocol and scratch traffic against this process's own buffers, so the launch costs
    what the peer-access kernel costs with zero link time.  The sums are over ``world_size`` copies of the local partial:
    timing only.  No process group, nothing to register, capturable as is."""

    def __init__(self, world_size: int, device, max_size: int = 8192 * 1024) -> None:
        from .. import _custom_ops as ops
        self._ops = ops
        self.device = torch.device(device)
        self.rank, self.world_size, self.max_size = 0, world_size, max_size
        self.disabled = False
        self._IS_CAPTURING = False
        with torch.cuda.device(self.device):
            self.meta = ops.custom_ar_alloc_meta(ops.meta_size() + max_size, self.device)
            self.rank_data = torch.empty(64 * 1024, dtype=torch.uint8, device=self.device)
            self._ptr = ops.init_custom_ar_loopback(self.meta, self.rank_data, world_size)

    should_custom_ar = CustomAllreduce.should_custom_ar
    fused_norm_eligible = CustomAllreduce.fused_norm_eligible

    @contextmanager
    def capture(self):
        yield

    def custom_all_reduce(self, input: torch.Tensor) -> Optional[torch.Tensor]:
        if not self.should_custom_ar(input):
            return None
        out = torch.empty_like(input)
        self._ops.all_reduce_reg(self._ptr, input, out)
        return out

    def fused_add_rms_norm(self, inp, residual, has_residual, weight, epsilon, pack=True, want_out=False, prefetch=None):
        if not self.fused_norm_eligible(inp):
            return None
        return self._ops.custom_ar_fused_add_rms_norm(self._ptr, inp, residual, has_residual, weight, epsilon, pack=pack,
                                                      want_out=want_out, prefetch=prefetch)

    def fused_add_rms_norm_quant_fp8(self, inp, residual, has_residual, weight, epsilon, want_out=False, static_scale=None):
        if not self.fused_norm_eligible(inp):
            return None
        return self._ops.custom_ar_fused_add_rms_norm_quant_fp8(self._ptr, inp, residual, has_residual, weight, epsilon,
                                                                want_out=want_out, static_scale=static_scale)

    def fused_add_rms_norm_router(self, inp, residual, has_residual, weight, epsilon, router_weight):
        if not self.fused_norm_eligible(inp) or router_weight.shape[0] > 16 \
                or not CustomAllreduce._router_form_pays(self, inp):
            return None
        return self._ops.custom_ar_fused_add_rms_norm_router(self._ptr, inp, residual, has_residual, weight, epsilon,
                                                             router_weight)

    def check(self) -> None:
        if self._ptr and self._ops.custom_ar_error(self._ptr):
            raise RuntimeError("loopback all-reduce: a barrier timed out")

    def close(self) -> None:
        if self._ptr:
            torch.cuda.synchronize(self.device)
            self._ops.dispose(self._ptr)
            self._ptr = None
            self.meta = None
            self.disabled = True
