"""CustomAllreduce -- the tensor-parallel sum over xGMI peer access, host side.  Mirror of
aphrodite/distributed/device_communicators/custom_all_reduce.py:41-300 (same constructor, same
``should_custom_ar`` / ``custom_all_reduce`` / ``capture`` / ``register_buffer`` /
``register_graph_buffers`` / ``close`` surface and eligibility rules) over
``csrc/custom_all_reduce.hip`` instead of the ``_C_custom_ar`` ops the reference compiles out on ROCm.

Differences that are MI355X decisions, not omissions:
  * the signal area and the two-shot scratch are allocated by the library as uncached fine-grained
    device memory (a torch tensor cannot be); IPC handles are taken with hipIpcGetMemHandle through
    the C ABI, not through ``storage._share_cuda_()``;
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
                 max_size: int = 8192 * 1024) -> None:
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
        self._check_every = int(os.environ.get("APHRODITE_CUSTOM_AR_CHECK_EVERY", "256"))
        ok, why = self._peers_eligible(device)
        self.full_nvlink = ok              # one xGMI hop between any two GPUs of the node
        if not ok:
            if rank == 0:
                import warnings
                warnings.warn(f"custom all-reduce is disabled ({why}); tensor parallelism falls back to RCCL")
            return
        lib = _lib.lib()
        self._hb = lib.aphro_ipc_handle_bytes()
        with torch.cuda.device(device):
            self._signal = ctypes.c_void_p()
            check(lib.aphro_custom_ar_alloc_shared(ctypes.byref(self._signal), lib.aphro_custom_ar_meta_size()),
                  "custom_ar_alloc_shared")
            self._scratch = ctypes.c_void_p()
            check(lib.aphro_custom_ar_alloc_shared(ctypes.byref(self._scratch), max_size), "custom_ar_alloc_shared")
            # staging buffer for unregistered inputs + the table of registered buffers (device memory
            # owned here and handed to the library, like self.buffer / self.rank_data in the reference)
            self.buffer = torch.empty(max_size, dtype=torch.uint8, device=device)
            self.rank_data = torch.empty(8 * 1024 * 1024, dtype=torch.uint8, device=device)
            sh, so = self._gather_ipc_meta(self._ipc_meta(self._signal.value))
            ch, co = self._gather_ipc_meta(self._ipc_meta(self._scratch.value))
            fa = ctypes.c_void_p()
            check(lib.aphro_custom_ar_init(ctypes.byref(fa), self._signal, sh, so, self._scratch, max_size, ch, co,
                                           self.rank_data.data_ptr(), self.rank_data.numel(), rank, world_size),
                  "custom_ar_init")
            self._ptr = fa
            self.disabled = False
            self.register_buffer(self.buffer)

    # -- eligibility (custom_all_reduce.py:73-77, 112-146; in_the_same_node_as, _can_p2p) ------------
    def _peers_eligible(self, device: torch.device):
        try:
            boot = open("/proc/sys/kernel/random/boot_id").read().strip()
        except OSError:
            boot = ""
        visible = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
        ids = [int(v) for v in visible.split(",")] if visible else None
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

    # -- IPC plumbing ------------------------------------------------------------------------------
    def _ipc_meta(self, ptr: int) -> Tuple[bytes, int]:
        h = ctypes.create_string_buffer(self._hb)
        off = ctypes.c_int64()
        check(_lib.lib().aphro_ipc_get_mem_handle(ctypes.c_void_p(ptr), h, ctypes.byref(off)), "ipc_get_mem_handle")
        return bytes(h.raw), int(off.value)

    def _gather_ipc_meta(self, mine):
        """Every rank's (handles, offsets) in group-rank order, as C arrays."""
        everyone: List[Optional[tuple]] = [None] * self.world_size
        dist.all_gather_object(everyone, mine, group=self.group)
        handles = b"".join(e[0] for e in everyone)
        offsets = (ctypes.c_int64 * self.world_size)(*[e[1] for e in everyone])
        return ctypes.create_string_buffer(handles, len(handles)), offsets

    def register_buffer(self, inp: torch.Tensor) -> None:
        h, o = self._gather_ipc_meta(self._ipc_meta(inp.data_ptr()))
        check(_lib.lib().aphro_custom_ar_register_buffer(self._ptr, inp.data_ptr(), h, o), "custom_ar_register_buffer")

    def register_graph_buffers(self) -> None:
        lib = _lib.lib()
        n = ctypes.c_int()
        check(lib.aphro_custom_ar_get_graph_buffer_ipc_meta(self._ptr, None, None, 0, ctypes.byref(n)),
              "custom_ar_get_graph_buffer_ipc_meta")
        count = n.value
        hbuf = ctypes.create_string_buffer(max(1, count * self._hb))
        obuf = (ctypes.c_int64 * max(1, count))()
        if count:
            check(lib.aphro_custom_ar_get_graph_buffer_ipc_meta(self._ptr, hbuf, obuf, count, ctypes.byref(n)),
                  "custom_ar_get_graph_buffer_ipc_meta")
        everyone: List[Optional[tuple]] = [None] * self.world_size
        dist.all_gather_object(everyone, (bytes(hbuf.raw[:count * self._hb]), list(obuf[:count])), group=self.group)
        if any(len(e[1]) != count for e in everyone):
            raise RuntimeError("custom all-reduce: ranks captured different numbers of graph buffers")
        handles = b"".join(e[0] for e in everyone)                    # rank-major [world][count]
        offsets = (ctypes.c_int64 * max(1, self.world_size * count))(*[o for e in everyone for o in e[1]])
        check(lib.aphro_custom_ar_register_graph_buffers(
            self._ptr, ctypes.create_string_buffer(handles, max(1, len(handles))), offsets, count),
            "custom_ar_register_graph_buffers")

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

    def _run(self, inp: torch.Tensor, out: Optional[torch.Tensor], staged: bool) -> torch.Tensor:
        if out is None:
            out = torch.empty_like(inp)
        check(_lib.lib().aphro_custom_ar_all_reduce(
            self._ptr, inp.data_ptr(), out.data_ptr(), inp.numel(), _DT[inp.dtype],
            self.buffer.data_ptr() if staged else None, self.buffer.numel() if staged else 0,
            torch.cuda.current_stream().cuda_stream), "custom_ar_all_reduce")
        return out

    def all_reduce_reg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self._run(inp, out, staged=False)

    def all_reduce_unreg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self._run(inp, out, staged=True)

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

    def check(self) -> None:
        """Raise if one of this rank's barriers timed out since the last call."""
        if not self.disabled and _lib.lib().aphro_custom_ar_error(self._ptr) != 0:
            raise RuntimeError("custom all-reduce: a peer did not arrive at a barrier (results are invalid)")

    def close(self) -> None:
        if self._ptr:
            lib = _lib.lib()
            torch.cuda.synchronize(self.device)
            lib.aphro_custom_ar_dispose(self._ptr)
            lib.aphro_custom_ar_free_shared(self._signal)
            lib.aphro_custom_ar_free_shared(self._scratch)
            self._ptr = None
            self.disabled = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
