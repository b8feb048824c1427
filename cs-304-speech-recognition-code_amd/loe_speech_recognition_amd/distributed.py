"""Multi-GPU plumbing: one process per GPU, utterances sharded in contiguous ranges, RCCL only to
gather decoded label records and to sum training statistics.  Replaces the reference's host-side
ProcessPoolExecutor fan-out (scripts/project5_test_ndigits_with_sil.py:29-42,
hidden_markov_model.py:746-750).

Two transports carry the same two exchange steps:
  * ``NativeComm`` - the library's own entry points (lsr_allgather_labels / lsr_allreduce_stats) on an
    RCCL communicator it creates itself; torch.distributed is only used once, to hand rank 0's
    ncclUniqueId to the other ranks.  When one has been made current (``NativeComm.from_torch_group``
    or ``set_native_comm``), the module functions below use it for device tensors.
  * torch.distributed collectives ("nccl" = RCCL on ROCm for device tensors, "gloo" for CPU tensors as
    used by the tests), with the records packed by the library's host functions, so both transports
    share one record layout.
All functions are the identity when no process group is initialised."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np


def _dist():
    import torch.distributed as dist

    return dist


def is_active() -> bool:
    dist = _dist()
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank_world() -> Tuple[int, int]:
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of `rank`; sizes differ by at most one, earlier ranks get the extra item
    (the rule of lsr_shard_bounds)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


# ---- record layout on the host (liblsr_hip host functions; no GPU needed) --------------------------------
def pack_label_records(labels: np.ndarray, n_labels: np.ndarray, shard_cap: int) -> np.ndarray:
    """uint8 [shard_cap, cap + 4]: label bytes then the little-endian int32 count; unused records 0xFF."""
    from . import _native

    lib = _native.load_library()
    labels = np.ascontiguousarray(labels, dtype=np.uint8)
    n_labels = np.ascontiguousarray(n_labels, dtype=np.int32)
    cap = labels.shape[1]
    out = np.empty((shard_cap, int(lib.lsr_label_record_bytes(cap))), np.uint8)
    rc = lib.lsr_host_pack_label_records(ctypes.c_void_p(labels.ctypes.data), ctypes.c_void_p(n_labels.ctypes.data),
                                         labels.shape[0], cap, shard_cap, ctypes.c_void_p(out.ctypes.data))
    if rc != 0:
        raise ValueError("lsr_host_pack_label_records: bad argument")
    return out


def unpack_label_records(records: np.ndarray, world: int, n_total: int, cap: int) -> Tuple[np.ndarray, np.ndarray]:
    """(uint8 [n_total, cap], int32 [n_total]) in global utterance order from world * shard_cap records."""
    from . import _native

    lib = _native.load_library()
    records = np.ascontiguousarray(records, dtype=np.uint8)
    shard_cap = shard_bounds(n_total, 0, world)[1]
    assert records.size == world * shard_cap * (cap + 4)
    labels = np.empty((n_total, cap), np.uint8)
    n_labels = np.empty(n_total, np.int32)
    rc = lib.lsr_host_unpack_label_records(ctypes.c_void_p(records.ctypes.data), world, shard_cap, n_total, cap,
                                           ctypes.c_void_p(labels.ctypes.data), ctypes.c_void_p(n_labels.ctypes.data))
    if rc != 0:
        raise ValueError("lsr_host_unpack_label_records: bad argument")
    return labels, n_labels


# ---- the library's own RCCL communicator --------------------------------------------------------------------
_native_comm: Optional["NativeComm"] = None


def set_native_comm(comm: Optional["NativeComm"]) -> None:
    global _native_comm
    _native_comm = comm


class NativeComm:
    """lsr_comm_init / lsr_allgather_labels / lsr_allreduce_stats on an engine's context."""

    def __init__(self, eng, unique_id: bytes, rank: int, world: int):
        self.eng, self.rank, self.world = eng, rank, world
        buf = ctypes.create_string_buffer(unique_id, 128)
        eng._check(eng.lib.lsr_comm_init(eng.ctx, buf, rank, world))
        self.open = True

    @staticmethod
    def unique_id(eng) -> bytes:
        buf = ctypes.create_string_buffer(128)
        eng._check(eng.lib.lsr_comm_get_unique_id(eng.ctx, buf))
        return buf.raw

    @classmethod
    def from_torch_group(cls, eng, make_current: bool = True) -> "NativeComm":
        """Bootstrap from the default torch.distributed group: rank 0's id is broadcast once."""
        dist = _dist()
        rank, world = rank_world()
        box = [cls.unique_id(eng) if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        comm = cls(eng, box[0], rank, world)
        if make_current:
            set_native_comm(comm)
        return comm

    def gather_label_records(self, labels, n_labels, n_total: int):
        t = self.eng.torch
        cap = labels.shape[1]
        out_l = self.eng.empty((n_total, cap), t.uint8)
        out_n = self.eng.empty((n_total,), t.int32)
        labels, n_labels = labels.contiguous(), n_labels.contiguous()
        self.eng._check(self.eng.lib.lsr_allgather_labels(self.eng.ctx, self.eng._ptr(labels), self.eng._ptr(n_labels),
                                                          labels.shape[0], cap, n_total, self.eng._ptr(out_l),
                                                          self.eng._ptr(out_n)))
        return out_l, out_n

    def all_reduce_stats(self, stats):
        assert stats.dtype == self.eng.torch.float64 and stats.is_contiguous()
        self.eng._check(self.eng.lib.lsr_allreduce_stats(self.eng.ctx, self.eng._ptr(stats), stats.numel()))
        return stats

    def close(self) -> None:
        global _native_comm
        if self.open:
            self.eng._check(self.eng.lib.lsr_comm_destroy(self.eng.ctx))
            self.open = False
        if _native_comm is self:
            _native_comm = None


# ---- the two exchange steps ------------------------------------------------------------------------------------
def all_reduce_stats(stats):
    """Sum the packed float64 statistics (count, sum_x, sum_xx, transition counts) over all ranks:
    one all-reduce per training iteration; every rank then runs the same host M-step."""
    if _native_comm is not None and _native_comm.world > 1 and stats.is_cuda:
        return _native_comm.all_reduce_stats(stats)
    if is_active():
        dist = _dist()
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats


def all_gather_strided(rows: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Complete a float64 [n, k] table of which every rank filled rows rank, rank + world, ...: one all-gather
    of the ranks' own rows (padded to the same count), returned in row order on every rank.  Bits travel
    unchanged (no arithmetic on the way).  Device tensors carry the exchange under the "nccl" backend."""
    import torch

    dist = _dist()
    n, k = rows.shape
    per = (n + world - 1) // world
    mine = np.zeros((per, k), dtype=np.float64)
    own = rows[rank::world]
    mine[:own.shape[0]] = own
    send = torch.from_numpy(mine)
    on_device = dist.get_backend() == "nccl"
    if on_device:
        send = send.cuda()
    recv = torch.empty((world * per, k), dtype=torch.float64, device=send.device)
    dist.all_gather_into_tensor(recv, send)
    got = (recv.cpu() if on_device else recv).numpy().reshape(world, per, k)
    out = np.empty((n, k), dtype=np.float64)
    for r in range(world):
        cnt = len(range(r, n, world))
        out[r::world] = got[r, :cnt]
    return out


def gather_label_records(labels, n_labels, n_total: int):
    """All ranks' fixed-width label records in global utterance order.

    labels: uint8 [B_local, cap] and n_labels: int32 [B_local] of this rank's contiguous shard
    (`shard_bounds(n_total, rank, world)`).  Returns (uint8 [n_total, cap], int32 [n_total]) on every rank.
    Shards are padded to the largest shard so that a single all-gather suffices."""
    import torch

    if _native_comm is not None and _native_comm.world > 1 and labels.is_cuda:
        return _native_comm.gather_label_records(labels, n_labels, n_total)
    if not is_active():
        return labels, n_labels
    dist = _dist()
    rank, world = rank_world()
    cap = labels.shape[1]
    biggest = shard_bounds(n_total, 0, world)[1]
    if labels.is_cuda:
        # device tensors: the library's pack / unpack kernels around torch's all-gather (RCCL)
        from . import _native

        eng = _native.get_engine()
        labels, n_labels = labels.contiguous(), n_labels.to(torch.int32).contiguous()
        packed = torch.empty((biggest, cap + 4), dtype=torch.uint8, device=labels.device)
        eng._check(eng.lib.lsr_pack_label_records(eng.ctx, eng._ptr(labels), eng._ptr(n_labels), labels.shape[0], cap,
                                                  biggest, eng._ptr(packed)))
        out = torch.empty((world * biggest, cap + 4), dtype=torch.uint8, device=labels.device)
        dist.all_gather_into_tensor(out, packed)
        g_lab = torch.empty((n_total, cap), dtype=torch.uint8, device=labels.device)
        g_cnt = torch.empty((n_total,), dtype=torch.int32, device=labels.device)
        eng._check(eng.lib.lsr_unpack_label_records(eng.ctx, eng._ptr(out), world, n_total, cap, eng._ptr(g_lab),
                                                    eng._ptr(g_cnt)))
        return g_lab, g_cnt
    packed = torch.from_numpy(pack_label_records(labels.numpy(), n_labels.numpy(), biggest))
    out = torch.empty((world * biggest, cap + 4), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, packed)
    lab, cnt = unpack_label_records(out.numpy(), world, n_total, cap)
    return torch.from_numpy(lab), torch.from_numpy(cnt)


def records_to_strings(labels: np.ndarray, n_labels: np.ndarray, names: Sequence[str]) -> List[str]:
    from . import _native

    return _native.records_to_strings(labels, n_labels, names)
