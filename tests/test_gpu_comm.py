"""The multi-GPU exchange steps behind the C ABI on the one GPU a test box has: the device pack / unpack kernels
against the host functions for simulated multi-rank layouts, and lsr_comm_init / lsr_allgather_labels /
lsr_allreduce_stats on a one-rank RCCL communicator (more ranks need more GPUs: RCCL refuses two ranks on one device)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _records(n_total, cap, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 12, size=(n_total, cap)).astype(np.uint8), rng.integers(0, cap + 3, size=n_total).astype(np.int32)


@pytest.mark.parametrize("n_total,world", [(11, 2), (7, 3), (3, 8), (10000, 8), (10001, 4), (5, 1)])
def test_device_pack_unpack_equal_host_layout(n_total, world):
    from loe_speech_recognition_amd import _native
    from loe_speech_recognition_amd import distributed as D

    eng = _native.get_engine()
    t = eng.torch
    cap = 16
    lab, n = _records(n_total, cap, n_total * 31 + world)
    shard_cap = D.shard_bounds(n_total, 0, world)[1]
    d_lab, d_n = eng.to_device(lab), eng.to_device(n)
    gathered = eng.empty((world, shard_cap, cap + 4), t.uint8)
    for r in range(world):
        lo, hi = D.shard_bounds(n_total, r, world)
        piece = gathered[r]
        # a rank without utterances (n_total < world) still contributes a full piece of padding records; its
        # empty tensors have a null data pointer, which the entry point accepts for an empty shard
        src_l, src_n = d_lab[lo:hi].contiguous(), d_n[lo:hi].contiguous()
        assert hi > lo or src_l.data_ptr() == 0
        eng._check(eng.lib.lsr_pack_label_records(eng.ctx, eng._ptr(src_l), eng._ptr(src_n), hi - lo, cap, shard_cap,
                                                  eng._ptr(piece)))
        eng.synchronize()
        assert np.array_equal(piece.cpu().numpy(), D.pack_label_records(lab[lo:hi], n[lo:hi], shard_cap))
    out_l, out_n = eng.empty((n_total, cap), t.uint8), eng.empty((n_total,), t.int32)
    eng._check(eng.lib.lsr_unpack_label_records(eng.ctx, eng._ptr(gathered), world, n_total, cap, eng._ptr(out_l),
                                                eng._ptr(out_n)))
    eng.synchronize()
    assert np.array_equal(out_l.cpu().numpy(), lab) and np.array_equal(out_n.cpu().numpy(), n)


def test_one_rank_communicator_gather_and_reduce():
    from loe_speech_recognition_amd import _native
    from loe_speech_recognition_amd import distributed as D

    eng = _native.get_engine()
    comm = D.NativeComm(eng, D.NativeComm.unique_id(eng), 0, 1)
    try:
        lab, n = _records(257, 16, 9)
        g_lab, g_n = comm.gather_label_records(eng.to_device(lab), eng.to_device(n), 257)
        eng.synchronize()
        assert np.array_equal(g_lab.cpu().numpy(), lab) and np.array_equal(g_n.cpu().numpy(), n)
        stats = np.random.default_rng(2).standard_normal(58 * (1 + 39 + 39 * 39) + 58 * 58)
        d = eng.to_device(stats)
        comm.all_reduce_stats(d)
        eng.synchronize()
        assert np.array_equal(d.cpu().numpy(), stats)
        # null pointers are only refused for a shard that has utterances
        rc = eng.lib.lsr_pack_label_records(eng.ctx, None, None, 3, 16, 8, eng._ptr(g_lab))
        assert rc == -1
        with pytest.raises(_native.NativeError):          # the shard size must match the communicator's rank
            comm.gather_label_records(eng.to_device(lab[:5]), eng.to_device(n[:5]), 257)
    finally:
        comm.close()
    with pytest.raises(_native.NativeError):              # no communicator any more
        eng._check(eng.lib.lsr_allreduce_stats(eng.ctx, eng._ptr(d), 4))
