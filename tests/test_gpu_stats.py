"""lsr_accumulate_stats (counting sort by slot + float64 MFMA Gram matrices) against NumPy float64 sums."""
import numpy as np
import pytest

from loe_speech_recognition_amd import _native

pytestmark = pytest.mark.gpu


def numpy_stats(x, slots, n):
    x = x.astype(np.float64)
    count = np.zeros(n)
    sum_x = np.zeros((n, 39))
    sum_xx = np.zeros((n, 39, 39))
    for s in range(n):
        sel = x[slots == s]
        count[s] = sel.shape[0]
        sum_x[s] = sel.sum(axis=0)
        sum_xx[s] = sel.T @ sel
    return count, sum_x, sum_xx


def device_stats(x, slots, n):
    eng = _native.get_engine()
    c, sx, sxx = eng.accumulate_stats(eng.to_device(x), eng.to_device(slots.astype(np.int32)), x.shape[0], n)
    return c.cpu().numpy(), sx.cpu().numpy(), sxx.cpu().numpy()


def check(x, slots, n):
    c, sx, sxx = device_stats(x, slots, n)
    wc, wsx, wsxx = numpy_stats(x, slots, n)
    assert np.array_equal(c, wc)                                   # counts are exact
    scale_x = np.maximum(1.0, np.abs(x).astype(np.float64).sum(axis=0))
    assert np.max(np.abs(sx - wsx) / scale_x) < 1e-13
    scale = np.maximum(1.0, np.einsum("i,j->ij", scale_x, np.abs(x).max(axis=0).astype(np.float64)))
    assert np.max(np.abs(sxx - wsxx) / scale) < 1e-13
    assert np.array_equal(sxx, np.transpose(sxx, (0, 2, 1)))       # exactly symmetric
    return c, sx, sxx


@pytest.mark.parametrize("n_frames,n_slots", [(1, 1), (3, 5), (4, 2), (1023, 7), (1025, 58), (50000, 58), (20000, 300)])
def test_runs_of_slots(n_frames, n_slots):
    rng = np.random.default_rng(n_frames + n_slots)
    x = (rng.standard_normal((n_frames, 39)) * rng.uniform(0.1, 30, size=39)).astype(np.float32)
    # runs of equal slot like an alignment produces, dropped frames (-1) and slots that never occur
    slots = np.empty(n_frames, np.int64)
    f = 0
    while f < n_frames:
        run = int(rng.integers(1, 40))
        slots[f:f + run] = rng.integers(-1, max(1, n_slots - 2))
        f += run
    check(x, slots, n_slots)


def test_single_slot_and_all_dropped():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((5000, 39)).astype(np.float32)
    check(x, np.zeros(5000, np.int64), 1)
    c, sx, sxx = check(x, np.full(5000, -1, np.int64), 4)
    assert not c.any() and not sx.any() and not sxx.any()
    check(x, np.full(5000, 9, np.int64), 4)            # out-of-range slots are ignored like -1


def test_bitwise_reproducible():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((30000, 39)).astype(np.float32)
    slots = rng.integers(-1, 58, size=30000)
    a = device_stats(x, slots, 58)
    b = device_stats(x, slots, 58)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
