"""world_size-2 checks of the multi-GPU plumbing on CPU tensors (gloo): shard bounds, the
all-gather of label records and the all-reduce of training statistics followed by the host
M-step, which must give bit-identical models on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_records(n_total, cap=8, seed=0):
    rng = np.random.default_rng(seed)
    n = rng.integers(0, cap + 3, size=n_total).astype(np.int32)
    lab = rng.integers(0, 12, size=(n_total, cap)).astype(np.uint8)
    return lab, n


def _aligned_data(seed=1):
    rng = np.random.default_rng(seed)
    frames = rng.standard_normal((400, 39)).astype(np.float32)
    state = np.sort(rng.integers(0, 5, size=400)).astype(np.int64)
    return frames, state


def _partial_stats(frames, state, n):
    x = frames.astype(np.float64)
    count = np.bincount(state, minlength=n).astype(np.float64)
    sum_x = np.zeros((n, 39))
    sum_xx = np.zeros((n, 39, 39))
    np.add.at(sum_x, state, x)
    np.add.at(sum_xx, state, x[:, :, None] * x[:, None, :])
    trans = np.bincount(state[:-1] * n + state[1:], minlength=n * n).astype(np.float64)
    return np.concatenate([count, sum_x.ravel(), sum_xx.ravel(), trans])


def _m_step(stats, n):
    from loe_speech_recognition_amd import HiddenMarkovModelTrainable

    m = HiddenMarkovModelTrainable("w", isTqdm=False)
    m._means = np.zeros((n, 39), np.float32)
    m._covariances = m._init_covariance(39, n)
    d = 39
    m._m_step(stats[:n], stats[n:n * (1 + d)].reshape(n, d), stats[n * (1 + d):n * (1 + d + d * d)].reshape(n, d, d),
              stats[n * (1 + d + d * d):].reshape(n, n))
    return m._means, m._covariances, m._transition_probs.to_array()


def _word_stats(seed, sizes=(5, 3, 4)):
    """Reduced statistics (count, sum_x, sum_xx, transition counts) of a few word models from random frames."""
    rng = np.random.default_rng(seed)
    n = sum(sizes)
    count = rng.integers(60, 200, size=n).astype(np.float64)
    sum_x = np.zeros((n, 39))
    sum_xx = np.zeros((n, 39, 39))
    for s in range(n):
        x = (rng.standard_normal((int(count[s]), 39)) * rng.uniform(0.5, 3, size=39) + rng.uniform(-2, 2, size=39)).astype(np.float32)
        x = x.astype(np.float64)
        sum_x[s] = x.sum(axis=0)
        sum_xx[s] = x.T @ x
    trans = np.zeros((n, n))
    lo = 0
    for k in sizes:
        for i in range(k):
            trans[lo + i, lo + i] = rng.integers(5, 50)
            if i + 1 < k:
                trans[lo + i, lo + i + 1] = rng.integers(1, 9)
        lo += k
    return sizes, count, sum_x, sum_xx, trans


def _continuous_trainer(sizes):
    from loe_speech_recognition_amd import HiddenMarkovModelTrainable, HiddenMarkovModelTrainContinuous, TransitionProbabilities

    tc = HiddenMarkovModelTrainContinuous(isTqdm=False)
    for label, k in zip("ABC", sizes):
        m = HiddenMarkovModelTrainable(label, isTqdm=False)
        m._means = np.zeros((k, 39), np.float32)
        m._covariances = m._init_covariance(39, k)
        m._transition_probs = TransitionProbabilities.from_num_of_states(k)
        m._update_inference_weights()
        tc.add_model(m)
    return tc


def _m_steps_of_trainer(tc, stats):
    sizes, count, sum_x, sum_xx, trans = stats
    base, _ = tc._layout()
    tc._m_steps(base, count, sum_x, sum_xx, trans)
    g = [g for lab in tc._models_loaded for g in tc._trainable_models[lab]._multivariate_normals]
    return (np.stack([x.whiten for x in g]), np.array([x.log_pdet for x in g]), np.stack([x.tri for x in g]),
            np.stack([x.mean for x in g]))


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from loe_speech_recognition_amd import distributed as D

    assert D.is_active() and D.rank_world() == (rank, world)
    # label records: 11 utterances over 2 ranks (6 + 5)
    lab, n = _fake_records(11)
    lo, hi = D.shard_bounds(11, rank, world)
    g_lab, g_n = D.gather_label_records(torch.from_numpy(lab[lo:hi]), torch.from_numpy(n[lo:hi]), 11)
    assert np.array_equal(g_lab.numpy(), lab) and np.array_equal(g_n.numpy(), n)
    # the strings every rank reads from the gathered records are the single-rank strings
    names = list("123456789OSZ")
    assert D.records_to_strings(g_lab.numpy(), g_n.numpy(), names) == D.records_to_strings(lab, n, names)
    # strong-scaling shape: one corpus of 10 utterances sharded 5 + 5, and an uneven 3-way split seen from 2 ranks' shards
    lab10, n10 = _fake_records(10, seed=5)
    lo, hi = D.shard_bounds(10, rank, world)
    g_lab, g_n = D.gather_label_records(torch.from_numpy(lab10[lo:hi]), torch.from_numpy(n10[lo:hi]), 10)
    assert D.records_to_strings(g_lab.numpy(), g_n.numpy(), names) == D.records_to_strings(lab10, n10, names)
    # statistics: each rank reduces its half of the frames (cut at a state-run boundary is not required)
    frames, state = _aligned_data()
    lo, hi = D.shard_bounds(400, rank, world)
    part = _partial_stats(frames[lo:hi], state[lo:hi], 5)
    if rank == 1:                       # the transition that crosses the shard boundary belongs to exactly one rank
        part[-25:] += np.bincount([state[lo - 1] * 5 + state[lo]], minlength=25)
    total = D.all_reduce_stats(torch.from_numpy(part)).numpy()
    means, cov, trans = _m_step(total, 5)
    # M-step of a continuous trainer with the per-state factorisation dealt out over the ranks (rank r factors
    # states r, r + 2, ...; one all-gather) and, for comparison, with every rank factoring every state
    stats = _word_stats(11)
    tc = _continuous_trainer(stats[0])
    assert tc.shard_factorization
    sharded = _m_steps_of_trainer(tc, stats)
    tc2 = _continuous_trainer(stats[0])
    tc2.shard_factorization = False
    full = _m_steps_of_trainer(tc2, stats)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), means=means, cov=cov, trans=trans, total=total,
             **{f"sharded{i}": a for i, a in enumerate(sharded)}, **{f"full{i}": a for i, a in enumerate(full)})
    dist.destroy_process_group()


def test_two_rank_gather_and_allreduce(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for k in ("means", "cov", "trans", "total"):
        assert np.array_equal(r0[k], r1[k], equal_nan=True), k          # every rank holds the same model
    # the dealt-out factorisation gives every rank the bits of the serial path (SciPy's eigendecomposition state by state)
    from loe_speech_recognition_amd import MultivariateNormal

    stats = _word_stats(11)
    tc = _continuous_trainer(stats[0])
    here = _m_steps_of_trainer(tc, stats)
    covs = np.concatenate([tc._trainable_models[lab]._covariances for lab in tc._models_loaded])
    mus = np.concatenate([tc._trainable_models[lab]._means for lab in tc._models_loaded])
    serial = [MultivariateNormal.from_means_covariances(m, c) for m, c in zip(mus, covs)]
    for i in range(4):
        for r in (r0, r1):
            assert np.array_equal(r[f"sharded{i}"], here[i]) and np.array_equal(r[f"full{i}"], here[i]), i
    assert np.array_equal(here[0], np.stack([g.whiten for g in serial]))
    assert np.array_equal(here[1], np.array([g.log_pdet for g in serial]))
    assert np.array_equal(here[2], np.tril(np.swapaxes(np.linalg.qr(np.swapaxes(here[0], 1, 2), mode="r"), 1, 2)))
    frames, state = _aligned_data()
    single = _partial_stats(frames, state, 5)
    assert np.allclose(r0["total"], single, rtol=1e-12, atol=1e-9)
    means, cov, trans = _m_step(single, 5)
    assert np.allclose(r0["means"], means, atol=1e-6) and np.allclose(r0["cov"], cov, atol=1e-6)
    assert np.array_equal(r0["trans"], trans, equal_nan=True)
    # the M-step agrees with the oracle's M-step on the same alignment
    from oracle import hmm_oracle as O

    o_means, o_cov, o_trans = O.m_step(5, [(frames, state.astype(np.int8))], np.zeros((5, 39), np.float32),
                                       (np.tile(np.eye(39), (5, 1, 1)) * 0.01).astype(np.float32))
    assert np.allclose(means, o_means, atol=1e-5) and np.allclose(cov, o_cov, atol=1e-5)
    assert np.array_equal(np.nan_to_num(trans), np.nan_to_num(o_trans))


def test_shard_bounds_cover_everything():
    from loe_speech_recognition_amd import distributed as D

    for n in (0, 1, 7, 10000, 10001):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert D.records_to_strings(np.array([[1, 2, 3]], np.uint8), np.array([2], np.int32), list("0123")) == ["12"]


@pytest.mark.parametrize("n_total,world", [(11, 2), (10, 2), (7, 3), (3, 8), (1, 1), (10000, 8), (10001, 4)])
def test_record_layout_round_trip(n_total, world):
    """lsr_host_pack_label_records per rank, concatenated as an all-gather would, then
    lsr_host_unpack_label_records: the original records in global order; lsr_shard_bounds is the Python rule."""
    import ctypes

    from loe_speech_recognition_amd import _native
    from loe_speech_recognition_amd import distributed as D

    lib = _native.load_library()
    cap = 6
    lab, n = _fake_records(n_total, cap=cap, seed=n_total + world)
    n[0] = -1 if n_total > 1 else n[0]                     # counts travel as raw int32
    shard_cap = D.shard_bounds(n_total, 0, world)[1]
    pieces = []
    for r in range(world):
        lo, hi = D.shard_bounds(n_total, r, world)
        c_lo, c_hi = ctypes.c_int64(-1), ctypes.c_int64(-1)
        lib.lsr_shard_bounds(n_total, r, world, ctypes.byref(c_lo), ctypes.byref(c_hi))
        assert (c_lo.value, c_hi.value) == (lo, hi)
        piece = D.pack_label_records(lab[lo:hi], n[lo:hi], shard_cap)
        assert piece.shape == (shard_cap, cap + 4) and (piece[hi - lo:] == 0xFF).all()
        if hi > lo:
            assert np.array_equal(piece[0, :cap], lab[lo]) and piece[0, cap:].view(np.int32)[0] == n[lo]
        pieces.append(piece)
    g_lab, g_n = D.unpack_label_records(np.concatenate(pieces), world, n_total, cap)
    assert np.array_equal(g_lab, lab) and np.array_equal(g_n, n)
    with pytest.raises(ValueError):
        D.pack_label_records(lab, n, max(0, n_total - 1))
