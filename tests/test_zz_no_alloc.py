"""lsr_recognize_batch_noalloc (static device workspace, no hipMalloc inside the call) against
lsr_recognize_batch through the C ABI: scores, paths and label records bit-identical, on two calls in
a row, the second with a smaller batch (the shared workspace keeps the first call's contents)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LABEL_CAP = 16


def _outputs(eng, ab):
    t = eng.torch
    return (eng.empty((ab.n,), t.float32), eng.empty((ab.total_frames,), t.int8),
            eng.empty((ab.n, LABEL_CAP), t.uint8), eng.empty((ab.n,), t.int32))


def _call(eng, fn, ab, out):
    scores, paths, labels, n_labels = out
    P = ctypes.c_void_p
    rc = fn(eng.ctx, P(ab.audio.data_ptr()), P(ab.sample_off.data_ptr()), P(ab.frame_off.data_ptr()), ab.n,
            ab.total_frames, ab.max_frames, P(scores.data_ptr()), P(paths.data_ptr()), P(labels.data_ptr()),
            P(n_labels.data_ptr()), LABEL_CAP)
    eng.synchronize()
    return rc


def test_noalloc_equals_recognize_batch_twice_second_smaller():
    from loe_speech_recognition_amd import HiddenMarkovModelInference, synthetic

    models = synthetic.load_word_models()
    inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
    inf._log_transition_probability_between_words = -100
    eng = inf._bind()
    lib = eng.lib
    big, _ = synthetic.batch(3, 96, 4, start=300)
    small, _ = synthetic.batch(2, 17, 1, start=40)
    for sigs in (big, small):
        ab = eng.audio_batch([np.asarray(s, dtype=np.float32) for s in sigs])
        ref, got = _outputs(eng, ab), _outputs(eng, ab)
        for o in got:
            o.fill_(0x5A if o.dtype != eng.torch.float32 else 1.0)
        assert _call(eng, lib.lsr_recognize_batch, ab, ref) == 0
        assert _call(eng, lib.lsr_recognize_batch_noalloc, ab, got) == 0, lib.lsr_last_error(eng.ctx)
        # the call allocates nothing: free device memory is the same before and after (the call above was the
        # warm-up: kernels are loaded, the outputs exist)
        free_before = eng.torch.cuda.mem_get_info(eng.device)[0]
        assert _call(eng, lib.lsr_recognize_batch_noalloc, ab, got) == 0, lib.lsr_last_error(eng.ctx)
        assert eng.torch.cuda.mem_get_info(eng.device)[0] == free_before
        for r, g in zip(ref, got):
            r, g = r.cpu().numpy(), g.cpu().numpy()
            if r.ndim == 2:                      # label records: only the first n_labels entries are written
                n = ref[3].cpu().numpy()
                for i in range(r.shape[0]):
                    assert np.array_equal(r[i, :min(n[i], LABEL_CAP)], g[i, :min(n[i], LABEL_CAP)])
            else:
                assert np.array_equal(r.view(np.uint8), g.view(np.uint8))
        assert (ref[3].cpu().numpy() > 0).all()


def test_noalloc_calls_on_two_streams_are_ordered():
    """Two calls in a row on different streams share the static workspace: the second waits for the first on the
    device (per-GPU event), so both produce the results of lsr_recognize_batch."""
    from loe_speech_recognition_amd import HiddenMarkovModelInference, synthetic

    models = synthetic.load_word_models()
    inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
    inf._log_transition_probability_between_words = -100
    eng = inf._bind()
    lib, t = eng.lib, eng.torch
    a_sigs, _ = synthetic.batch(3, 64, 4, start=700)
    b_sigs, _ = synthetic.batch(3, 48, 2, start=900)
    a, b = (eng.audio_batch([np.asarray(s, dtype=np.float32) for s in sigs]) for sigs in (a_sigs, b_sigs))
    ref_a, ref_b, got_a, got_b = (_outputs(eng, x) for x in (a, b, a, b))
    assert _call(eng, lib.lsr_recognize_batch, a, ref_a) == 0 and _call(eng, lib.lsr_recognize_batch, b, ref_b) == 0
    side = t.cuda.Stream(device=eng.device)
    main = t.cuda.current_stream(eng.device)
    t.cuda.synchronize()
    P = ctypes.c_void_p

    def submit(ab, out):
        scores, paths, labels, n_labels = out
        return lib.lsr_recognize_batch_noalloc(eng.ctx, P(ab.audio.data_ptr()), P(ab.sample_off.data_ptr()),
                                               P(ab.frame_off.data_ptr()), ab.n, ab.total_frames, ab.max_frames,
                                               P(scores.data_ptr()), P(paths.data_ptr()), P(labels.data_ptr()),
                                               P(n_labels.data_ptr()), LABEL_CAP)
    try:
        assert submit(a, got_a) == 0                                   # on the context's stream, not waited for
        eng._check(lib.lsr_set_stream(eng.ctx, P(side.cuda_stream)))
        assert submit(b, got_b) == 0                                   # on another stream right behind it
    finally:
        t.cuda.synchronize()
        eng._check(lib.lsr_set_stream(eng.ctx, P(main.cuda_stream)))
    for ref, got in ((ref_a, got_a), (ref_b, got_b)):
        assert np.array_equal(ref[0].cpu().numpy().view(np.uint8), got[0].cpu().numpy().view(np.uint8))
        assert np.array_equal(ref[1].cpu().numpy(), got[1].cpu().numpy())
        assert np.array_equal(ref[3].cpu().numpy(), got[3].cpu().numpy())


def test_noalloc_at_both_limits_at_once():
    """4 096 utterances of 16 frames: the most utterances and the most frames the static workspace takes, which is
    also where the back-pointer groups (one partly filled group per utterance) need all of their part of it."""
    from loe_speech_recognition_amd import HiddenMarkovModelInference, synthetic

    models = synthetic.load_word_models()
    inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
    inf._log_transition_probability_between_words = -100
    eng = inf._bind()
    lib = eng.lib
    long, _ = synthetic.batch(3, 8, 2, start=1500)
    rng = np.random.default_rng(3)
    clips = []
    for i in range(4096):                                  # 15 * 160 .. 15 * 160 + 159 samples: 16 frames each
        src = np.asarray(long[i % len(long)], dtype=np.float32)
        at = int(rng.integers(0, src.shape[0] - 2600))
        clips.append(np.ascontiguousarray(src[at:at + 2400 + int(rng.integers(0, 160))]))
    ab = eng.audio_batch(clips)
    assert ab.n == 4096 and ab.total_frames == 65536 and ab.max_frames == 16
    ref, got = _outputs(eng, ab), _outputs(eng, ab)
    assert _call(eng, lib.lsr_recognize_batch, ab, ref) == 0
    assert _call(eng, lib.lsr_recognize_batch_noalloc, ab, got) == 0, lib.lsr_last_error(eng.ctx)
    assert np.array_equal(ref[0].cpu().numpy().view(np.uint8), got[0].cpu().numpy().view(np.uint8))
    assert np.array_equal(ref[1].cpu().numpy(), got[1].cpu().numpy())
    assert np.array_equal(ref[3].cpu().numpy(), got[3].cpu().numpy())
    # the workspace is intact for the next, ordinary call
    small, _ = synthetic.batch(2, 9, 1, start=60)
    ab2 = eng.audio_batch([np.asarray(s, dtype=np.float32) for s in small])
    ref2, got2 = _outputs(eng, ab2), _outputs(eng, ab2)
    assert _call(eng, lib.lsr_recognize_batch, ab2, ref2) == 0
    assert _call(eng, lib.lsr_recognize_batch_noalloc, ab2, got2) == 0
    assert np.array_equal(ref2[1].cpu().numpy(), got2[1].cpu().numpy())
    assert np.array_equal(ref2[0].cpu().numpy().view(np.uint8), got2[0].cpu().numpy().view(np.uint8))


def test_noalloc_rejects_batches_beyond_the_static_workspace():
    from loe_speech_recognition_amd import HiddenMarkovModelInference, synthetic

    models = synthetic.load_word_models()
    inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
    eng = inf._bind()
    sigs, _ = synthetic.batch(2, 4, 1)
    ab = eng.audio_batch([np.asarray(s, dtype=np.float32) for s in sigs])
    out = _outputs(eng, ab)
    P = ctypes.c_void_p
    args = [P(ab.audio.data_ptr()), P(ab.sample_off.data_ptr()), P(ab.frame_off.data_ptr())]
    outs = [P(o.data_ptr()) for o in out]
    # sizes beyond the limits are refused before anything is launched
    assert eng.lib.lsr_recognize_batch_noalloc(eng.ctx, *args, 4097, 60000, 100, *outs, LABEL_CAP) == -3
    assert eng.lib.lsr_recognize_batch_noalloc(eng.ctx, *args, 4, 65537, 100, *outs, LABEL_CAP) == -3
    assert b"static workspace" in eng.lib.lsr_last_error(eng.ctx)
    eng.profile_enable(True)
    try:
        assert eng.lib.lsr_recognize_batch_noalloc(eng.ctx, *args, ab.n, ab.total_frames, ab.max_frames, *outs,
                                                   LABEL_CAP) == -4
    finally:
        eng.profile_enable(False)
