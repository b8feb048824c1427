"""The chunked host -> device pipeline (lsr_recognize_host) must return exactly what one resident
batch returns (lsr_recognize_batch): scores, label records and strings, for float32 and int16 clips,
whatever the chunk size and thread count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def decoder():
    from loe_speech_recognition_amd import HiddenMarkovModelInference, synthetic

    models = synthetic.load_word_models()
    inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
    inf._log_transition_probability_between_words = -100
    return inf


def _resident(decoder, sigs, cap):
    eng = decoder._bind()
    ab = eng.audio_batch(sigs)
    scores, _, labels, n_labels = eng.recognize(ab, want_paths=False, label_cap=cap)
    return scores.cpu().numpy(), labels.cpu().numpy(), n_labels.cpu().numpy()


def _same(a, b, cap):
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert np.array_equal(a[2], b[2])
    for i in range(a[1].shape[0]):
        n = min(int(a[2][i]), cap)
        assert np.array_equal(a[1][i, :n], b[1][i, :n])


@pytest.mark.parametrize("as_int16", [False, True])
def test_streamed_equals_resident(decoder, as_int16):
    from loe_speech_recognition_amd import synthetic

    sigs, truth = synthetic.batch(3, 300, [1, 2, 4, 7], start=1200)
    if as_int16:
        sigs = [np.round(s).astype(np.int16) for s in sigs]
    cap = 16
    ref = _resident(decoder, sigs, cap)
    eng = decoder._bind()
    total = sum(s.shape[0] for s in sigs)
    # one chunk, many chunks, chunks of a single clip, odd thread counts
    for chunk, threads in ((0, 0), (total // 7, 3), (1, 2), (total // 2 + 1, 16)):
        got = eng.recognize_host(sigs, label_cap=cap, chunk_samples=chunk, threads=threads)
        _same(ref, got, cap)
    strings = decoder.recognize_audio(sigs, label_cap=cap, streamed=True, chunk_samples=total // 5)
    assert strings == decoder.recognize_audio(sigs, label_cap=cap)
    assert np.mean([a == b for a, b in zip(strings, truth)]) >= 0.9


def test_streamed_repeated_calls_and_growth(decoder):
    """Staging buffers are reused and grow: small batch, larger batch, small again."""
    from loe_speech_recognition_amd import synthetic

    eng = decoder._bind()
    for count, digits, start in ((5, 1, 0), (200, 4, 50), (3, 7, 400)):
        sigs, _ = synthetic.batch(5, count, digits, start=start)
        _same(_resident(decoder, sigs, 8), eng.recognize_host(sigs, label_cap=8, chunk_samples=400000), 8)


def test_streamed_mixed_dtype_and_views(decoder):
    """Non-contiguous views and float64 inputs are converted like the resident path converts them."""
    from loe_speech_recognition_amd import synthetic

    sigs, _ = synthetic.batch(2, 12, 2, start=77)
    odd = [np.repeat(s, 2)[::2] if i % 2 else s.astype(np.float64) for i, s in enumerate(sigs)]
    assert decoder.recognize_audio(odd, streamed=True) == decoder.recognize_audio(sigs)


def test_streamed_rejects_short_clip_and_overflow_falls_back(decoder):
    from loe_speech_recognition_amd import synthetic

    eng = decoder._bind()
    with pytest.raises(ValueError):
        eng.recognize_host([np.zeros(100, np.float32)])
    sigs, _ = synthetic.batch(5, 4, 7, start=900)
    assert decoder.recognize_audio(sigs, label_cap=2, streamed=True) == decoder.recognize_audio(sigs, label_cap=32)
