"""Device silence stripping (lsr_separation_scan / lsr_gather_segments behind SignalSeparation) against
the reference's outputs (tests/golden/sep_g1.npz) and the CPU restatement: decisions, slice bounds and
noise clips are integer/byte work and must be identical; frame energies must be bit-identical float32."""
import numpy as np
import pytest

from loe_speech_recognition_amd import MFCC, SignalSeparation, _native, synthetic
from oracle import separation_oracle as so
from helpers import load

pytestmark = pytest.mark.gpu
SETTINGS = ["script", "default", "long"]


def fixture_clips(g, dtype=np.float32):
    audio, off = g["audio"].astype(dtype), g["offsets"]
    return [audio[off[i]:off[i + 1]] for i in range(off.shape[0] - 1)]


def make(g, name):
    sr, ft, hi, lo, sil = g[f"{name}_settings"]
    return SignalSeparation(int(sr), float(ft), float(hi), float(lo), float(sil))


@pytest.mark.parametrize("dtype", [np.float32, np.int16])
@pytest.mark.parametrize("name", SETTINGS)
def test_reference_fixture(name, dtype):
    g = load("sep_g1.npz")
    clips = fixture_clips(g, dtype)
    sep = make(g, name)
    fs = sep.frame_size
    kept = sep.remove_empty_batch(clips)
    ok = g[f"{name}_ok"].astype(bool)
    assert len(kept) == int(ok.sum())
    for arr, i in zip(kept, np.nonzero(ok)[0]):
        f, e = int(g[f"{name}_first"][i]), int(g[f"{name}_end"][i])
        assert arr.dtype == np.float32 and np.array_equal(arr, clips[i][f * fs:e * fs].astype(np.float32)), i
    noff, flat = g[f"{name}_noise_off"], g[f"{name}_noise"].astype(np.float32)
    noises = sep.get_all_noises()
    assert len(noises) == noff.shape[0] - 1
    for k, nz in enumerate(noises):
        assert nz.dtype == np.float32 and np.array_equal(nz, flat[noff[k]:noff[k + 1]]), k
    # second call on the same object: pending noise of the trailing failed clips joins the next clip
    assert len(sep.remove_empty_batch(clips[:3])) == int(g[f"{name}_ok2"].sum())
    noff2, flat2 = g[f"{name}_noise_off2"], g[f"{name}_noise2"].astype(np.float32)
    assert len(sep.get_all_noises()) == noff2.shape[0] - 1
    for k, nz in enumerate(sep.get_all_noises()):
        assert np.array_equal(nz, flat2[noff2[k]:noff2[k + 1]]), k


def test_single_clip_interface():
    g = load("sep_g1.npz")
    clips = fixture_clips(g)
    sep = make(g, "script")
    fs = sep.frame_size
    out = sep.remove_empty(clips[0])
    assert np.array_equal(out, clips[0][int(g["script_first"][0]) * fs:int(g["script_end"][0]) * fs])
    with pytest.raises(SignalSeparation.FailToProcess):
        sep.remove_empty(clips[6])                      # speech to the end
    with pytest.raises(SignalSeparation.FailToProcess):
        sep.remove_empty(clips[7])                      # fewer than 9 frames kept
    assert sep.remove_empty_batch([]) == []
    with pytest.raises(TypeError):
        sep.remove_empty_batch([[1.0, 2.0]])


@pytest.mark.parametrize("frame_size", [160, 80, 7, 8, 130, 200, 256])
def test_frame_energies_bit_exact(frame_size):
    rng = np.random.default_rng(frame_size)
    clips = [(3000 * rng.standard_normal(int(n))).astype(np.float32)
             for n in rng.integers(1, 12 * frame_size, size=40)]
    clips.append(np.zeros(4 * frame_size, np.float32))                     # empty trailing frame -> NaN
    clips.append((100 * rng.standard_normal(frame_size - 1 or 1)).astype(np.float32))
    eng = _native.get_engine()
    scan = eng.separation_scan(clips, frame_size, 0.06, 0.01, 2, want_energy=True)
    want = np.concatenate([so.frame_energies(c, frame_size) for c in clips])
    got = scan["energy"]
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32)[~np.isnan(want)], want.view(np.uint32)[~np.isnan(want)])
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(scan["max_volume"], np.array([np.abs(c).max() for c in clips], np.float32))


def test_frame_size_limit():
    eng = _native.get_engine()
    with pytest.raises(_native.NativeError):
        eng.separation_scan([np.ones(1000, np.float32)], 257, 0.06, 0.01, 2)


def test_many_synthetic_clips_against_restatement():
    bank = synthetic.word_bank()
    clips = []
    for i in range(600):
        rng = np.random.default_rng([77, i])
        digits = "".join(rng.choice(synthetic.WORDS, size=int(rng.integers(1, 4))))
        c = synthetic.clip(digits, rng, bank)
        if i % 5 == 0:
            c = c[:int(rng.integers(c.shape[0] // 3, c.shape[0]))]        # cut somewhere: may never end
        clips.append(c)
    sep = SignalSeparation(speech_high_threshold=0.06, speech_low_threshold=0.01)
    ref = so.SignalSeparationRef(16000, 0.01, 0.06, 0.01, 0.02)
    kept = sep.remove_empty_batch(clips)
    want = ref.remove_empty_batch(clips)
    assert len(kept) == len(want) and 0 < len(kept) < len(clips) + 1
    for a, b in zip(kept, want):
        assert np.array_equal(a, b)
    assert len(sep.get_all_noises()) == len(ref.noises)
    for a, b in zip(sep.get_all_noises(), ref.noises):
        assert np.array_equal(a, b)


def test_device_gather_feeds_mfcc():
    bank = synthetic.word_bank()
    clips = [synthetic.clip(lab, np.random.default_rng([5, i]), bank) for i, lab in enumerate("1Z49O72")]
    clips.insert(3, np.zeros(2000, np.float32))                           # dropped
    sep = SignalSeparation(speech_high_threshold=0.06, speech_low_threshold=0.01)
    host = SignalSeparation(speech_high_threshold=0.06, speech_low_threshold=0.01).remove_empty_batch(clips)
    batch, keep = sep.remove_empty_batch_device(clips)
    assert keep == [0, 1, 2, 4, 5, 6, 7] and batch.n == len(host)
    assert np.array_equal(batch.audio.cpu().numpy(), np.concatenate(host))
    eng = _native.get_engine()
    fb = eng.mfcc(batch)
    got = _native.split_ragged(fb.feats.cpu().numpy(), fb.host_off)
    for a, b in zip(got, MFCC.batch(host, 16000)):
        assert np.array_equal(a, b)
