"""The silence-stripper restatement (oracle/separation_oracle.py) against the outputs the real
reference produced for the same clips (tests/golden/sep_g1.npz, tools/make_golden_separation.py)."""
import numpy as np
import pytest

from oracle import separation_oracle as so
from helpers import load

SETTINGS = ["script", "default", "long"]


def fixture_clips(g):
    audio, off = g["audio"].astype(np.float32), g["offsets"]
    return [audio[off[i]:off[i + 1]] for i in range(off.shape[0] - 1)]


def make_ref(g, name):
    sr, ft, hi, lo, sil = g[f"{name}_settings"]
    return so.SignalSeparationRef(int(sr), float(ft), float(hi), float(lo), float(sil))


@pytest.mark.parametrize("name", SETTINGS)
def test_matches_reference(name):
    g = load("sep_g1.npz")
    clips = fixture_clips(g)
    ref = make_ref(g, name)
    fs = ref.frame_size
    for i, s in enumerate(clips):
        status, kept = ref.remove_empty(s)
        assert (status == so.OK) == bool(g[f"{name}_ok"][i]), i
        if status == so.OK:
            f, e = int(g[f"{name}_first"][i]), int(g[f"{name}_end"][i])
            assert kept.dtype == np.float32 and np.array_equal(kept, s[f * fs:e * fs]), i
    noff = g[f"{name}_noise_off"]
    assert len(ref.noises) == noff.shape[0] - 1
    flat = g[f"{name}_noise"].astype(np.float32)
    for k, nz in enumerate(ref.noises):
        assert np.array_equal(nz, flat[noff[k]:noff[k + 1]]), k
    # second call on the same object: noise left over from the trailing failed clips is carried over
    kept2 = ref.remove_empty_batch(clips[:3])
    assert len(kept2) == int(g[f"{name}_ok2"].sum())
    noff2, flat2 = g[f"{name}_noise_off2"], g[f"{name}_noise2"].astype(np.float32)
    assert len(ref.noises) == noff2.shape[0] - 1
    for k, nz in enumerate(ref.noises):
        assert np.array_equal(nz, flat2[noff2[k]:noff2[k + 1]]), k


def test_energy_is_float32_pairwise_mean():
    rng = np.random.default_rng(0)
    s = (3000 * rng.standard_normal(1000)).astype(np.float32)
    e = so.frame_energies(s, 160)
    assert e.dtype == np.float32 and e.shape == (7,)
    assert e[2] == np.abs(s[320:480]).mean(dtype=np.float32)
    assert e[6] == np.abs(s[960:]).mean(dtype=np.float32)
    assert np.isnan(so.frame_energies(s[:960], 160)[6])        # empty trailing frame


def test_scan_status_codes():
    hi, lo = 0.06, 0.01
    e = np.array([1, 1, 900, 900, 1, 1], np.float32)
    assert so.scan(e, 10000.0, hi, lo, 2)[:3] == (2, 5, so.TOO_SHORT)
    e = np.array([1] + [900] * 12, np.float32)
    assert so.scan(e, 10000.0, hi, lo, 2)[:3] == (1, -1, so.NO_END)
    e = np.array([1] + [900] * 12 + [1, np.nan], np.float32)
    first, end, status, noise = so.scan(e, 10000.0, hi, lo, 2)
    assert (first, end, status) == (1, 14, so.OK) and noise.tolist() == [1] + [0] * 13 + [1]


def test_host_bookkeeping_without_gpu():
    """The host half of the product class (slicing, noise filing with carry-over, failures) fed with scan
    results computed by the restatement instead of the device: must reproduce the reference's outputs."""
    from loe_speech_recognition_amd import SignalSeparation

    g = load("sep_g1.npz")
    clips = fixture_clips(g)
    sep = SignalSeparation(speech_high_threshold=0.06, speech_low_threshold=0.01)
    fs = sep.frame_size

    def fake_scan(signals):
        first, end, status, noise, vol, off = [], [], [], [], [], [0]
        for s in signals:
            e = so.frame_energies(s, fs)
            f, t, st, nz = so.scan(e, float(np.max(np.abs(s))), 0.06, 0.01, sep.maximum_silence_frames)
            first.append(f); end.append(t); status.append(st); noise.append(nz); vol.append(np.max(np.abs(s)))
            off.append(off[-1] + e.shape[0])
        return {"first": np.array(first), "end": np.array(end), "status": np.array(status),
                "noise": np.concatenate(noise), "max_volume": np.array(vol, np.float32), "scan_off": np.array(off)}

    sep._scan = fake_scan
    kept = sep.remove_empty_batch(clips)
    ok = g["script_ok"].astype(bool)
    assert len(kept) == int(ok.sum())
    for arr, i in zip(kept, np.nonzero(ok)[0]):
        assert np.array_equal(arr, clips[i][int(g["script_first"][i]) * fs:int(g["script_end"][i]) * fs])
    noff, flat = g["script_noise_off"], g["script_noise"].astype(np.float32)
    assert len(sep.get_all_noises()) == noff.shape[0] - 1
    for k, nz in enumerate(sep.get_all_noises()):
        assert np.array_equal(nz, flat[noff[k]:noff[k + 1]])
    with pytest.raises(SignalSeparation.FailToProcess):
        sep.remove_empty(clips[6])
    assert sep.detect_speech(np.full(160, 5000.0, np.float32), "high") in (True, False)
