"""Generate tests/golden/sep_g1.npz by running the REAL reference silence stripper
(src/loe_speech_recognition/signal_separation.py, NumPy only) on seeded clips.

Runs only in the build container, where /root/reference is mounted; the module is loaded through
an empty package object whose __path__ points at the reference sources (its __init__ needs
librosa/sounddevice).  Nothing from the reference is copied: the fixture holds the input clips
(int16-valued, stored as int16) and what the reference returned for them.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_separation.py
"""
import importlib
import logging
import os
import sys
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
_pkg = types.ModuleType("loe_speech_recognition")
_pkg.__path__ = ["/root/reference/src/loe_speech_recognition"]
sys.modules["loe_speech_recognition"] = _pkg
SS = importlib.import_module("loe_speech_recognition.signal_separation")
logging.disable(logging.CRITICAL)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import synthetic          # noqa: E402  (NumPy-only generator)

OUT = os.path.join(HERE, "..", "tests", "golden", "sep_g1.npz")
FS = 160


def from_levels(levels, rng, tail=0, jitter=0.0):
    """A clip whose k-th 160-sample frame has mean |x| close to levels[k] (alternating sign)."""
    sign = np.where(np.arange(FS) % 2 == 0, 1.0, -1.0)
    frames = [np.rint(sign * a + jitter * rng.standard_normal(FS)) for a in levels]
    if tail:
        frames.append(np.rint(sign[:tail] * levels[-1]))
    return np.concatenate(frames).astype(np.float32)


def clips(rng):
    out = []
    bank = synthetic.word_bank()
    for i, lab in enumerate(["1", "7", "O", "Z", "4", "9"]):          # one-digit clips with silence around
        out.append(np.rint(synthetic.clip(lab, rng, bank)).astype(np.float32))
    S, L, M = 10000.0, 20.0, 300.0          # speech, silence, between the thresholds (for max 10000)
    out.append(from_levels([L] * 5 + [S] * 30, rng, tail=37))                       # speech to the end: never ends
    out.append(from_levels([L] * 4 + [S] * 3 + [L] * 6, rng, jitter=3.0))           # ends, but fewer than 9 kept frames
    out.append(from_levels([L] * 7 + [S] * 20 + [L], rng))                          # length multiple of 160: empty tail frame trips
    out.append(from_levels([L] * 3 + [S] * 12 + [M] * 6 + [S] * 5 + [L] * 4, rng, tail=11, jitter=2.0))
    out.append(from_levels([L] * 3 + [S] * 12 + [M, L, S, S, S, S, M, M, L, L, L], rng, tail=159))
    out.append(np.zeros(1000, np.float32))                                          # all zero: thresholds 0, never speech
    out.append(from_levels([S] * 15 + [L] * 3, rng, tail=80))                       # speech from the very first frame
    out.append(from_levels([L, S, L, S, L, S, L, S, L, S, L, S, L, L, L], rng, jitter=1.0))   # restarts every other frame
    out.append(from_levels([599.0, 601.0, 600.0] * 4 + [S] * 10 + [101.0, 99.0, 100.0] * 2 + [L] * 3, rng))
    out.append((np.rint(rng.uniform(-50, 50, 3 * FS + 5))).astype(np.float32))      # faint noise only
    # Frames whose float32 energy equals the float32 ROUNDING of a threshold but exceeds the threshold itself:
    # maximum 5020 -> high threshold 0.06 * 5020 = 301.2 and low threshold 0.01 * 5020 = 50.2 in float64, while
    # the frame means 48192 / 160 and 8032 / 160 are float32(301.2) = 301.200012 and float32(50.2) = 50.200001.
    # The reference compares in float64 (its maximum is an np.float64), so both frames count as speech.
    sign = np.where(np.arange(FS) % 2 == 0, 1.0, -1.0)
    on_high = sign * np.concatenate([np.full(32, 302.0), np.full(128, 301.0)])
    on_low = sign * np.concatenate([np.full(32, 51.0), np.full(128, 50.0)])
    quiet, loud = sign * 20.0, sign * 5020.0
    out.append(np.concatenate([quiet] * 3 + [on_high] + [on_low] * 3 + [loud] * 8 + [on_low] * 2 + [quiet] * 4).astype(np.float32))
    return out


def run(sep, signals):
    ok, first, end = [], [], []
    fs = sep.frame_size
    for s in signals:
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                kept = sep.remove_empty(s)
        except sep.FailToProcess:
            ok.append(0); first.append(-1); end.append(-1)
            continue
        n = kept.shape[0]
        assert n % fs == 0
        starts = [f for f in range(0, s.shape[0] - n + 1, fs) if np.array_equal(s[f:f + n], kept)]
        assert len(starts) >= 1
        ok.append(1); first.append(starts[0] // fs); end.append((starts[0] + n) // fs)
    noises = sep.get_all_noises()
    noff = np.concatenate([[0], np.cumsum([x.shape[0] for x in noises])]).astype(np.int64)
    flat = np.concatenate(noises) if noises else np.zeros(0, np.float32)
    assert np.array_equal(flat, np.rint(flat)) and np.abs(flat).max(initial=0) < 32768
    return (np.array(ok, np.int8), np.array(first, np.int32), np.array(end, np.int32), flat.astype(np.int16), noff)


def main():
    rng = np.random.default_rng(304)
    signals = clips(rng)
    for s in signals:
        assert np.abs(s).max() < 32768
    data = {"audio": np.concatenate(signals).astype(np.int16),
            "offsets": np.concatenate([[0], np.cumsum([s.shape[0] for s in signals])]).astype(np.int64)}
    settings = {"script": dict(sample_rate=16000, speech_high_threshold=0.06, speech_low_threshold=0.01),
                "default": dict(),
                "long": dict(speech_high_threshold=0.06, speech_low_threshold=0.01, silence_duration_threshold=0.05)}
    for name, kw in settings.items():
        sep = SS.SignalSeparation(**kw)
        ok, first, end, nflat, noff = run(sep, signals)
        # a second pass over the same object: the carried-over noise of the trailing failures joins the next clip
        ok2, first2, end2, nflat2, noff2 = run(sep, signals[:3])
        data.update({f"{name}_ok": ok, f"{name}_first": first, f"{name}_end": end, f"{name}_noise": nflat,
                     f"{name}_noise_off": noff, f"{name}_ok2": ok2, f"{name}_noise2": nflat2, f"{name}_noise_off2": noff2,
                     f"{name}_settings": np.array([sep.sample_rate, sep.frame_time, sep.speech_high_threshold,
                                                   sep.speech_low_threshold, sep.silence_duration_threshold])})
        print(name, "ok", ok.tolist(), "first", first.tolist(), "end", end.tolist(), "noise clips", len(noff) - 1)
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
