"""Generate tests/golden/mfcc_g8_glue.npz by running the REAL reference mfcc.py.

Runs only in the build container, where /root/reference is mounted.  mfcc.py imports librosa, which is
not installed, so a stand-in module named ``librosa`` is registered first (the same trick
tools/make_golden_hmm.py uses for ``uniplot``).  Two things are recorded:

  (i)  MFCC.normalize_mfccs (mfcc.py:50-69) needs nothing from librosa: the real static method is called on
       seeded (13, T) arrays.  This pins the per-frame (axis=0) normalisation.
  (ii) MFCC.__post_init__ and MFCC.batch (mfcc.py:24-44, 71-84) are run for real with the stand-in's
       melspectrogram / power_to_db / mfcc / delta bound to the stage functions of oracle/mfcc_oracle.py.
       This pins ONLY what mfcc.py itself defines: the arguments it passes, that both deltas are taken from the
       un-normalised statics (mfcc.py:39-40), the concatenation order (mfcc.py:43) and the transposition in
       batch (mfcc.py:84).  It does NOT pin librosa's arithmetic, which stays "parity unpinned": the numbers
       inside the stages are the oracle's own.

Nothing from the reference is copied: the fixture holds seeded inputs, the outputs the reference computed for
them, and the log of the calls it made.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_mfcc_glue.py
"""
import importlib
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import mfcc_oracle as M  # noqa: E402

CALLS = []
_last = {}


def _melspectrogram(y=None, sr=None, **kw):
    CALLS.append(["melspectrogram", {"sr": sr, **{k: (float(v) if isinstance(v, float) else v) for k, v in sorted(kw.items())}}])
    assert kw == dict(n_mels=40, n_fft=320, hop_length=160, fmin=133.33, fmax=6855.4976), kw
    return M.mel_power(y, sr)


def _power_to_db(S, ref=1.0):
    CALLS.append(["power_to_db", {"ref": getattr(ref, "__name__", repr(ref))}])
    assert ref is np.max
    return M.power_to_db(S)


def _mfcc(S=None, sr=None, n_mfcc=20):
    CALLS.append(["mfcc", {"sr": sr, "n_mfcc": n_mfcc}])
    out = M.dct_mfcc(S, n_mfcc)
    _last["mfcc"] = out
    return out


def _delta(data, order=1, **kw):
    CALLS.append(["delta", {"order": order, "input_is_the_unnormalised_mfcc_output": data is _last.get("mfcc"), **kw}])
    return M.delta(data, order)


librosa = types.ModuleType("librosa")
librosa.feature = types.ModuleType("librosa.feature")
librosa.feature.melspectrogram = _melspectrogram
librosa.feature.mfcc = _mfcc
librosa.feature.delta = _delta
librosa.power_to_db = _power_to_db
sys.modules["librosa"] = librosa
sys.modules["librosa.feature"] = librosa.feature
_pkg = types.ModuleType("loe_speech_recognition")
_pkg.__path__ = ["/root/reference/src/loe_speech_recognition"]
sys.modules["loe_speech_recognition"] = _pkg
R = importlib.import_module("loe_speech_recognition.mfcc")


def clip(rng, n):
    """int16-range float32 audio: a few tones that change over time plus noise."""
    t = np.arange(n) / 16000.0
    y = np.zeros(n)
    for _ in range(4):
        f = rng.uniform(200, 6000)
        y += rng.uniform(500, 5000) * np.sin(2 * np.pi * f * t + rng.uniform(0, 6)) * (0.5 + 0.5 * np.sin(2 * np.pi * rng.uniform(1, 9) * t))
    return (y + 80 * rng.standard_normal(n)).astype(np.float32)


def main():
    rng = np.random.default_rng(808)
    out = {}
    # (i) the real normalize_mfccs
    for i, T in enumerate((9, 17, 64)):
        x = (rng.standard_normal((13, T)) * rng.uniform(0.5, 30, size=(13, 1)) + rng.uniform(-200, 50, size=(13, 1))).astype(np.float32)
        out[f"norm{i}_in"] = x
        out[f"norm{i}_out"] = R.MFCC.normalize_mfccs(x)
    x64 = rng.standard_normal((13, 12)) * 7.0                     # float64 input as well
    out["norm3_in"], out["norm3_out"] = x64, R.MFCC.normalize_mfccs(x64)
    const = np.tile(np.float32(3.5), (13, 10))                     # zero variance in every frame: the 1e-8 guard
    out["norm4_in"], out["norm4_out"] = const, R.MFCC.normalize_mfccs(const)
    # (ii) the real __post_init__ / batch over the stand-in stages
    signals = [clip(rng, n) for n in (1440, 1600, 2555, 4000, 8001)]
    CALLS.clear()
    for i, s in enumerate(signals):
        out[f"sig{i}"] = s
        out[f"feat{i}"] = R.MFCC(s, 16000).feature_vector
    n_single = len(CALLS)
    batch = R.MFCC.batch(signals, 16000)
    for i, b in enumerate(batch):
        out[f"batch{i}"] = b
    out["n_signals"] = np.array(len(signals))
    out["calls_of_first_clip"] = np.array(json.dumps(CALLS[:n_single // len(signals)]))
    out["pins"] = np.array("mfcc.py:39-43,84 and 50-69 only; the stage arithmetic is oracle/mfcc_oracle.py, NOT librosa (parity unpinned)")
    path = os.path.join(ROOT, "tests", "golden", "mfcc_g8_glue.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", json.dumps(CALLS[:n_single // len(signals)]))


if __name__ == "__main__":
    main()
