"""Seeded synthetic 16 kHz digit-string audio standing in for TI-Digits (which cannot be
shipped).  A word is five consecutive segments of 40-100 ms, each a sum of three sinusoids at
word/segment-specific frequencies plus white noise; silence is low-level white noise of
80-250 ms before, between and after the words.  Amplitudes are in int16 range like the
reference's wav loader delivers (src/loe_speech_recognition/ti_digits.py:130-134)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

SAMPLE_RATE = 16000
WORDS: List[str] = ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "Z"]
BANK_SEED = 1234
UTT_SEED = 2025
WORD_NOISE = 100.0
SILENCE_NOISE = 30.0


def word_bank(seed: int = BANK_SEED) -> Tuple[np.ndarray, np.ndarray]:
    """(frequencies Hz, amplitudes) of shape [11 words, 5 segments, 3 partials]."""
    rng = np.random.default_rng(seed)
    return rng.uniform(200.0, 6000.0, size=(len(WORDS), 5, 3)), rng.uniform(1000.0, 6000.0, size=(len(WORDS), 5, 3))


def _silence(rng, lo_ms: float = 80.0, hi_ms: float = 250.0) -> np.ndarray:
    n = int(rng.uniform(lo_ms, hi_ms) * SAMPLE_RATE / 1000.0)
    return (SILENCE_NOISE * rng.standard_normal(n)).astype(np.float32)


def _word(rng, w: int, bank) -> np.ndarray:
    freqs, amps = bank
    parts = []
    for k in range(5):
        n = int(rng.uniform(40.0, 100.0) * SAMPLE_RATE / 1000.0)
        t = np.arange(n, dtype=np.float32) * np.float32(2.0 * np.pi / SAMPLE_RATE)
        seg = WORD_NOISE * rng.standard_normal(n).astype(np.float32)
        for j in range(3):
            seg += np.float32(amps[w, k, j]) * np.sin(np.float32(freqs[w, k, j]) * t)
        parts.append(seg)
    return np.concatenate(parts).astype(np.float32)


def word_clip(label: str, rng, bank=None) -> np.ndarray:
    """One isolated word without surrounding silence."""
    return _word(rng, WORDS.index(label), bank or word_bank())


def silence_clip(rng) -> np.ndarray:
    return _silence(rng, 90.0, 400.0)


def clip(digits: str, rng, bank=None) -> np.ndarray:
    """silence, word, silence, word, ..., silence."""
    bank = bank or word_bank()
    parts = [_silence(rng)]
    for c in digits:
        parts.append(_word(rng, WORDS.index(c), bank))
        parts.append(_silence(rng))
    return np.concatenate(parts)


def clip_with_spans(digits: str, rng, bank=None) -> Tuple[np.ndarray, List[Tuple[str, int, int]]]:
    """Like ``clip`` (same random stream) plus [(label, first sample, end sample)] per piece,
    "S" for the silences: the ground-truth segmentation used to bootstrap models."""
    bank = bank or word_bank()
    parts = [_silence(rng)]
    names = ["S"]
    for c in digits:
        parts.append(_word(rng, WORDS.index(c), bank))
        names.append(c)
        parts.append(_silence(rng))
        names.append("S")
    ends = np.cumsum([p.shape[0] for p in parts])
    spans = [(n, int(e - p.shape[0]), int(e)) for n, p, e in zip(names, parts, ends)]
    return np.concatenate(parts), spans


def utterance(config_id: int, index: int, n_digits: int, bank=None) -> Tuple[np.ndarray, str]:
    """Utterance `index` of a configuration: its own SeedSequence([2025, config_id, index])."""
    rng = np.random.default_rng(np.random.SeedSequence([UTT_SEED, config_id, index]))
    digits = "".join(WORDS[i] for i in rng.integers(0, len(WORDS), size=n_digits))
    return clip(digits, rng, bank), digits


def batch(config_id: int, count: int, n_digits, start: int = 0) -> Tuple[List[np.ndarray], List[str]]:
    """`count` utterances; `n_digits` is an int or a sequence cycled over the utterances."""
    bank = word_bank()
    sigs, truth = [], []
    for i in range(start, start + count):
        n = n_digits if isinstance(n_digits, int) else n_digits[i % len(n_digits)]
        s, d = utterance(config_id, i, n, bank)
        sigs.append(s)
        truth.append(d)
    return sigs, truth


def group_by_label(signals: Sequence[np.ndarray], truth: Sequence[str]) -> Dict[str, List[np.ndarray]]:
    """{digit string: [clips]} like TIDigits' DataLoader.get_all_n_digits returns."""
    out: Dict[str, List[np.ndarray]] = {}
    for s, d in zip(signals, truth):
        out.setdefault(d, []).append(s)
    return out


MODEL_FILE = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "data", "synth_models.npz")
LOOP_ORDER: List[str] = ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "S", "Z"]   # sorted() order of the label folders


def load_word_arrays(path: str = MODEL_FILE) -> Dict[str, Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """{label: (means f32[N,39], covariances f32[N,39,39], transition matrix f32[N,N])} of the
    models trained on this module's synthetic words (tests/tools/make_synth_models.py)."""
    with np.load(path, allow_pickle=False) as z:
        return {lab: (z[f"{lab}_means"], z[f"{lab}_cov"], z[f"{lab}_trans"]) for lab in WORDS + ["S"]}


def load_word_models(path: str = MODEL_FILE):
    """{label: HiddenMarkovModelTrainable} ready for decoding or further training."""
    from .hidden_markov_model import HiddenMarkovModelTrainable
    from .transition_probability import TransitionProbabilities

    out = {}
    for lab, (means, cov, trans) in load_word_arrays(path).items():
        m = HiddenMarkovModelTrainable(lab, isTqdm=False)
        m._means, m._covariances = means.copy(), cov.copy()
        m._transition_probs = TransitionProbabilities.from_transition_probability(trans)
        m._update_inference_weights()
        out[lab] = m
    return out
