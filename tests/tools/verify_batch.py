"""One-off large-sample check of the decode stage: every clip of a bench-sized batch is decoded on the device
(scores, paths, strings) and by the CPU restatement from the same device features; all three must be identical.

    python tests/tools/verify_batch.py [clips] [digits]
"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "cs-304-speech-recognition-code_amd"))
sys.path.insert(0, ROOT)
import numpy as np

from loe_speech_recognition_amd import synthetic

BETWEEN = -100
_graph = None


def _gen(a):
    return synthetic.batch(*a)


def _decode(chunk):
    global _graph
    from oracle import hmm_oracle as O
    if _graph is None:
        arrays = synthetic.load_word_arrays()
        words = [O.WordModelRef(lab, *arrays[lab]) for lab in synthetic.LOOP_ORDER]
        ltp, gauss, bounds = O.stack_models(words)
        mu, U, ld = O.export_gaussians(gauss)
        band = O.band_from_ltp(ltp, bounds.num_states, [bounds.find_lower(s) for s in range(bounds.num_states)])
        _graph = (O, mu, U, ld, band, bounds)
    O, mu, U, ld, band, bounds = _graph
    out = []
    for x in chunk:
        score, path = O.viterbi_loop_fast(O.emission_fast(x, mu, U, ld), band, bounds.lower, bounds.upper, BETWEEN)
        out.append((score, path, "".join(bounds.get_labels(path))))
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    digits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    with mp.get_context("fork").Pool(16) as pool:
        parts = pool.map(_gen, [(3, min(250, n - s), digits, s) for s in range(0, n, 250)])
    sigs = [s for p in parts for s in p[0]]
    from loe_speech_recognition_amd import MFCC, HiddenMarkovModelInference
    models = synthetic.load_word_models()
    inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
    inf._log_transition_probability_between_words = BETWEEN
    feats = MFCC.batch(sigs, 16000)
    t = time.time()
    vit = inf.viterbi_batch(feats)
    strings = inf.predict_batch(feats)
    fused = inf.recognize_audio(sigs)
    print(f"device: {time.time() - t:.2f} s for {n} clips", flush=True)
    t = time.time()
    with mp.get_context("spawn").Pool(16) as pool:
        ref = [r for part in pool.map(_decode, [feats[s:s + 50] for s in range(0, n, 50)]) for r in part]
    print(f"cpu restatement: {time.time() - t:.1f} s", flush=True)
    bad_score = sum(v[0] != r[0] for v, r in zip(vit, ref))
    bad_path = sum(not np.array_equal(v[1], r[1]) for v, r in zip(vit, ref))
    bad_str = sum(a != r[2] for a, r in zip(strings, ref))
    bad_fused = sum(a != b for a, b in zip(fused, strings))
    print(f"clips {n}: score mismatches {bad_score}, path mismatches {bad_path}, string mismatches {bad_str}, "
          f"fused-vs-staged mismatches {bad_fused}")
    sys.exit(1 if bad_score or bad_path or bad_str or bad_fused else 0)


if __name__ == "__main__":
    main()
