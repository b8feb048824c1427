"""lsr_recognize_host on the config-3 batch from host arrays, float32 and int16, by chunk size (samples per chunk)."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import synthetic  # noqa: E402


def gen(a):
    return synthetic.batch(*a)


def main():
    with mp.get_context("fork").Pool(16) as pool:
        parts = pool.map(gen, [(3, 250, 4, 250 * k) for k in range(40)])
    sigs = [s for p in parts for s in p[0]]
    from loe_speech_recognition_amd import HiddenMarkovModelInference
    models = synthetic.load_word_models()
    inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
    inf._log_transition_probability_between_words = -100
    eng = inf._bind()
    for tag, clips in (("float32", sigs), ("int16", [np.round(c).astype(np.int16) for c in sigs])):
        for chunk in (12 << 20, 24 << 20, 48 << 20, 96 << 20):
            for _ in range(2):
                eng.recognize_host(clips, label_cap=16, chunk_samples=chunk, threads=16)
            t = time.perf_counter()
            for _ in range(5):
                eng.recognize_host(clips, label_cap=16, chunk_samples=chunk, threads=16)
            dt = (time.perf_counter() - t) / 5
            print(f"{tag} chunk {chunk >> 20} Mi samples: {dt * 1e3:.2f} ms, {eng.recognize_host_timing()}")


if __name__ == "__main__":
    main()
