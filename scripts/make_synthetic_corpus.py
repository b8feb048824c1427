"""Write a miniature corpus in the folder layout TIDigits reads
(<root>/{Adults,Children}/TIDIGITS/{TRAIN,TEST}/<speaker>/<digits><take>.WAV, int16 PCM, 16 kHz),
filled with the seeded synthetic words of loe_speech_recognition_amd.synthetic — TI-Digits itself
cannot be shipped."""
import argparse
import os
import sys

import numpy as np
import scipy.io.wavfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import synthetic        # noqa: E402


def write_corpus(root: str, isolated_per_word: int = 12, strings_per_length: int = 12, lengths=(2, 4), seed: int = 7):
    bank = synthetic.word_bank()
    rng = np.random.default_rng(seed)
    takes = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"
    count = {}
    for split in ("TRAIN", "TEST"):
        jobs = [w for w in synthetic.WORDS for _ in range(isolated_per_word)]
        for n in lengths:
            jobs += ["".join(rng.choice(synthetic.WORDS, size=n)) for _ in range(strings_per_length)]
        for i, digits in enumerate(jobs):
            group = "Adults" if i % 3 else "Children"
            folder = os.path.join(root, group, "TIDIGITS", split, f"spk{i % 4}")
            os.makedirs(folder, exist_ok=True)
            k = count.get((folder, digits), 0)
            count[(folder, digits)] = k + 1
            pcm = np.clip(np.rint(synthetic.clip(digits, rng, bank)), -32768, 32767).astype(np.int16)
            scipy.io.wavfile.write(os.path.join(folder, f"{digits}{takes[k]}.WAV"), synthetic.SAMPLE_RATE, pcm)
    return root


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--isolated-per-word", type=int, default=12)
    args = ap.parse_args()
    write_corpus(args.root, args.isolated_per_word)
