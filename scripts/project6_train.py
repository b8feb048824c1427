"""Continuous-speech re-training of all word models, the flow of the reference's scripts/project6_train.py
(lines 14-41): load the isolated-word models (eleven digits and the silence model), collect the MFCCs of every
2..7-digit training clip keyed by its digit string, run HiddenMarkovModelTrainContinuous.train and save the
models whatever happens (try / finally, so an interrupted run keeps its progress).  Each iteration is one
forced-alignment batch, one statistics reduction and the host M-step instead of the reference's process pool."""
import argparse
import os
import sys
from typing import Dict, List

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import MFCC, TI_DIGITS_LABELS, HiddenMarkovModelTrainContinuous, TIDigits      # noqa: E402


def main(corpus: str, cache: str, model_name: str = "big_model_speech_only", max_iterations: int = 200) -> str:
    ti_digits = TIDigits(corpus, isLazyLoading=True)
    models_to_load: List[str] = list(TI_DIGITS_LABELS) + ["S"]
    hmm_train = HiddenMarkovModelTrainContinuous.from_folder(os.path.join(cache, model_name), models_to_load)
    hmm_train.isTqdm = False
    train_dataset = ti_digits.train_dataset
    labeled_mfccs: Dict[str, List[np.ndarray]] = {}
    for n in range(2, 8):
        for label, signals in train_dataset.get_all_n_digits(n).items():
            labeled_mfccs[label] = MFCC.batch(signals, sample_rate=16000)
    out = os.path.join(cache, f"{model_name}_continuous_2")
    try:
        hmm_train.train(labeled_mfccs=labeled_mfccs, max_iterations=max_iterations)
    except KeyboardInterrupt:
        print("Keyboard interrupt received")
    finally:
        hmm_train.save(out)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("corpus")
    ap.add_argument("--cache", default=".cache")
    ap.add_argument("--iterations", type=int, default=200)
    args = ap.parse_args()
    print("models written to", main(args.corpus, args.cache, max_iterations=args.iterations))
