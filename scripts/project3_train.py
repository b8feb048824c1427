"""Isolated-word training from the raw one-digit clips, the flow of the reference's scripts/project3_train.py:
per label MFCC of every training clip, segmental k-means (five states, at most 100 iterations), save.  The
reference aligns the clips of a label in a process pool; here each iteration is one device call per label."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import MFCC, TI_DIGITS_LABELS, HiddenMarkovModelTrainable, TIDigits   # noqa: E402


def main(corpus: str, cache: str, model_name: str = "big_model_no_silence", max_iterations: int = 100) -> str:
    out = os.path.join(cache, model_name)
    train = TIDigits(corpus, isLazyLoading=True).train_dataset
    for label in TI_DIGITS_LABELS:
        mfccs = MFCC.batch_device(train[label], sample_rate=16000)        # features stay on the device
        HiddenMarkovModelTrainable.from_data(label, mfccs, num_of_states=5, max_iterations=max_iterations,
                                             isMultiProcessingTraining=True, isTqdm=False).save(out)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("corpus")
    ap.add_argument("--cache", default=".cache")
    args = ap.parse_args()
    print("models written to", main(args.corpus, args.cache))
