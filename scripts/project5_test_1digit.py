"""One-digit clips through the free-loop decoder, the flow of the reference's scripts/project5_test_1digit.py:
the eleven digit models without silence, the class-default between-word log-probability np.log(0.005) (an
np.float64: the cross-word candidate is added in float64), every one-digit training clip in one device call."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import TI_DIGITS_LABELS, HiddenMarkovModelInference, TIDigits       # noqa: E402


def main(corpus: str, model_folder: str):
    decoder = HiddenMarkovModelInference.from_folder(model_folder, list(TI_DIGITS_LABELS))
    clips = TIDigits(corpus, isLazyLoading=True, keep_int16=True).train_dataset.get_all_n_digits(1)
    truth = [label for label, signals in clips.items() for _ in signals]
    prediction = decoder.recognize_audio([s for signals in clips.values() for s in signals]) if truth else []
    accuracy = sum(t == p for t, p in zip(truth, prediction)) / max(1, len(truth))
    print(f"In total, {len(truth)} data points tested, the accuracy rate is {accuracy * 100: 2f}%")
    return truth, prediction, accuracy


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("corpus")
    ap.add_argument("models")
    args = ap.parse_args()
    main(args.corpus, args.models)
