"""Connected-digit test WITHOUT the silence model, the flow of the reference's
scripts/project5_test_ndigits_no_sil.py: the eleven digit models in a free loop (55 states), between-word
log-probability -250, all N-digit clips of the training or the test split in one device call."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import TI_DIGITS_LABELS, HiddenMarkovModelInference, TIDigits       # noqa: E402


def main(corpus: str, model_folder: str, n: int = 7, between=-250, split: str = "train"):
    decoder = HiddenMarkovModelInference.from_folder(model_folder, list(TI_DIGITS_LABELS))
    decoder._log_transition_probability_between_words = between
    data = TIDigits(corpus, isLazyLoading=True, keep_int16=True)
    clips = (data.train_dataset if split == "train" else data.test_dataset).get_all_n_digits(n)
    truth = [label for label, signals in clips.items() for _ in signals]
    prediction = decoder.recognize_audio([s for signals in clips.values() for s in signals]) if truth else []
    accuracy = sum(t == p for t, p in zip(truth, prediction)) / max(1, len(truth))
    print(f"In total, {len(truth)} data points tested, the accuracy rate is {accuracy * 100: 2f}%")
    return truth, prediction, accuracy


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("corpus")
    ap.add_argument("models")
    ap.add_argument("-n", type=int, default=7)
    ap.add_argument("--split", choices=("train", "test"), default="train")
    args = ap.parse_args()
    main(args.corpus, args.models, args.n, split=args.split)
