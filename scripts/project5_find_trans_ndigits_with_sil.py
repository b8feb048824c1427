"""Sweep of the between-word log-probability, the flow of the reference's
scripts/project5_find_trans_ndigits_with_sil.py (0, -50, ..., -4950 on five clips per two-digit
string): the reference re-decodes everything per value; here the emission scores are computed once
and the decoder runs once per value (HiddenMarkovModelInference.predict_sweep)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import MFCC, TI_DIGITS_LABELS, HiddenMarkovModelInference, TIDigits   # noqa: E402


def main(corpus: str, model_folder: str, n: int = 2, per_label: int = 5, penalties=None):
    penalties = [-i for i in range(0, 5000, 50)] if penalties is None else list(penalties)
    decoder = HiddenMarkovModelInference.from_folder(model_folder, list(TI_DIGITS_LABELS) + ["S"])
    clips = TIDigits(corpus, isLazyLoading=True).train_dataset.get_all_n_digits(n)
    truth = [label for label, signals in clips.items() for _ in signals[:per_label]]
    mfccs = MFCC.batch_device([s for signals in clips.values() for s in signals[:per_label]], sample_rate=16000)
    results = []
    for value, row in zip(penalties, decoder.predict_sweep(mfccs, penalties)):
        accuracy = sum(t == p for t, p in zip(truth, row)) / len(truth) * 100
        print(f"For Log Transition Probability between word {value}: {accuracy: 2f}%")
        results.append(accuracy)
    return penalties, results


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("corpus")
    ap.add_argument("models")
    args = ap.parse_args()
    main(args.corpus, args.models)
