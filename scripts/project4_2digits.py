"""The flow of the reference's scripts/project4_2digits.py (lines 17-34): ten random two-digit strings, each
built by concatenating clip number 2 of its two one-digit labels (DataLoader.get_combined), turned into MFCC
features and handed to the isolated-word classifier ModelCollection.predict - which answers with ONE label for
the whole concatenation (the reference logs "".join(pred_labels) of that single label).  The reference calls
mc.predict clip by clip; here the ten clips are one device batch."""
import argparse
import itertools
import os
import random
import sys
from typing import List, Tuple

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import MFCC, TI_DIGITS_LABELS, ModelCollection, TIDigits      # noqa: E402


def combine_and_predict(corpus: str, model_folder: str, n_digits: int, n_samples: int, seed: int = 0
                        ) -> Tuple[List[str], List[str]]:
    ti_digits = TIDigits(corpus, isLazyLoading=True)
    train_dataset = ti_digits.train_dataset
    mc = ModelCollection.load_from_files(model_folder)
    candidates = ["".join(c) for c in itertools.product("".join(TI_DIGITS_LABELS.keys()), repeat=n_digits)]
    chosen = random.Random(seed).sample(candidates, n_samples)
    clips = [train_dataset.get_combined(test_data, key=2) for test_data in chosen]
    pred_labels = mc.predict_batch(MFCC.batch_device(clips, sample_rate=16000))
    return chosen, pred_labels


def main(corpus: str, model_folder: str, n_samples: int = 10, seed: int = 0):
    truth, pred = combine_and_predict(corpus, model_folder, 2, n_samples, seed)
    for ground_truth, pred_labels in zip(truth, pred):
        print(f"Predict labels: {''.join(pred_labels)}, ground truth: {ground_truth}")
    return truth, pred


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("corpus")
    ap.add_argument("models")
    args = ap.parse_args()
    main(args.corpus, args.models)
