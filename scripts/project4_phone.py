"""The flow of the reference's scripts/project4_phone.py (lines 17-33): fifty random four-digit strings out of
all 11^4, each a concatenation of one-digit training clips (DataLoader.get_combined(..., key=2)), MFCC, then
the isolated-word classifier on the whole concatenation.  The reference maps mc.predict over a process pool;
here the fifty clips are one device batch (ModelCollection.predict_batch).  As in the reference the classifier
returns one label per clip, so the "phone number" is never recovered: the script documents that flow, the
connected-digit decoder of project5_test_ndigits_with_sil is what reads strings."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from project4_2digits import combine_and_predict      # noqa: E402


def main(corpus: str, model_folder: str, n_samples: int = 50, seed: int = 0):
    truth, pred = combine_and_predict(corpus, model_folder, 4, n_samples, seed)
    for ground_truth, pred_labels in zip(truth, pred):
        print(f"Predict labels: {''.join(pred_labels)}, ground truth: {ground_truth}")
    return truth, pred


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("corpus")
    ap.add_argument("models")
    args = ap.parse_args()
    main(args.corpus, args.models)
