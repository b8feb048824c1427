"""Isolated-digit classification over a corpus folder, the flow of the reference's
scripts/project3_predict_simple.py (lines 14-50): for every label of the seen (train) and unseen (test)
split, MFCC.batch of the label's clips and ModelCollection.predict on each.  The reference maps
mc.predict over a process pool; here all clips of a label are scored by the eleven word models in one
device call (ModelCollection.predict_batch).  The confusion-matrix plots of the reference are left out."""
import argparse
import os
import sys
from typing import List, Tuple

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import MFCC, TI_DIGITS_LABELS, DataLoader, ModelCollection, TIDigits      # noqa: E402


def make_prediction(mc: ModelCollection, dataset: DataLoader) -> Tuple[List[str], List[str]]:
    truth: List[str] = []
    pred: List[str] = []
    for label in TI_DIGITS_LABELS:
        clips = dataset[label]
        if not clips:
            continue
        pred.extend(mc.predict_batch(MFCC.batch_device(clips, sample_rate=16000)))
        truth.extend([label] * len(clips))
    return truth, pred


def main(corpus: str, model_folder: str):
    ti_digits = TIDigits(corpus, isLazyLoading=True)
    mc = ModelCollection.load_from_files(model_folder)
    out = {}
    for name, dataset in (("seen", ti_digits.train_dataset), ("unseen", ti_digits.test_dataset)):
        truth, pred = make_prediction(mc, dataset)
        accuracy = sum(p == t for p, t in zip(pred, truth)) / max(1, len(pred))
        print(f"Accuracy of the {name} data is {accuracy * 100:2f}%")
        out[name] = (truth, pred, accuracy)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("corpus")
    ap.add_argument("models")
    args = ap.parse_args()
    main(args.corpus, args.models)
