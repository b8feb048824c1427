"""Isolated-word training on silence-stripped clips, the flow of the reference's
scripts/project5_train_no_empty.py on the device path: strip silence from every one-digit training
clip (one device scan per label), MFCC, segmental k-means per word, then a three-state silence model
from the noise the stripper collected."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import (MFCC, TI_DIGITS_LABELS, HiddenMarkovModelTrainable, SignalSeparation,   # noqa: E402
                                        TIDigits)


def main(corpus: str, cache: str, model_name: str = "big_model_speech_only", max_iterations: int = 100) -> str:
    out = os.path.join(cache, model_name)
    train = TIDigits(corpus, isLazyLoading=True).train_dataset
    stripper = SignalSeparation(sample_rate=16000, speech_high_threshold=0.06, speech_low_threshold=0.01)
    for label in TI_DIGITS_LABELS:
        speech = stripper.remove_empty_batch(train[label])
        mfccs = MFCC.batch_device(speech, sample_rate=16000)                # features stay on the device
        HiddenMarkovModelTrainable.from_data(label, mfccs, num_of_states=5, max_iterations=max_iterations,
                                             isTqdm=False).save(out)
    noise = [n for n in stripper.get_all_noises() if n.shape[0] >= 9 * 160]       # MFCC needs 9 frames
    HiddenMarkovModelTrainable.from_data("S", MFCC.batch_device(noise, sample_rate=16000), num_of_states=3,
                                         max_iterations=max_iterations, isTqdm=False).save(out)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("corpus")
    ap.add_argument("--cache", default=".cache")
    args = ap.parse_args()
    print("models written to", main(args.corpus, args.cache))
