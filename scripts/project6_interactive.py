"""The file-based half of the reference's scripts/project6_interactive.py (lines 28-39): a recorded wav file
-> float32 samples -> MFCC -> connected-digit decode with the silence word loaded and a between-word penalty
of -100.  The first half of the reference script records that file from a microphone (Segmentation.main, audio
hardware); here the path of an existing recording is given instead."""
import argparse
import os
import sys
from typing import List

import numpy as np
import scipy.io.wavfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import MFCC, TI_DIGITS_LABELS, HiddenMarkovModelInference      # noqa: E402


def main(wav_path: str, model_folder: str) -> str:
    sr, raw_signal = scipy.io.wavfile.read(wav_path)
    raw_signal_float = raw_signal.astype(np.float32)
    signal = MFCC(raw_signal_float, sr).feature_vector.T

    models_to_load: List[str] = list(TI_DIGITS_LABELS.keys())
    models_to_load.append("S")      # load silence
    hmm_inference = HiddenMarkovModelInference.from_folder(model_folder, models_to_load)
    hmm_inference._log_transition_probability_between_words = -100
    result = hmm_inference.predict(signal)
    print(f"Predict result: {result}")
    return result


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("wav")
    ap.add_argument("models")
    args = ap.parse_args()
    main(args.wav, args.models)
