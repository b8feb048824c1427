"""The corpus loader mirror (ti_digits.py of the reference: file name -> label, train/test x
adults/children, lazy reading) on a miniature corpus written by the test itself."""
import os

import numpy as np
import pytest
import scipy.io.wavfile

from loe_speech_recognition_amd import DataLoader, TIDigits, TI_DIGITS_LABELS


def write_corpus(root):
    rng = np.random.default_rng(0)
    made = {}
    layout = {("Adults", "TRAIN"): ["1A.WAV", "1B.WAV", "Z9A.WAV", "O12A.WAV"],
              ("Adults", "TEST"): ["2A.WAV", "55A.WAV"],
              ("Children", "TRAIN"): ["1A.wav", "Z9B.wav", "notes.txt"],
              ("Children", "TEST"): ["2B.wav"]}
    for (group, split), names in layout.items():
        folder = os.path.join(root, group, "TIDIGITS", split, "spk", "x")
        os.makedirs(folder)
        for name in names:
            path = os.path.join(folder, name)
            if name.endswith(".txt"):
                open(path, "w").write("not audio")
                continue
            pcm = rng.integers(-3000, 3000, size=int(rng.integers(200, 400))).astype(np.int16)
            scipy.io.wavfile.write(path, 16000, pcm)
            made[(group, split, name)] = pcm
    return made


def test_labels_and_filename_parser():
    assert list(TI_DIGITS_LABELS) == ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "Z"]
    assert DataLoader.filename_parser("1Z3A.WAV") == "1Z3"
    assert DataLoader.filename_parser("OB.wav") == "O"


def test_splits_and_lazy_loading(tmp_path):
    made = write_corpus(str(tmp_path))
    ti = TIDigits(str(tmp_path), isLazyLoading=True)
    train, test = ti.train_dataset, ti.test_dataset
    assert sorted(train.data) == ["1", "O12", "Z9"] and sorted(test.data) == ["2", "55"]
    assert len(train) == 3 and len(train.data["1"]) == 3 and len(train.data["Z9"]) == 2
    assert all(isinstance(c, str) for c in train.data["1"])            # still paths
    clips = train["1"]
    assert all(c.dtype == np.float32 for c in clips)
    want = sorted(made[k].astype(np.float32).tobytes() for k in made if k[1] == "TRAIN" and k[2][:-5] == "1")
    assert sorted(c.tobytes() for c in clips) == want
    two = train.get_all_n_digits(2)
    assert list(two) == ["Z9"] and len(two["Z9"]) == 2
    assert sum(1 for _ in train) == 6
    assert {lab for _, lab in test} == {"2", "55"}
    only_adults = TIDigits(str(tmp_path), include_children=False)
    assert len(only_adults.train_dataset.data["1"]) == 2
    with pytest.raises(Exception):
        TIDigits(str(tmp_path), include_adult=False, include_children=False)


def test_eager_int16_and_combination(tmp_path):
    made = write_corpus(str(tmp_path))
    ti = TIDigits(str(tmp_path), include_children=False, isLazyLoading=False, keep_int16=True)
    clips = ti.train_dataset.data["1"]
    assert all(isinstance(c, np.ndarray) and c.dtype == np.int16 for c in clips)
    combined = ti.train_dataset.get_combined("11", key=1)
    assert np.array_equal(combined, np.concatenate([clips[1], clips[1]]))
    a = DataLoader({"1": [np.zeros(3, np.float32)]})
    b = DataLoader({"1": [np.ones(2, np.float32)], "2": [np.ones(4, np.float32)]})
    c = a + b
    assert len(c.data["1"]) == 2 and len(c.data["2"]) == 1 and c.data is a.data     # in place, like the reference
    assert DataLoader.lazy_loading(np.arange(3)) is not None
    with pytest.raises(NotImplementedError):
        DataLoader.lazy_loading(3.5)
    assert DataLoader.from_folder_path(os.path.join(str(tmp_path), "missing")).data == {}
