"""The reference's driver-script flows (train on silence-stripped clips -> connected-digit test ->
penalty sweep) run end to end on the device path over a miniature synthetic corpus on disk."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SCRIPTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts")


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(SCRIPTS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_train_test_sweep(tmp_path):
    corpus = load("make_synthetic_corpus").write_corpus(str(tmp_path / "corpus"))
    models = load("project5_train_no_empty").main(corpus, str(tmp_path / "cache"), max_iterations=30)
    assert sorted(os.listdir(models)) == ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "S", "Z"]
    test = load("project5_test_ndigits_with_sil")
    for n in (2, 4):
        truth, prediction, accuracy = test.main(corpus, models, n=n)
        assert len(truth) == 12 and accuracy >= 0.9, (n, list(zip(truth, prediction)))
    penalties, results = load("project5_find_trans_ndigits_with_sil").main(corpus, models, penalties=[0, -50, -100, -1e5])
    assert len(results) == 4 and results[2] >= 90.0 and results[3] == 0.0      # -1e5 forbids leaving the first word
