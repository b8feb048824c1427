"""Model import: folders saved by the reference's own save() become model.npz files with the
same numbers.  Needs the reference sources (build container only); skipped elsewhere."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference sources not mounted")

_MAKE = r'''
import sys, os
sys.dont_write_bytecode = True
sys.path.insert(0, sys.argv[1])
import numpy as np
from tools.convert_reference_models import import_reference
H = import_reference("/root/reference")
import importlib
TP = importlib.import_module("loe_speech_recognition.transition_probability")
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "hmm_g3_loop.npz"))
for lab in ("3", "S"):
    m = H.HiddenMarkovModelTrainable(lab, isMultiProcessing=False, isTqdm=False)
    m._means, m._covariances = g[lab + "_means"], g[lab + "_cov"]
    m._transition_probs = TP.TransitionProbabilities.from_transition_probability(g[lab + "_trans"])
    m._update_inference_weights()
    m.save(sys.argv[2])
'''


def test_reference_pickles_convert_to_npz(tmp_path):
    src, dst = tmp_path / "ref_models", tmp_path / "npz_models"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, "-c", _MAKE, ROOT, str(src)], check=True, env=env, cwd=ROOT)
    assert (src / "3" / "multivariate_normals.pickle").exists()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_reference_models.py"), "--reference", REF,
                    "--src", str(src), "--dst", str(dst)], check=True, env=env, cwd=ROOT)
    from helpers import load, word
    from loe_speech_recognition_amd import HiddenMarkovModel

    g = load("hmm_g3_loop.npz")
    for lab in ("3", "S"):
        got = HiddenMarkovModel.from_folder(str(dst / lab))
        ref = word(g, lab)
        assert got.label == lab and got.num_of_states == ref.num_of_states
        for a, b in zip(got._multivariate_normals, ref._multivariate_normals):
            assert np.array_equal(a.mean, b.mean) and np.array_equal(a.whiten, b.whiten) and a.log_pdet == b.log_pdet
        assert np.array_equal(got._log_transition_probs.band(), ref._log_transition_probs.band())
