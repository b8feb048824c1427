"""Model import: folders in the reference's save() layout (two pickles per label) become model.npz files with the
same numbers, read through a restricted unpickler that neither imports the reference nor accepts other globals."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import scipy.stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from helpers import LOOP_LABELS, load, word  # noqa: E402
from tools import convert_reference_models as C  # noqa: E402

_MAKE = r'''
import sys
sys.path.insert(0, sys.argv[1])
from tools.make_golden_models import save_with_reference
save_with_reference(sys.argv[2], ["3", "S"])
'''


def _same_model(got, ref):
    assert got.num_of_states == ref.num_of_states
    for a, b in zip(got._multivariate_normals, ref._multivariate_normals):
        assert np.array_equal(a.mean, b.mean) and np.array_equal(a.whiten, b.whiten) and a.log_pdet == b.log_pdet
    assert np.array_equal(got._log_transition_probs.band(), ref._log_transition_probs.band())


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference sources not mounted")
def test_reference_pickles_convert_to_npz(tmp_path):
    """Pickles written by the reference's own save() (run in a child process) convert without the reference."""
    from loe_speech_recognition_amd import HiddenMarkovModel

    src, dst = tmp_path / "ref_models", tmp_path / "npz_models"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, "-c", _MAKE, ROOT, str(src)], check=True, env=env, cwd=ROOT)
    assert (src / "3" / "multivariate_normals.pickle").exists()
    assert "loe_speech_recognition" not in sys.modules
    C.convert_tree(str(src), str(dst))
    assert "loe_speech_recognition" not in sys.modules          # the converter did not import the reference
    g = load("hmm_g3_loop.npz")
    for lab in ("3", "S"):
        got = HiddenMarkovModel.from_folder(str(dst / lab))
        assert got.label == lab
        _same_model(got, word(g, lab))
        # and the committed fixture is exactly what this conversion produces
        with np.load(os.path.join(ROOT, "tests", "golden", "models_g3", lab, "model.npz")) as a, \
                np.load(str(dst / lab / "model.npz")) as b:
            assert sorted(a.files) == sorted(b.files) and all(np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f") for k in a.files if a[k].dtype.kind != "U")


def test_committed_converted_models_match_the_fixture_arrays():
    """tests/golden/models_g3 (reference save() -> converter) holds the Gaussians the G3 arrays define."""
    from loe_speech_recognition_amd import HiddenMarkovModel, HiddenMarkovModelInference

    g = load("hmm_g3_loop.npz")
    folder = os.path.join(ROOT, "tests", "golden", "models_g3")
    for lab in LOOP_LABELS:
        _same_model(HiddenMarkovModel.from_folder(os.path.join(folder, lab)), word(g, lab))
    inf = HiddenMarkovModelInference.from_folder(folder, LOOP_LABELS)
    assert inf._model_boundaries._labels == LOOP_LABELS


def _stub_pickles(folder, label, n_states=3):
    """A model folder in the reference's layout written from stand-in classes registered under the reference's names."""
    rng = np.random.default_rng(5)
    ltp = C.LogTransitionProbabilities()
    ltp._core = {(i, j): np.float32(np.log(0.5)) if j - i in (0, 1) else np.float32(-np.inf)
                 for i in range(n_states) for j in range(n_states)}
    mvns = []
    for _ in range(n_states):
        a = rng.standard_normal((39, 39))
        mv = C.MultivariateNormal()
        mv._core = scipy.stats.multivariate_normal(mean=rng.standard_normal(39), cov=a @ a.T / 39 + 1e-3 * np.eye(39))
        mvns.append(mv)
    os.makedirs(os.path.join(folder, label))
    names = {C.LogTransitionProbabilities: ("loe_speech_recognition.transition_probability", "LogTransitionProbabilities"),
             C.MultivariateNormal: ("loe_speech_recognition.hidden_markov_model", "MultivariateNormal")}
    parent = type(sys)("loe_speech_recognition")
    parent.__path__ = []
    sys.modules["loe_speech_recognition"] = parent
    for cls, (mod, name) in names.items():          # pickle them under the names the reference's files carry
        cls.__module__, cls.__qualname__ = mod, name
        sys.modules.setdefault(mod, type(sys)(mod))
        setattr(sys.modules[mod], name, cls)
    try:
        for fname, obj in (("log_trans_probs.pickle", ltp), ("multivariate_normals.pickle", mvns)):
            with open(os.path.join(folder, label, fname), "wb") as f:
                pickle.dump(obj, f)
    finally:
        for cls, (mod, name) in names.items():
            cls.__module__, cls.__qualname__ = C.__name__, cls.__name__
            sys.modules.pop(mod, None)
        sys.modules.pop("loe_speech_recognition", None)
    return ltp, mvns


def test_stub_pickles_round_trip_and_foreign_globals_are_refused(tmp_path):
    from loe_speech_recognition_amd import HiddenMarkovModel

    ltp, mvns = _stub_pickles(str(tmp_path / "src"), "7")
    C.convert_tree(str(tmp_path / "src"), str(tmp_path / "dst"))
    got = HiddenMarkovModel.from_folder(str(tmp_path / "dst" / "7"))
    for g, m in zip(got._multivariate_normals, mvns):
        assert np.array_equal(g.mean, m._core.mean) and np.array_equal(g.whiten, m._core.cov_object._LP)
        assert g.log_pdet == m._core.cov_object._log_pdet
    assert got._log_transition_probs[(0, 1)] == np.float32(np.log(0.5))
    # a file that names anything else is refused before the global is looked up
    evil = tmp_path / "src" / "7" / "log_trans_probs.pickle"
    evil.write_bytes(b"cos\nsystem\n(S'echo pwned'\ntR.")
    with pytest.raises(pickle.UnpicklingError):
        C.convert_tree(str(tmp_path / "src"), str(tmp_path / "dst2"))
    evil.write_bytes(pickle.dumps({"_core": {}}))               # well-formed but not a reference model
    with pytest.raises(ValueError):
        C.convert_tree(str(tmp_path / "src"), str(tmp_path / "dst3"))
