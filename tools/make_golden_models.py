"""Write tests/golden/models_g3/<label>/model.npz: the twelve word models of the G3 fixture saved by the
REAL reference's save() (hidden_markov_model.py:93-115, two pickles per label) and converted with
tools/convert_reference_models.py.  Build container only (the reference is mounted at /root/reference and is
loaded with the same shims as tools/make_golden_hmm.py); the committed files are data - float arrays.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_models.py [out_dir_for_the_pickles]
"""
import importlib
import os
import shutil
import sys
import tempfile
import types
import typing

import numpy as np
import typing_extensions

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
LOOP_LABELS = ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "S", "Z"]


def import_reference(root: str = "/root/reference"):
    """The reference's HMM modules without its package __init__ (which needs librosa, sounddevice and Python >= 3.12)."""
    sys.dont_write_bytecode = True
    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self
    sys.modules.setdefault("uniplot", types.ModuleType("uniplot"))
    pkg = types.ModuleType("loe_speech_recognition")
    pkg.__path__ = [os.path.join(root, "src", "loe_speech_recognition")]
    sys.modules["loe_speech_recognition"] = pkg
    return (importlib.import_module("loe_speech_recognition.hidden_markov_model"),
            importlib.import_module("loe_speech_recognition.transition_probability"))


def save_with_reference(dst: str, labels=LOOP_LABELS) -> None:
    H, TP = import_reference()
    g = np.load(os.path.join(ROOT, "tests", "golden", "hmm_g3_loop.npz"), allow_pickle=False)
    for lab in labels:
        m = H.HiddenMarkovModelTrainable(lab, isMultiProcessing=False, isTqdm=False)
        m._means, m._covariances = g[lab + "_means"], g[lab + "_cov"]
        with np.errstate(divide="ignore"):
            m._transition_probs = TP.TransitionProbabilities.from_transition_probability(g[lab + "_trans"])
            m._update_inference_weights()
        m.save(dst)


def main():
    from tools.convert_reference_models import convert_tree

    out = os.path.join(ROOT, "tests", "golden", "models_g3")
    tmp = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp()
    save_with_reference(tmp)
    shutil.rmtree(out, ignore_errors=True)
    for path in convert_tree(tmp, out):
        print("wrote", os.path.relpath(path, ROOT))


if __name__ == "__main__":
    main()
