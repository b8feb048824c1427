"""Isolated-word classifier with the reference's interface
(src/loe_speech_recognition/model_collection.py): every word model scores the utterance with
its own left-to-right Viterbi pass and the best score wins.  All models of the collection run
side by side in one wavefront per utterance (lsr_decode_chain_batch with one start state per
model and no transitions between models)."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import _native
from .hidden_markov_model import HiddenMarkovModel, stack_gaussians
from .ti_digits import TI_DIGITS_LABELS


@dataclass
class ModelCollection:
    num_of_states: int = field(default=5)
    dim_of_feature: int = field(default=39)

    _models: List[HiddenMarkovModel] = field(default_factory=list)

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_dev_cache", None)
        return state

    def _bind(self):
        eng = _native.get_engine()
        cache = self.__dict__.get("_dev_cache")
        # the upload belongs to these very parameter objects: a replaced model (or a re-estimated one, which
        # gets new lists) invalidates it even when the number of models is unchanged
        srcs = [(m._multivariate_normals, m._log_transition_probs) for m in self._models]
        if cache is None or len(cache["srcs"]) != len(srcs) or \
                any(a[0] is not b[0] or a[1] is not b[1] for a, b in zip(cache["srcs"], srcs)):
            sizes = [m.num_of_states for m in self._models]
            if sum(sizes) > _native.NARROW_STATES:
                raise ValueError(f"{sum(sizes)} states in the collection; scoring all models side by side handles "
                                 f"{_native.NARROW_STATES}")
            band = np.concatenate([m._log_transition_probs.band() for m in self._models], axis=1)
            ends = np.cumsum(sizes)
            start = np.zeros(int(ends[-1]), dtype=np.uint8)
            start[np.concatenate([[0], ends[:-1]])] = 1
            cache = {"srcs": srcs, "token": _native.new_token(),
                     "model": stack_gaussians([g for m in self._models for g in m._multivariate_normals]),
                     "band": band, "start": start, "last": ends - 1}
            self.__dict__["_dev_cache"] = cache
        eng.set_model(cache["token"], *cache["model"])
        return eng, cache

    def scores_device(self, signals):
        """Device tensor float32 [B, n_models]: each model's Viterbi score, left on the device."""
        eng, c = self._bind()
        fb = signals if isinstance(signals, _native.FeatureBatch) else eng.feature_batch(signals)
        n = c["start"].shape[0]
        d_off, total, d_ids, d_band, d_start = eng.uniform_chains(c["token"], fb.n, np.arange(n, dtype=np.int32), c["band"],
                                                                  c["start"])
        _, _, final = eng.decode_chain_device(fb, d_off, total, d_ids, d_band, d_start, want_paths=False, want_final=True)
        if "d_last" not in c:
            c["d_last"] = eng.to_device(np.asarray(c["last"], dtype=np.int64))
        return final[:, c["d_last"]]

    def scores_batch(self, signals) -> np.ndarray:
        """float32 [B, n_models]: each model's Viterbi score (hidden_markov_model.py:160-198)."""
        return self.scores_device(signals).cpu().numpy()

    def predict_batch(self, signals) -> List[str]:
        scores = self.scores_batch(signals)
        # model_collection.py:26-28 sorts (label, score) pairs with a stable sort, reverse=True:
        # among equal scores the model that comes first in the collection wins = first argmax
        names = [str(m) for m in self._models]
        return [names[i] for i in np.argmax(scores, axis=1).tolist()]

    def predict(self, signal: np.ndarray) -> str:
        return self.predict_batch([signal])[0]

    @classmethod
    def load_from_files(cls, folder_path: str):
        mc = cls()
        for label in TI_DIGITS_LABELS:
            mc._models.append(HiddenMarkovModel.from_folder(os.path.join(folder_path, f"{label}")))
        return mc
