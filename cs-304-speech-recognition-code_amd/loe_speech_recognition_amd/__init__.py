"""MI355X-native drop-in for the numeric hot path of `loe_speech_recognition`:
MFCC front-end, Gaussian emission scores, Viterbi decoding and hard-alignment training
statistics, computed by hand-written HIP kernels in liblsr_hip.so (see include/lsr_hip.h).

The class and method names follow the reference package so that its driver scripts work
with only the import changed; every class also has a batched entry point."""
from .mfcc import MFCC
from .transition_probability import SparseMatrix, TransitionProbabilities, LogTransitionProbabilities
from .model_boundary import ModelBoundary
from .signal import Signal, SortedSignals
from .hidden_markov_model import (MultivariateNormal, GaussianMixture, HiddenMarkovModel, HiddenMarkovModelTrainable,
                                  HiddenMarkovModelInference, HiddenMarkovModelMultiWord,
                                  HiddenMarkovModelTrainContinuous)
from .model_collection import ModelCollection
from .ti_digits import TIDigits, DataLoader, TI_DIGITS_LABELS, TI_DIGITS_LABEL_TYPE
from .signal_separation import SignalSeparation
from . import _native, distributed, synthetic

__all__ = ["MFCC", "SparseMatrix", "TransitionProbabilities", "LogTransitionProbabilities", "ModelBoundary", "Signal",
           "SortedSignals", "MultivariateNormal", "GaussianMixture", "HiddenMarkovModel", "HiddenMarkovModelTrainable",
           "HiddenMarkovModelInference", "HiddenMarkovModelMultiWord", "HiddenMarkovModelTrainContinuous",
           "ModelCollection", "TI_DIGITS_LABELS", "TI_DIGITS_LABEL_TYPE", "TIDigits", "DataLoader", "SignalSeparation"]
