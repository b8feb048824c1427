"""TI-Digits corpus access with the reference's interface (src/loe_speech_recognition/ti_digits.py):
label table, ``DataLoader`` (label -> clips, lazily read from wav files) and ``TIDigits`` (train/test
x adults/children).  File handling is host work; what changes for the device path is the sample
type: ``DataLoader(..., keep_int16=True)`` hands the clips over as the int16 PCM the wav files hold,
which the device front-end converts itself (half the bytes to upload), instead of the float32 copies
the reference makes (:130-134)."""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass, field
from typing import Any, Dict, Generator, List, Literal, Tuple, Union

import numpy as np
import scipy.io.wavfile

logger = logging.getLogger(__name__)

TI_DIGITS_LABEL_TYPE = Literal["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "Z"]
TI_DIGITS_LABELS: Dict[str, int] = {"1": 1, "2": 2, "3": 3, "4": 4, "5": 5, "6": 6, "7": 7, "8": 8, "9": 9,
                                    "O": 0, "Z": 10}

Clip = Union[str, np.ndarray]


@dataclass
class DataLoader:
    """label (digit string) -> list of clips; a clip is a wav path until it is first read."""
    data: Dict[str, List[Clip]]
    keep_int16: bool = field(default=False)

    def _read(self, clip: Clip) -> np.ndarray:
        if isinstance(clip, np.ndarray):
            return clip
        if not isinstance(clip, str):
            raise NotImplementedError(f"Cannot deal with {type(clip)}")
        samples = scipy.io.wavfile.read(clip)[1]
        if self.keep_int16 and samples.dtype == np.int16:
            return samples
        return samples.astype(np.float32)

    @classmethod
    def lazy_loading(cls, file_path: Clip) -> np.ndarray:
        """float32 samples of a wav file; arrays pass through (ti_digits.py:122-139)."""
        return cls({})._read(file_path)

    def __iter__(self) -> Generator[Tuple[np.ndarray, str], Any, Any]:
        for label, clips in self.data.items():
            for clip in clips:
                yield self._read(clip), label

    def __add__(self, other: "DataLoader") -> "DataLoader":
        merged = self.data                      # the reference extends the left operand's lists in place (:43-50)
        for label, clips in other.data.items():
            if label in merged:
                merged[label].extend(clips)
            else:
                merged[label] = clips
        return type(self)(merged, keep_int16=self.keep_int16 or other.keep_int16)

    def __len__(self) -> int:
        return len(self.data)

    def __getitem__(self, key: str) -> List[np.ndarray]:
        return [self._read(clip) for clip in self.data[key]]

    def get_combined(self, labels: str, key: int = 0) -> np.ndarray:
        """Clip number ``key`` of every single-digit label in ``labels``, concatenated (:67-75)."""
        return np.concatenate([self[label][key] for label in labels])

    def get_all_n_digits(self, n: int) -> Dict[str, List[np.ndarray]]:
        return {label: [self._read(clip) for clip in clips] for label, clips in self.data.items() if len(label) == n}

    @classmethod
    def from_folder_path(cls, folder_path: str, isLazyLoading: bool = True, keep_int16: bool = False) -> "DataLoader":
        """Every ``*.wav`` / ``*.WAV`` below ``folder_path``, grouped by the label in its file name."""
        loader = cls({}, keep_int16=keep_int16)
        # a missing folder yields an empty loader: os.walk reports nothing, as in the reference
        for dirpath, _, filenames in os.walk(folder_path):
            for filename in filenames:
                if not (filename.endswith(".wav") or filename.endswith(".WAV")):
                    continue
                path = os.path.join(dirpath, filename)
                label = cls.filename_parser(filename)
                loader.data.setdefault(label, []).append(path if isLazyLoading else loader._read(path))
        return loader

    @staticmethod
    def filename_parser(file_name: str) -> str:
        """``"1Z3A.WAV"`` -> ``"1Z3"``: the stem minus the production letter (:116-120)."""
        return file_name.split(".")[0][:-1]


@dataclass
class TIDigits:
    # Mains
    folder_path: str

    # Settings
    include_adult: bool = field(default=True)
    include_children: bool = field(default=True)
    include_percentage: float = field(default=1.0)      # accepted and unused, as in the reference
    isLazyLoading: bool = field(default=True)
    keep_int16: bool = field(default=False)

    # Internals
    _train_dataset: DataLoader = field(init=False)
    _test_dataset: DataLoader = field(init=False)

    def __post_init__(self) -> None:
        if not self.include_adult and not self.include_children:
            logger.error("Both Adults and Children are not included")
            raise Exception
        self._train_dataset = DataLoader({}, keep_int16=self.keep_int16)
        self._test_dataset = DataLoader({}, keep_int16=self.keep_int16)
        groups = (["Adults"] if self.include_adult else []) + (["Children"] if self.include_children else [])
        for group in groups:
            root = os.path.join(self.folder_path, group, "TIDIGITS")
            self._train_dataset += DataLoader.from_folder_path(os.path.join(root, "TRAIN"), self.isLazyLoading,
                                                               self.keep_int16)
            self._test_dataset += DataLoader.from_folder_path(os.path.join(root, "TEST"), self.isLazyLoading,
                                                              self.keep_int16)

    @property
    def train_dataset(self) -> DataLoader:
        return self._train_dataset

    @property
    def test_dataset(self) -> DataLoader:
        return self._test_dataset
