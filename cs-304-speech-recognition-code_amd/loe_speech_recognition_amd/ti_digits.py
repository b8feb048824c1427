"""Label table of the TI-Digits corpus (src/loe_speech_recognition/ti_digits.py:13-26).
The corpus loader itself is outside the accelerated path."""
from typing import Dict

TI_DIGITS_LABELS: Dict[str, int] = {"1": 1, "2": 2, "3": 3, "4": 4, "5": 5, "6": 6, "7": 7, "8": 8, "9": 9,
                                    "O": 0, "Z": 10}
