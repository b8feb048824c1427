"""Time lsr_accumulate_stats on an alignment-like slot sequence (runs of equal slot), config-4 size."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cs-304-speech-recognition-code_amd"))
import numpy as np
import torch
from loe_speech_recognition_amd import _native

n, n_slots = 2_468_561, 58
rng = np.random.default_rng(0)
x = rng.standard_normal((n, 39)).astype(np.float32)
runs = rng.integers(3, 25, size=n // 3)
slots = np.repeat(rng.integers(-1, n_slots, size=runs.shape[0]), runs)[:n].astype(np.int32)
eng = _native.get_engine()
dx, ds = eng.to_device(x), eng.to_device(slots)
for _ in range(2):
    eng.accumulate_stats(dx, ds, n, n_slots)
torch.cuda.synchronize()
eng.profile_reset(); eng.profile_enable(True)
for _ in range(5):
    eng.accumulate_stats(dx, ds, n, n_slots)
ms, k = eng.profile_get("stats")
print(f"stats: {ms / k:.3f} ms per call over {n} frames, {n_slots} slots")
