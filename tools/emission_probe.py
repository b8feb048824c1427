"""Time per (64-frame tile, state) pair of the all-states emission kernel and of the reachable-states form, on
the config-3 batch, for chains that hold all 58 states, 23 states in runs of 5, and 23 scattered states."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import synthetic  # noqa: E402


def gen(a):
    return synthetic.batch(*a)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    with mp.get_context("fork").Pool(16) as pool:
        parts = pool.map(gen, [(3, 250, 4, 250 * k) for k in range(n // 250)])
    sigs = [s for p in parts for s in p[0]]
    from loe_speech_recognition_amd import HiddenMarkovModelInference, _native
    eng = _native.get_engine()
    models = synthetic.load_word_models()
    inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
    inf._bind()
    fb = eng.mfcc(eng.audio_batch(sigs))
    tiles_full = -(-fb.total_frames // 64)
    tiles_anch = int(sum(-(-int(t) // 64) for t in fb.lengths))

    def run(label, fn, pairs):
        for _ in range(3):
            fn()
        eng.profile_reset()
        eng.profile_enable(True)
        for _ in range(10):
            fn()
        ms, k = eng.profile_get("emission")
        eng.profile_enable(False)
        print(f"{label}: {ms / k:.3f} ms, {pairs} pairs, {ms / k * 1e6 / pairs:.3f} ns per pair")

    run("all-states kernel", lambda: eng.emission(fb), tiles_full * 58)
    rng = np.random.default_rng(0)
    for label, states in (("reachable, chain of all 58 states", np.arange(58)),
                          ("reachable, 23 states in runs (S + 4 digits)", np.concatenate([np.arange(50, 53), np.arange(0, 20)])),
                          ("reachable, 23 states as one run", np.arange(10, 33)),
                          ("reachable, 23 scattered states", np.sort(rng.choice(58, 23, replace=False))),
                          ("reachable, 8 states", np.arange(8))):
        k = states.shape[0]
        band = np.zeros((3, k), np.float32)
        start = np.zeros(k, np.uint8)
        start[0] = 1
        d = eng.uniform_chains(("probe", label), fb.n, states.astype(np.int32), band, start)
        eng.chain_emission_mode("reachable")
        run(label, lambda: eng.decode_chain_device(fb, *d, want_paths=False), tiles_anch * k)
        eng.chain_emission_mode("auto")
    print("tiles", tiles_full, "anchored", tiles_anch)


if __name__ == "__main__":
    main()
