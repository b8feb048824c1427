"""Where one continuous-training iteration (config 4: 12 000 clips of 1-7 digits) spends its time: wall time per
iteration, HIP-event kernel times, and the host phases timed with perf_counter around the trainer's own methods."""
import collections
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cs-304-speech-recognition-code_amd"))
sys.path.insert(0, ROOT)
from loe_speech_recognition_amd import synthetic  # noqa: E402


def gen(a):
    return synthetic.batch(*a)


def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    with mp.get_context("fork").Pool(16) as pool:
        parts = pool.map(gen, [(4, 240, [1, 2, 3, 4, 5, 7], 240 * k) for k in range(50)])
    sigs = [s for p in parts for s in p[0]]
    truth = [d for p in parts for d in p[1]]
    import torch

    from loe_speech_recognition_amd import HiddenMarkovModelTrainContinuous, _native
    from loe_speech_recognition_amd import hidden_markov_model as H

    eng = _native.get_engine()
    models = synthetic.load_word_models()
    tc = HiddenMarkovModelTrainContinuous(isTqdm=False)
    for lab in synthetic.WORDS + ["S"]:
        tc.add_model(models[lab])
    tc.use_device_batch(eng.mfcc(eng.audio_batch(sigs)), truth)
    spent = collections.defaultdict(float)

    def timed(owner, name, label):
        fn = getattr(owner, name)

        def wrapper(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                spent[label] += time.perf_counter() - t
        setattr(owner, name, wrapper)

    timed(tc, "_align", "align (model upload, band, launches)")
    timed(tc, "_m_steps", "m_steps (all)")
    timed(tc, "_refresh_inference_weights", "  of which refresh_inference_weights")
    timed(H, "factor_covariances", "    of which factor_covariances")
    timed(eng, "to_host_pinned", "wait for the device + copy of the statistics")
    timed(eng, "remux_slots", "remux launch")
    timed(eng, "accumulate_stats", "stats launch")
    timed(eng, "set_model", "  set_model (inside align)")
    for _ in range(3):
        tc._iteration(None)
    torch.cuda.synchronize()
    spent.clear()
    eng.profile_reset()
    eng.profile_enable(True)
    t = time.perf_counter()
    factored = []
    for _ in range(n_iter):
        tc._iteration(None)
        factored.append(tc.__dict__.get("_factored_states"))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / n_iter
    eng.profile_enable(False)
    print(f"per iteration {wall * 1e3:.3f} ms over {n_iter} iterations, {len(sigs)} clips; states factored per iteration "
          f"(models whose means moved): {factored}; models that reported convergence so far: {tc._num_of_finished_models}")
    total_k = 0.0
    for k in ("emission", "viterbi", "stats"):
        ms, n = eng.profile_get(k)
        total_k += ms / n_iter
        print(f"  kernel {k}: {ms / n_iter:.3f} ms per iteration ({n} launches)")
    print(f"  kernels together {total_k:.3f} ms")
    # work of the reachable-states emission pass: tiles anchored at utterance starts x states of the utterance's chain
    fb = tc.__dict__["_data_cache"]["fb"]
    sizes = {lab: tc._trainable_models[lab].num_of_states for lab in tc._models_loaded}
    pairs = sum(-(-int(t) // 64) * (sizes["S"] + sum(sizes[c] for c in set(d))) for t, d in zip(fb.lengths, truth))
    full = -(-fb.total_frames // 64) * sum(sizes.values())
    ms, n = eng.profile_get("emission")
    print(f"  emission work: {pairs} (tile, state) pairs = {pairs / full:.3f} of the all-states form ({full}); "
          f"{ms / n * 1e6 / pairs:.3f} ns per pair")
    for label, s in spent.items():
        print(f"  host {label}: {s / n_iter * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
