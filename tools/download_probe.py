"""Where MFCC.batch spends its time when the features go back to host arrays (10 000 clips, 348 MB of features):
upload, front-end, read-back into a fresh array, read-back into an array whose pages exist already."""
import ctypes
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cs-304-speech-recognition-code_amd"))
import numpy as np  # noqa: E402

from loe_speech_recognition_amd import synthetic  # noqa: E402


def gen(a):
    return synthetic.batch(*a)


def main():
    with mp.get_context("fork").Pool(16) as pool:
        parts = pool.map(gen, [(3, 250, 4, 250 * k) for k in range(40)])
    sigs = [s for p in parts for s in p[0]]
    from loe_speech_recognition_amd import MFCC, _native
    eng = _native.get_engine()
    t = eng.torch
    MFCC.batch(sigs[:64], 16000)

    def sync():
        t.cuda.synchronize()

    for rep in range(3):
        t0 = time.perf_counter()
        ab = eng.audio_batch(sigs)
        sync()
        t1 = time.perf_counter()
        fb = eng.mfcc(ab)
        sync()
        t2 = time.perf_counter()
        out = eng.to_host(fb.feats)
        t3 = time.perf_counter()
        parts = _native.split_ragged(out, fb.host_off)
        t4 = time.perf_counter()
        nbytes = out.nbytes
        eng._check(eng.lib.lsr_download_host(eng.ctx, eng._ptr(fb.feats), nbytes, ctypes.c_void_p(out.ctypes.data), 0, 0))
        t5 = time.perf_counter()
        print(f"rep {rep}: upload {1e3 * (t1 - t0):.1f} ms, front-end {1e3 * (t2 - t1):.1f} ms, read-back into a fresh array "
              f"{1e3 * (t3 - t2):.1f} ms ({nbytes / (t3 - t2) / 1e9:.1f} GB/s), split {1e3 * (t4 - t3):.1f} ms, read-back into "
              f"touched pages {1e3 * (t5 - t4):.1f} ms ({nbytes / (t5 - t4) / 1e9:.1f} GB/s)", flush=True)
        del parts, out
    t0 = time.perf_counter()
    feats = MFCC.batch(sigs, 16000)
    print(f"MFCC.batch whole call: {1e3 * (time.perf_counter() - t0):.1f} ms for {len(feats)} clips")


if __name__ == "__main__":
    main()
