import os, sys, time, numpy as np, multiprocessing as mp
sys.path.insert(0, "cs-304-speech-recognition-code_amd")
from loe_speech_recognition_amd import _lsr_pyhelper as H
import scipy.linalg.cython_lapack as cl
caps = cl.__pyx_capi__
rng = np.random.default_rng(1); S, D = 58, 39
A = rng.standard_normal((S, D, D)); cov = np.ascontiguousarray((A @ A.transpose(0, 2, 1) / D + 1e-3 * np.eye(D)).astype(np.float32).astype(float))
def work(args):
    r, w, th, bar = args
    lam = np.zeros((S, D)); U = np.zeros((S, D, D)); L = np.zeros((S, D, D)); st = np.zeros(S, np.int32)
    H.factor_batch(caps["dsyevr"], caps["dgeqrf"], cov, lam, U, L, st, S, D, r, w, th)
    ts = []
    for _ in range(20):
        bar.wait()
        t = time.perf_counter(); H.factor_batch(caps["dsyevr"], caps["dgeqrf"], cov, lam, U, L, st, S, D, r, w, th); ts.append((time.perf_counter() - t) * 1e3)
    return min(ts), sorted(ts)[10]
for procs, th in ((1, 4), (2, 4), (4, 4), (4, 2), (8, 2), (8, 1), (16, 1)):
    m = mp.Manager(); bar = m.Barrier(procs)
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(work, [(r, procs, th, bar) for r in range(procs)])
    print(procs, "procs x", th, "threads: slowest min %.2f med %.2f ms" % (max(r[0] for r in res), max(r[1] for r in res)))
