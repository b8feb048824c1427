"""Timing of the batched covariance factorisation (58 states of 39 x 39) against the serial SciPy / NumPy path,
by thread count, with and without OpenBLAS' own threads.  Host only."""
import os
import sys
import time

import numpy as np
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import hidden_markov_model as H  # noqa: E402

rng = np.random.default_rng(1)
S, D = 58, 39
A = rng.standard_normal((S, D, D))
cov = (A @ A.transpose(0, 2, 1) / D + 1e-3 * np.eye(D)).astype(np.float32)
mu = rng.standard_normal((S, D)).astype(np.float32)


def best(fn, n=15):
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t) * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]


print("OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"), "cpu_count", os.cpu_count())
ref = [H.MultivariateNormal.from_means_covariances(m, c) for m, c in zip(mu, cov)]
print("serial scipy eigh x58: min %.2f med %.2f ms" % best(lambda: [scipy.linalg.eigh(c.astype(float), lower=True) for c in cov], 5))
for th in (1, 2, 4, 8, 16):
    print("threads %2d: min %.2f med %.2f ms" % ((th,) + best(lambda: H.factor_covariances(mu, cov, threads=th))))
got = H.factor_covariances(mu, cov)
print("bits equal:", all(np.array_equal(a.whiten, b.whiten) and a.log_pdet == b.log_pdet for a, b in zip(ref, got)))
try:
    from threadpoolctl import ThreadpoolController

    c = ThreadpoolController()
    print([(i["internal_api"], i["num_threads"]) for i in c.info()])
    with c.limit(limits=1, user_api="blas"):
        for th in (1, 4, 8, 16):
            print("blas=1 threads %2d: min %.2f med %.2f ms" % ((th,) + best(lambda: H.factor_covariances(mu, cov, threads=th))))
        got = H.factor_covariances(mu, cov)
        print("bits equal:", all(np.array_equal(a.whiten, b.whiten) and a.log_pdet == b.log_pdet for a, b in zip(ref, got)))
except ImportError:
    pass
