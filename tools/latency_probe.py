"""Single-call latency of the Python surface on one GPU (B = 1 and B = 64): python tools/latency_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cs-304-speech-recognition-code_amd"))
from loe_speech_recognition_amd import synthetic, MFCC, HiddenMarkovModelInference, ModelCollection, _native
sigs, truth = synthetic.batch(3, 64, 4, 0)
one, t1 = synthetic.batch(2, 64, 1, 0)
import torch
models = synthetic.load_word_models()
inf = HiddenMarkovModelInference.from_models([models[l] for l in synthetic.LOOP_ORDER]); inf._log_transition_probability_between_words = -100
mc = ModelCollection(); mc._models = [models[l] for l in synthetic.WORDS]
def T(name, fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): r = fn(i)  if fn.__code__.co_argcount else fn()
    torch.cuda.synchronize(); print(f"{name:40s} {(time.perf_counter()-t)/n*1e3:8.3f} ms", r if isinstance(r, str) else ""); return r
T("MFCC(signal).feature_vector", lambda i=0: MFCC(sigs[i % 64], 16000).feature_vector)
feats = [MFCC(s, 16000).feature_vector.T for s in sigs]
T("inf.predict(features)", lambda i=0: inf.predict(feats[i % 64]))
f1 = [MFCC(s, 16000).feature_vector.T for s in one]
T("ModelCollection.predict(features)", lambda i=0: mc.predict(f1[i % 64]))
T("recognize_audio([clip])", lambda i=0: inf.recognize_audio([sigs[i % 64]])[0])
T("recognize_audio(64 clips)", lambda: inf.recognize_audio(sigs), n=20)
# a bench-sized batch from host arrays to Python strings (the upload pipeline is chosen automatically)
import bench  # noqa: E402

big, _ = bench.generate(0, 10000, 8)
big16 = [np.round(c).astype(np.int16) for c in big]
T("recognize_audio(10 000 float32 clips)", lambda: inf.recognize_audio(big), n=5)
T("recognize_audio(10 000 int16 clips)", lambda: inf.recognize_audio(big16), n=5)
T("recognize_audio(10 000, streamed=False)", lambda: inf.recognize_audio(big, streamed=False), n=3)
T("MFCC.batch(10 000 clips) -> host arrays", lambda: MFCC.batch(big, 16000), n=3)
fl = MFCC.batch(big, 16000)
T("predict_batch(10 000 host feature arrays)", lambda: inf.predict_batch(fl), n=3)
