import sys, os, time, cProfile, pstats
sys.path.insert(0,'cs-304-speech-recognition-code_amd'); sys.path.insert(0,'.')
import numpy as np, multiprocessing as mp
from loe_speech_recognition_amd import synthetic
def gen(a): return synthetic.batch(*a)
with mp.get_context("fork").Pool(16) as pool:
    parts = pool.map(gen, [(4, 240, [1,2,3,4,5,7], 240*k) for k in range(50)])
sigs=[s for p in parts for s in p[0]]; truth=[d for p in parts for d in p[1]]
import torch
from loe_speech_recognition_amd import HiddenMarkovModelTrainContinuous, _native
eng=_native.get_engine(); models=synthetic.load_word_models()
tc=HiddenMarkovModelTrainContinuous(isTqdm=False)
for lab in synthetic.WORDS+["S"]: tc.add_model(models[lab])
fb=eng.mfcc(eng.audio_batch(sigs)); labeled={"x":None}
tc.use_device_batch(fb, truth); labeled = None
tc._iteration(labeled); torch.cuda.synchronize()
eng.profile_reset(); eng.profile_enable(True)
pr=cProfile.Profile(); t=time.time(); pr.enable()
for _ in range(3): tc._iteration(labeled)
torch.cuda.synchronize(); pr.disable(); print('per iteration', (time.time()-t)/3)
for k in ("emission","viterbi","stats"): print(k, eng.profile_get(k))
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
