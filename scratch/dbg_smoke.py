import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_forward_gpu as TF
from conftest import sub
synth, params = sub("synthetic"), sub("params")
for (B, W, N) in [(8, 3, 128), (8, 3, 160), (8, 2, 128), (16, 3, 128)]:
    try:
        TF.test_encoder_train_slots_match_oracle(synth, params, B, W, N)
        print(B, W, N, "encoder train forward OK")
    except AssertionError as e:
        print(B, W, N, "FAIL", str(e)[:300])
