#!/usr/bin/env python3
"""A/B inside one process, interleaved rounds: the bench step (B = 32, 80 x 512, T = 4) with the conditioner factors gathered inside the persistent
kernel (cond_inkernel = 1, FACT instances) against expanded into cp first (0) against the dense conditioner GEMM (cond_factored = 0)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))
rs = np.random.RandomState(0)
B, L, T = 32, 85, 512
texts = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).cuda()
lens = torch.full((B,), L, dtype=torch.int64, device="cuda")
def step():
    out = model.duration_pitch_energy_net(None, texts, lens, max_mel_len=T)
    nz = torch.randn(5, B, 1, T, cfg.n_mels, device="cuda")
    return host.sample_with_cond(model, out["cond_ct"], None, 4, nz)
def timed(n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
modes = {"inkernel": (1, 1), "expand": (1, 0), "dense": (0, 0)}
res = {k: [] for k in modes}
for k, (f, ik) in modes.items():
    _lib.internal_set("cond_factored", f); _lib.internal_set("cond_inkernel", ik); timed(3)
for r in range(6):
    for k, (f, ik) in modes.items():
        _lib.internal_set("cond_factored", f); _lib.internal_set("cond_inkernel", ik)
        res[k].append(timed())
for k, v in res.items():
    print(f"{k:9s} median {np.median(v):.3f} ms  min {min(v):.3f}  rounds {[round(x, 3) for x in v]}")
