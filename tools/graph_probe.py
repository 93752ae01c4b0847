#!/usr/bin/env python3
"""Does replaying the text->mel step as a captured HIP graph shrink the inter-kernel gaps?  (GPU only)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

cfg = get_config("LJSpeech")
dev = "cuda:0"
model = host.CMTotalTTS(cfg, dev).load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))
B, L, T, N = 32, 85, 512, 4
rs = np.random.RandomState(0)
texts = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).to(dev)
lens = torch.full((B,), L, dtype=torch.int64, device=dev)
noise = torch.randn(N + 1, B, 1, T, cfg.n_mels, device=dev)

def step():
    out = model.duration_pitch_energy_net(None, texts, lens, max_mel_len=T)
    return host.sample_with_cond(model, out["cond_ct"], None, N, noise)

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

ref = step(); torch.cuda.synchronize()
print("eager  %.3f ms/step" % bench(step))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        mel = step()
    g.replay(); torch.cuda.synchronize()
    print("graph == eager:", torch.equal(mel, ref))
    print("graph  %.3f ms/step" % bench(g.replay))
except Exception as e:
    print("capture failed:", repr(e)[:300])
