#!/usr/bin/env python3
"""A/B in one process: the FFT blocks' out-projection on the generic 64x64 kernel (text_xres = 5) against conv_xres (7: with the launcher's rule that
is the 32-column instance for M = 256), B = 32 / 64 / 8."""
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
for B, L in ((32, 85), (64, 85), (32, 171), (8, 85)):
    texts = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).cuda()
    lens = torch.full((B,), L, dtype=torch.int64, device="cuda")
    ref, res = None, {}
    for r in range(4):
        for bits in (5, 7):
            _lib.internal_set("text_xres", bits)
            for _ in range(2): o = model.duration_pitch_energy_net(None, texts, lens, max_mel_len=6 * L)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): o = model.duration_pitch_energy_net(None, texts, lens, max_mel_len=6 * L)
            torch.cuda.synchronize(); res.setdefault(bits, []).append((time.perf_counter() - t0) / 20 * 1e3)
            if ref is None: ref = o["cond_ct"].clone()
            assert torch.equal(o["cond_ct"], ref)
    _lib.internal_set("text_xres", 5)
    print(f"B={B} L={L}: out-projection generic {np.median(res[5]):.3f} ms, conv_xres {np.median(res[7]):.3f} ms (same bits)")
