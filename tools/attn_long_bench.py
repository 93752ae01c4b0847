#!/usr/bin/env python3
"""Long sequences through the FFT blocks: fused key-chunked attention (attention_long_kernel) against the three-launch path.
Text side at B x L phonemes and the FastspeechDecoder at B x T frames; median wall time of 20 calls."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict, synth_decoder_state_dict

cfg = get_config("LJSpeech")
sd = synth_cmtts_state_dict(cfg, seed=0, dur_frames=2.0, dur_spread=0.0)
sd.update(synth_decoder_state_dict(cfg, seed=1))
m = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(sd)
rs = np.random.RandomState(0)


def med(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


for B, L in ((8, 400), (4, 1000), (32, 512)):
    tx = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).cuda()
    ln = torch.full((B,), L, dtype=torch.int64, device="cuda")
    x = torch.randn(B, L, cfg.hidden, device="cuda")
    for attn in (0, 1):
        _lib.internal_set(b"attn_fused", attn)
        t_text = med(lambda: m.duration_pitch_energy_net(None, tx, ln, max_mel_len=2 * L))
        t_dec = med(lambda: m.decoder(x))
        print(f"B={B:2d} L={L:4d} attn_fused={attn}: text side {t_text:.3f} ms, decoder {t_dec:.3f} ms", flush=True)
_lib.internal_set(b"attn_fused", 1)
