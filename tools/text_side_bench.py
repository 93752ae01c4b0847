#!/usr/bin/env python3
"""Wall time of the text side alone (DurationPitchSpeakerNet: encoder + variance adaptor + length regulator + frame-level
pitch chain) and of the conditioner GEMM + one T=1 sampler call, for B = 32 x 85 phonemes and B = 1 x 25, with the
round-2 switches on / off.  Median of 30 calls after warm-up, stream idle between calls."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

lib = _lib.load()
cfg = get_config("LJSpeech")
m = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))
rs = np.random.RandomState(0)


def med(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


for B, L in ((32, 85), (1, 25)):
    tx = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).cuda()
    ln = torch.full((B,), L, dtype=torch.int64, device="cuda")
    noise = torch.randn(2, B, 1, 6 * L, cfg.n_mels, device="cuda")
    for attn in (0, 1):
        for bs in (0, 1):
            _lib.internal_set(b"attn_fused", attn)
            lib.cmtts_set_option(b"branch_streams", bs)
            t_text = med(lambda: m.duration_pitch_energy_net(None, tx, ln, max_mel_len=6 * L))
            o = m.duration_pitch_energy_net(None, tx, ln, max_mel_len=6 * L)
            t_all = med(lambda: host.sample_with_cond(m, m.duration_pitch_energy_net(None, tx, ln, max_mel_len=6 * L)["cond_ct"], None, 1, noise))
            print(f"B={B:2d} L={L:3d} attn_fused={attn} branch_streams={bs}: text side {t_text:.3f} ms, text->mel T=1 {t_all:.3f} ms", flush=True)
_lib.internal_set(b"attn_fused", 1); lib.cmtts_set_option(b"branch_streams", 1)
