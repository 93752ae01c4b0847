#!/usr/bin/env python3
"""BASELINE.json configs[2] (VCTK B = 64, 85 phonemes -> 512 frames, T = 2, bf16 residual blocks), text -> mel, a few passes: rocprofv3 target."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

cfg = get_config(os.environ.get("VAR", "VCTK"))
m = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))
m.set_precision(os.environ.get("LP", "bf16"))
m.set_option("text16", int(os.environ.get("TEXT16", 0)))
B, L, NS = int(os.environ.get("VB", 64)), int(os.environ.get("VL", 85)), int(os.environ.get("STEPS", 2))
rs = np.random.RandomState(3)
tx = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).cuda()
ln = torch.full((B,), L, dtype=torch.int64, device="cuda")
spk = torch.randn(B, cfg.external_speaker_dim, device="cuda") if cfg.multi_speaker else None
nz = torch.randn(NS + 1, B, 1, L * 6 + 2, cfg.n_mels, device="cuda")
for _ in range(int(os.environ.get("VN", 3))):
    o = m.duration_pitch_energy_net(None, tx, ln, spker_embeds=spk, max_mel_len=L * 6 + 2)
    mel = host.sample_with_cond(m, o["cond_ct"], o["speaker_emb"], NS, nz)
torch.cuda.synchronize()
