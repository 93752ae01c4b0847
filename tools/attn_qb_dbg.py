#!/usr/bin/env python3
"""attention_qb_kernel vs attention_kernel: where do the encoder outputs differ?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
for variant, B, L in (("LJSpeech", 32, 85), ("LJSpeech", 2, 32), ("VCTK", 1, 25), ("LJSpeech", 3, 97)):
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=23, dur_frames=3.0, dur_spread=0.0))
    rs = np.random.RandomState(1000 + L)
    lens = np.maximum((rs.uniform(0.2, 1.0, size=B) * L).astype(np.int64), 1); lens[0] = L
    texts = rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)
    texts[np.arange(L)[None, :] >= lens[:, None]] = 0
    spk = torch.from_numpy(rs.standard_normal(size=(B, cfg.external_speaker_dim)).astype(np.float32)) if cfg.multi_speaker else None
    outs = []
    for qb in (0, 1, 1, 0):
        _lib.internal_set(b"attn_qb", qb)
        o = model.duration_pitch_energy_net(None, torch.from_numpy(texts), torch.from_numpy(lens), spker_embeds=spk, max_mel_len=3 * L)
        outs.append(o["enc_out"].clone().cpu().numpy())
    _lib.internal_set(b"attn_qb", 1)
    d = np.abs(outs[1] - outs[0])
    print(variant, B, L, "old vs qb max", d.max(), "qb vs qb", np.abs(outs[1] - outs[2]).max(), "old vs old", np.abs(outs[0] - outs[3]).max(),
          "n diff", (d > 0).sum(), "of", d.size)
    if d.max() > 0:
        bi, li, hi = np.unravel_index(np.argmax(d), d.shape)
        print("  worst at utterance", bi, "phoneme", li, "channel", hi, "len", lens[bi], "; per-utterance max:", [float(d[i].max()) for i in range(min(B, 8))])
        print("  per-phoneme max of utterance", bi, [float(x) for x in d[bi].max(1)[:40]])
