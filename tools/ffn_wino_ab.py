#!/usr/bin/env python3
"""FFT blocks' k = 9 FFN conv: direct form vs three F(4,3) tap groups (cmtts_internal_set("ffn_wino")): the text side's outputs (encoder -> durations, mel lengths,
conditioning) at several batch shapes — integer outputs must agree, the conditioning by fp32 rounding — and the bits of an utterance alone vs inside a batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
ok = True
for variant in ("LJSpeech", "VCTK"):
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=5, dur_frames=4.0, dur_spread=0.0))
    for B, L in [(1, 7), (2, 33), (3, 85), (32, 85), (9, 100), (4, 128), (2, 300), (40, 97)]:
        g = torch.Generator().manual_seed(B * 1000 + L)
        lens = torch.randint(max(1, L // 2), L + 1, (B,), generator=g); lens[0] = L
        tx = torch.randint(1, cfg.n_symbols, (B, L), generator=g)
        tx[torch.arange(L)[None, :] >= lens[:, None]] = 0
        spk = torch.randn(B, cfg.external_speaker_dim, generator=g).cuda() if cfg.multi_speaker else None
        tx, lens = tx.cuda(), lens.cuda()
        out = {}
        for wn in (0, 1):
            _lib.internal_set(b"ffn_wino", wn)
            o = model.duration_pitch_energy_net(None, tx, lens, spker_embeds=spk)
            out[wn] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()}
            one = model.duration_pitch_energy_net(None, tx[:1], lens[:1], spker_embeds=None if spk is None else spk[:1])
            out[wn]["alone"] = one["cond_ct"].clone(); out[wn]["alone_len"] = one["mel_lens"].clone()
        torch.cuda.synchronize()
        same_int = torch.equal(out[0]["mel_lens"], out[1]["mel_lens"]) and torch.equal(out[0]["d_rounded"], out[1]["d_rounded"]) if "d_rounded" in out[0] else torch.equal(out[0]["mel_lens"], out[1]["mel_lens"])
        T0, T1 = out[0]["cond_ct"].shape[-1], out[1]["cond_ct"].shape[-1]
        d = float((out[0]["cond_ct"] - out[1]["cond_ct"]).abs().max()) if T0 == T1 else float("nan")
        n = max(int(out[1]["alone_len"][0]) - 16, 1)      # (the reference's frame-level predictors do not mask: an utterance's last frames see the batch's padding)
        alone_same = [torch.equal(out[wn]["alone"][0, :, :n], out[wn]["cond_ct"][0, :, :n]) for wn in (0, 1)]
        print(f"{variant} B={B} L={L}: lengths equal {same_int}; max|d cond| {d:.2e} (scale {float(out[0]['cond_ct'].abs().max()):.2f}); first utterance alone == in batch: direct {alone_same[0]}, F(4,3) {alone_same[1]}", flush=True)
        ok &= same_int and d <= 1e-4 and all(alone_same)
_lib.internal_set(b"ffn_wino", 1)
print("ALL OK" if ok else "FAILED")
