#!/usr/bin/env python3
"""Edge shapes (T = 1, 2, 63, 64, 65, 129; B = 1..3; L = 1) through every denoiser execution mode: persistent == per-layer bitwise, fp32 and bf16 (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
lib = _lib.load()
cfg = get_config("VCTK")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=1, dur_frames=3.0, dur_spread=0.0))
for (B, T) in [(1, 1), (1, 2), (2, 63), (1, 64), (1, 65), (3, 129)]:
    g = torch.Generator().manual_seed(B * 100 + T)
    x = torch.randn(B, 1, T, 80, generator=g); cond = torch.randn(B, T, 256, generator=g); spk = torch.randn(B, 256, generator=g)
    t = torch.full((B,), 1095.5)
    outs = []
    _lib.internal_set(b"persist_wino", 0)          # direct conv form: persistent == per-layer bit for bit
    for mode in (0, 2):
        lib.cmtts_set_persistent_denoiser(mode)
        for prec in ("fp32", "bf16"):
            model.set_precision(prec)
            outs.append(model.net(x, t, cond, spk))
    model.set_precision("fp32")
    _lib.internal_set(b"persist_wino", 1)          # the default (Winograd) form of the fp32 stack: fp32 rounding only
    wino = model.net(x, t, cond, spk)
    lib.cmtts_set_persistent_denoiser(1)
    torch.cuda.synchronize()
    print(B, T, "fp32 equal:", torch.equal(outs[0], outs[2]), "bf16 equal:", torch.equal(outs[1], outs[3]), "finite:", all(bool(torch.isfinite(o).all()) for o in outs),
          "winograd stack max|d|: %.1e" % float((wino - outs[0]).abs().max()))
# text side edge: one phoneme, single utterance
out = model.duration_pitch_energy_net(None, torch.tensor([[5]]), torch.tensor([1]), spker_embeds=torch.randn(1, 512))
print("L=1:", out["mel_lens"].tolist(), tuple(out["cond"].shape), bool(torch.isfinite(out["cond"]).all()))
