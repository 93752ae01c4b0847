#!/usr/bin/env python3
"""Phase timing of the persistent denoiser kernel from in-kernel cycle stamps of the middle layer (GPU only)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0))
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 512)
x = torch.randn(B, 1, T, 80, device="cuda"); cond = torch.randn(B, T, 256, device="cuda"); t = torch.full((B,), 1095.5, device="cuda")
lib = _lib.load()
WINO = int(os.environ.get("WINO", 0))
_lib.internal_set(b"persist_wino", WINO)      # 1: the 8-wave Winograd F(2,3) instances, 2: one wave per SIMD (denoiser_persist4.hip)
_lib.load().cmtts_set_persistent_denoiser(2)
FACT = os.environ.get("FACT") == "1"      # the production path: conditioner factors gathered in-kernel (FACT instances) through the sampler
if FACT:
    from cmtts_amd.weights import synth_cmtts_state_dict as _sd
    model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(_sd(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))
    L = T // 6
    texts = torch.randint(1, cfg.n_symbols, (B, L), device="cuda"); lens = torch.full((B,), L, dtype=torch.int64, device="cuda")
    out = model.duration_pitch_energy_net(None, texts, lens, max_mel_len=T)
    nz = torch.randn(1, B, 1, T, 80, device="cuda")
    run = lambda: host.sample_with_cond(model, out["cond_ct"], None, 1, nz, factors=out.get("cond_factors"))
else:
    run = lambda: model.net(x, t, cond, None)
for _ in range(2):
    run()
nblk = ((T + 63) // 64) * B
NW = 4 if WINO == 2 else 8
buf = torch.zeros(nblk * NW * 8, dtype=torch.int64, device="cuda")
lib.cmtts_set_debug_stamps(buf.data_ptr())
run()
torch.cuda.synchronize()
lib.cmtts_set_debug_stamps(None)
s = buf.cpu().numpy().reshape(nblk, NW, 8).astype(np.float64)
names = ["wait barrier(1)", "phase B loop", "gate", "wait barrier(3)", "phase C loop", "epilogue regs", "publish/u/halo"]
if os.environ.get("PUB") == "1":      # a -DPUB_STAMP build (CMTTS_LIB): the publish phase's own steps in slots 0..5
    names = ["granule stores + index loads issued", "indices + all gathers landed", "u rows formed and written to LDS", "halo wait + halo column", "x' stored to xst", "(slot 5 -> 6: next layer)", "(slot 6 -> 7)"]
d = np.diff(s, axis=2)       # [blk][wave][7]
print(f"B={B} T={T}: cycle-counter ticks per phase of layer {cfg.res_layers // 2} (mean over workgroups)")
for grp, sl in ((("waves 0-3", slice(0, 4)),) if NW == 4 else (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8)))):
    print(" ", grp)
    for i, n in enumerate(names):
        v = d[:, sl, i]
        print(f"    {n:18s} mean {v.mean():9.0f}  min {v.min():9.0f}  max {v.max():9.0f}")
tot = s[:, :, 7] - s[:, :, 0]
print("  layer total per wave: mean %.0f max %.0f" % (tot.mean(), tot.max()))
