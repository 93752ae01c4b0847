#!/usr/bin/env python3
"""Phase timing of the fused residual-block kernel from in-kernel s_memtime stamps (GPU only)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0))
B, T = 32, 512
x = torch.randn(B, 1, T, 80, device="cuda"); cond = torch.randn(B, T, 256, device="cuda"); t = torch.full((B,), 1095.5, device="cuda")
lib = _lib.load()
lib.cmtts_set_persistent_denoiser(0)      # this tool times the per-layer kernel (tools/persist_timing.py: the persistent one)
for _ in range(2):
    model.net(x, t, cond, None)
nblk = (T // 64) * B        # 64-frame tiles (auto-selected at this size)
buf = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
lib.cmtts_set_debug_stamps(buf.data_ptr())
model.net(x, t, cond, None)      # stamps of the LAST layer remain
torch.cuda.synchronize()
lib.cmtts_set_debug_stamps(None)
s = buf.cpu().numpy().reshape(nblk, 8)[:, :7].astype(np.float64)
t0 = s[:, 0].min()
names = ["stage u", "phase B loop", "barrier2", "gate", "phase C loop", "epilogue C"]
d = np.diff(s, axis=1)
print("clock ticks (s_memtime, 100 MHz?) per phase: mean / min / max")
for i, n in enumerate(names):
    print(f"  {n:14s} {d[:, i].mean():10.0f} {d[:, i].min():10.0f} {d[:, i].max():10.0f}")
print("block start spread:", s[:, 0].max() - t0, " block end spread:", s[:, 6].max() - s[:, 6].min(), " total:", s[:, 6].max() - t0)
print("per-block total mean:", (s[:, 6] - s[:, 0]).mean())
