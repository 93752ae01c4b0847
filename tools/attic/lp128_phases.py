#!/usr/bin/env python3
"""Phase timing of the 16-bit persistent denoiser with 128-frame tiles (denoiser_persist_lp128.hip built with -DLP_STAMP into
tools/bin/libcmtts_lp128stamp.so: tools/abl_build.sh denoiser_persist_lp128 "lp128stamp:-DLP_STAMP"): cycles per phase of the middle layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CMTTS_LIB"] = os.path.join(ROOT, "tools", "bin", "libcmtts_lp128stamp.so")
import numpy as np, torch
sys.path.insert(0, ROOT)
import cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0))
model.set_precision("bf16")
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 512)
x = torch.randn(B, 1, T, 80, device="cuda"); cond = torch.randn(B, T, 256, device="cuda"); t = torch.full((B,), 1095.5, device="cuda")
lib = _lib.load()
_lib.internal_set(b"persist_lp128", 2)
for _ in range(2):
    model.net(x, t, cond, None)
nblk = ((T + 127) // 128) * B
NW = 8
buf = torch.zeros(nblk * NW * 8, dtype=torch.int64, device="cuda")
lib.cmtts_set_debug_stamps(buf.data_ptr())
model.net(x, t, cond, None)
torch.cuda.synchronize()
lib.cmtts_set_debug_stamps(None)
s = buf.cpu().numpy().reshape(nblk, NW, 8).astype(np.float64)
names = ["wait barrier(1)", "conv loop", "gate -> z^T", "flag", "projection loop", "epilogue + state + u^T rows", "granule + halo"]
d = np.diff(s, axis=2)
print(f"B={B} T={T}: cycles per phase of layer {cfg.res_layers // 2} (128-frame tiles), mean / min / max over workgroups and waves")
for i, n in enumerate(names):
    v = d[:, :, i]
    print(f"    {n:28s} mean {v.mean():9.0f}  min {v.min():9.0f}  max {v.max():9.0f}")
tot = s[:, :, 7] - s[:, :, 0]
print("  layer total per wave: mean %.0f max %.0f" % (tot.mean(), tot.max()))
