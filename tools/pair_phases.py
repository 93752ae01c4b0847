#!/usr/bin/env python3
"""Phase timing of resblock_pair_kernel from in-kernel cycle stamps (PAIR_DBG=5 build: tools/pair_ablation.sh, run with
CMTTS_LIB=cm-tts_amd/libcmtts_hip_dbg5.so).  One pair launch per (C, k, dil) on a [B, C, T] tensor; prints the mean
cycles per phase over all waves and the MFMA-only lower bound."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict

lib = _lib.load()
lib.cmtts_set_option(b"branch_streams", 0)
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
voc(mel); torch.cuda.synchronize()
# stamps of the LAST pair launches overwrite earlier ones per (tile, b, wave) slot: the last launch of a forward is the
# C = 32, k = 11, d = 5 pair; to look at another pair, set VSTOP (count of pair launches to keep) — simplest: run all, read last
n = B * ((T * 256 + 245) // 246 + 2) * 8 * 8
buf = torch.zeros(n, dtype=torch.int64, device="cuda")
lib.cmtts_set_debug_stamps(C.c_void_p(buf.data_ptr()))
voc(mel); torch.cuda.synchronize()
lib.cmtts_set_debug_stamps(None)
s = buf.cpu().numpy().reshape(-1, 8)
s = s[s[:, 0] != 0]
d = np.diff(s, axis=1).astype(np.float64)
names = ["stage x", "barrier", "conv1 loop", "conv1 epilogue", "barrier", "conv2 loop", "conv2 epilogue"]
tot = (s[:, 7] - s[:, 0]).mean()
print(f"waves {len(s)}; mean cycles per wave per tile {tot:.0f}")
for nme, v in zip(names, d.mean(0)):
    print(f"  {nme:16s} {v:9.0f}  {100 * v / tot:5.1f} %")
