#!/usr/bin/env python3
"""Cycle counters of the generic conv kernel (conv_mfma.hip) for the launches of one text-side pass whose (M, K) match
CMTTS_CONV_DBG_MK="M,K" (default 256,1024 = the FFN linear of the FFT blocks): prologue | sum of MFMA blocks | sum of
(LDS stores + barrier waits) | epilogue, mean cycles per wave of the LAST matching launch."""
import ctypes as C, os, sys
import numpy as np, torch
os.environ.setdefault("CMTTS_CONV_DBG_MK", "256,1024")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

lib = _lib.load()
lib.cmtts_set_option(b"branch_streams", 0)
cfg = get_config("LJSpeech")
m = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))
B, L = int(os.environ.get("PB", 32)), int(os.environ.get("PL", 85))
rs = np.random.RandomState(0)
tx = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).cuda()
ln = torch.full((B,), L, dtype=torch.int64, device="cuda")
for _ in range(3):
    m.duration_pitch_energy_net(None, tx, ln, max_mel_len=6 * L)
torch.cuda.synchronize()
buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
lib.cmtts_set_debug_stamps(C.c_void_p(buf.data_ptr()))
m.duration_pitch_energy_net(None, tx, ln, max_mel_len=6 * L)
torch.cuda.synchronize()
lib.cmtts_set_debug_stamps(None)
s = buf.cpu().numpy().reshape(-1, 8)
s = s[s[:, 0] != 0]
tot = (s[:, 5] - s[:, 0]).astype(np.float64)
print(f"{os.environ['CMTTS_CONV_DBG_MK']} B={B} L={L}: waves {len(s)}; mean cycles per wave {tot.mean():.0f} (max {tot.max():.0f})")
for nme, v in (("prologue", s[:, 1] - s[:, 0]), ("MFMA blocks", s[:, 2]), ("stores + barriers", s[:, 3]),
               ("loop other (load issue)", (s[:, 4] - s[:, 1]) - s[:, 2] - s[:, 3]), ("epilogue", s[:, 5] - s[:, 4])):
    print(f"  {nme:24s} mean {v.mean():9.0f}  max {v.max():9.0f}")
