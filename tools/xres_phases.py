#!/usr/bin/env python3
"""Phase timing of conv_xres_kernel (the k=9 FFN conv of the FFT blocks, and since round 2 the other K=256 text-side
contractions) from in-kernel cycle stamps: stage X | barrier | K loop | epilogue, mean cycles per wave of the LAST
conv_xres launch of one text-side pass."""
import ctypes as C, os, sys
import numpy as np, torch
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
buf = torch.zeros(B * 64 * 4 * 8, dtype=torch.int64, device="cuda")
lib.cmtts_set_debug_stamps(C.c_void_p(buf.data_ptr()))
m.duration_pitch_energy_net(None, tx, ln, max_mel_len=6 * L)
torch.cuda.synchronize()
lib.cmtts_set_debug_stamps(None)
s = buf.cpu().numpy().reshape(-1, 8)
s = s[(s[:, 0] != 0) & (s[:, 4] != 0)][:, :5]
d = np.diff(s, axis=1).astype(np.float64)
span = s[:, 4].max() - s[:, 0].min()
print(f"B={B} L={L}: waves {len(s)}; kernel span {span} cycles; mean cycles per wave {(s[:, 4] - s[:, 0]).mean():.0f}; "
      f"first start spread {s[:, 0].max() - s[:, 0].min()}")
for nme, v, mx in zip(["stage x", "barrier wait", "K loop", "epilogue"], d.mean(0), d.max(0)):
    print(f"  {nme:14s} mean {v:9.0f}  max {mx:9.0f}")
