import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
lib = _lib.load()
cfg = get_config("LJSpeech")
m = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))
rs = np.random.RandomState(0)
def med(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))
for B, L in ((32, 85), (1, 25)):
    tx = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).cuda()
    ln = torch.full((B,), L, dtype=torch.int64, device="cuda")
    for tx_opt in (0, 1, 4, 5, 7):
        _lib.internal_set(b"text_xres", tx_opt)
        t = med(lambda: m.duration_pitch_energy_net(None, tx, ln, max_mel_len=6 * L))
        print(f"B={B} L={L} text_xres={tx_opt}: text side {t:.3f} ms", flush=True)
