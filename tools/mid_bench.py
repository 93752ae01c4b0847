import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
lib = _lib.load()
cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=3))
for B, T in [(4, 512), (8, 512), (12, 512), (16, 512), (20, 512), (24, 512), (32, 256), (8, 1024)]:
    x = torch.randn(B, 1, T, 80, device="cuda"); cond = torch.randn(B, T, 256, device="cuda"); t = torch.full((B,), 1095.5, device="cuda")
    r = {}
    for mode in (0, 2):
        lib.cmtts_set_persistent_denoiser(mode)
        for _ in range(2): model.net(x, t, cond, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): model.net(x, t, cond, None)
        torch.cuda.synchronize(); r[mode] = (time.perf_counter() - t0) / 5 * 1e3
    print(f"B={B} T={T} tiles={B*((T+63)//64)}: per-layer {r[0]:.2f} ms, persistent {r[2]:.2f} ms", flush=True)
lib.cmtts_set_persistent_denoiser(1)
