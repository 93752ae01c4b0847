import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
lib = _lib.load()
cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=3))
lib.cmtts_set_persistent_denoiser(2)
B, T = 8, 512
streams = [torch.cuda.Stream() for _ in range(4)]
data = [(torch.randn(B, 1, T, 80, device="cuda"), torch.randn(B, T, 256, device="cuda"), torch.full((B,), 1095.5, device="cuda")) for _ in range(4)]
def run(ns, reps=4):
    for i in range(ns):
        with torch.cuda.stream(streams[i]):
            x, c, t = data[i]
            for _ in range(reps): model.net(x, t, c, None)
for ns in (1, 2, 3, 4):
    run(ns); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(ns); torch.cuda.synchronize()
    print(f"{ns} streams x 4 evaluations of B={B} T={T} (64 workgroups each): {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
