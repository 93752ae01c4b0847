import sys, os, dataclasses
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from test_gpu_precision import _stress_case
lib = _lib.load()
cfg = dataclasses.replace(get_config("VCTK"), res_layers=4)
sd, x, cond, B, T = _stress_case("beyond_fp16_max", cfg)
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(sd)
spk = torch.from_numpy(np.random.RandomState(5).standard_normal(size=(B, cfg.hidden)).astype(np.float32)).cuda()
cond_ct = torch.from_numpy(np.ascontiguousarray(cond.transpose(0, 2, 1))).cuda()
noise = torch.randn(3, B, 1, T, cfg.n_mels, generator=torch.Generator().manual_seed(2)).cuda()
for forced in (0, 2):
    lib.cmtts_set_persistent_denoiser(forced)
    for dt in ("fp32", "bf16", "fp16", "fp16x3"):
        if dt == "fp16x3" and forced == 0: continue
        model.set_precision(dt)
        mel = host.sample_with_cond(model, cond_ct, spk, 1, noise)
        torch.cuda.synchronize()
        rc = lib.cmtts_poll_error()
        print(forced, dt, "finite", bool(torch.isfinite(mel).all()), "absmax", float(mel.abs().nan_to_num(0).max()), "poll rc", rc, lib.cmtts_last_error() if rc else "")
