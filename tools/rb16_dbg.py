import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict
lib = _lib.load()
lib.cmtts_set_option(b"branch_streams", 0)
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=6))
for prec in ("bf16", "fp16"):
    voc.set_precision(prec)
    mel = (torch.randn(2, 80, 61, generator=torch.Generator().manual_seed(61)) * 1.5 - 4).cuda()
    _lib.internal_set(b"voc_rb16", 0); ref = voc(mel).clone()
    _lib.internal_set(b"voc_rb16", 1); got = voc(mel).clone()
    torch.cuda.synchronize()
    d = (got - ref).abs()[0, 0].cpu().numpy()
    idx = np.nonzero(d)[0]
    print(prec, "n diff", len(idx), "of", d.size, "max", d.max(), "first", idx[:10], "last", idx[-5:] if len(idx) else None)
    if len(idx):
        print("  hist over position/1024:", np.bincount(idx // 1024, minlength=d.size // 1024 + 1))
