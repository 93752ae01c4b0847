#!/usr/bin/env python3
"""cp = stacked conditioner projections through the library (cmtts_internal_cond_projections) against torch, fp32 and 16-bit modes."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

raw = C.CDLL(_lib.LIB_PATH)
raw.cmtts_internal_cond_projections.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
cfg = get_config(os.environ.get("VAR", "LJSpeech"))
sd = synth_cmtts_state_dict(cfg, seed=7)
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(sd)
B, T = int(os.environ.get("CB", 3)), int(os.environ.get("CT", 200))
NL, Cc, H = cfg.res_layers, cfg.res_channels, cfg.hidden
W = torch.cat([torch.from_numpy(np.asarray(sd[f"net.residual_layers.{l}.conditioner_projection.conv.weight"]))[:, :, 0] for l in range(NL)], 0)
bias = torch.cat([torch.from_numpy(np.asarray(sd[f"net.residual_layers.{l}.conditioner_projection.conv.bias"])) for l in range(NL)], 0)
cond = torch.randn(B, H, T, generator=torch.Generator().manual_seed(3))
cd = cond.cuda()
for mode in ("fp32", "bf16", "fp16"):
    model.set_precision(mode)
    cp = torch.full((B, NL * Cc, T), float("nan"), device="cuda")
    h = model._h if hasattr(model, "_h") else model.handle
    rc = raw.cmtts_internal_cond_projections(C.c_void_p(h if isinstance(h, int) else h.value), C.c_void_p(cd.data_ptr()), B, T, C.c_void_p(cp.data_ptr()), None)
    torch.cuda.synchronize()
    tdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[mode]
    ref = torch.einsum("mk,bkt->bmt", W.to(tdt).double(), cond.to(tdt).double()) + bias.double()[None, :, None]
    ref32 = torch.einsum("mk,bkt->bmt", W.double(), cond.double()) + bias.double()[None, :, None]
    d = (cp.cpu().double() - ref).abs(); d32 = (cp.cpu().double() - ref32).abs()
    print(mode, "rc", rc, "vs quantised-operand reference: max", float(d.max()), "rms", float(d.pow(2).mean().sqrt()), "| vs fp32-operand reference: max", float(d32.max()), "scale", float(ref.abs().max()))
