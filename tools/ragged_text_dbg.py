#!/usr/bin/env python3
"""Localise a padded-length dependence of the phoneme-level half: one group alone (L) against the same group padded to Lbig with pad_lens = L."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
DEV = "cuda:0"
lib = _lib.load()
for variant in ("LibriTTS", "LJSpeech"):
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, DEV).load_state_dict(synth_cmtts_state_dict(cfg, seed=12, dur_frames=4.0, dur_spread=0.03))
    rs = np.random.RandomState(16)
    for L, Lbig, n in ((32, 128, 5), (96, 128, 3), (32, 64, 2), (33, 40, 2)):
        ln = np.maximum((rs.uniform(0.3, 1.0, size=n) * L).astype(np.int64), 1); ln[0] = L
        tx = rs.randint(1, cfg.n_symbols, size=(n, L)).astype(np.int64); tx[np.arange(L)[None, :] >= ln[:, None]] = 0
        sp = torch.from_numpy(rs.standard_normal(size=(n, cfg.external_speaker_dim)).astype(np.float32)).to(DEV) if cfg.multi_speaker else None
        txb = np.zeros((n, Lbig), np.int64); txb[:, :L] = tx
        def run(texts, Lc, pad):
            B = texts.shape[0]
            f32 = lambda *sh: torch.full(sh, float("nan"), dtype=torch.float32, device=DEV)
            o = dict(log_d=f32(B, Lc), d=f32(B, Lc), e=f32(B, Lc), ei=torch.zeros(B, Lc, dtype=torch.int64, device=DEV),
                     ml=torch.zeros(B, dtype=torch.int64, device=DEV), enc=f32(B, cfg.hidden, Lc), spk=f32(B, cfg.hidden) if cfg.multi_speaker else None)
            nb = lib.cmtts_text_workspace_bytes(model._h, B, Lc)
            tws = torch.empty(nb, dtype=torch.uint8, device=DEV)
            p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
            t_d = torch.from_numpy(texts).to(DEV); l_d = torch.from_numpy(ln).to(DEV)
            pd = None if pad is None else torch.full((B,), pad, dtype=torch.int64, device=DEV)
            _lib.check(lib.cmtts_text_forward_ragged(model._h, p(t_d), p(l_d), p(pd), p(sp), None, B, Lc, 1.0, p(o["log_d"]), p(o["d"]), p(o["ml"]),
                                                     p(o["e"]), p(o["ei"]), p(o["enc"]), p(o["spk"]), p(tws), nb, None))
            torch.cuda.synchronize()
            return o
        a = run(tx, L, None)
        b = run(txb, Lbig, L)
        c = run(txb, Lbig, None)
        def d(x, y, cut=True):
            if x is None: return "-"
            y = y[..., :L] if cut and y.shape[-1] != x.shape[-1] else y
            return f"{float((x - y).abs().max()):.2e}"
        print(f"{variant} L={L}->{Lbig}: padded+pad_lens vs alone: enc {d(a['enc'], b['enc'])} log_d {d(a['log_d'], b['log_d'])} e_pred {d(a['e'], b['e'])} "
              f"spk {d(a['spk'], b['spk'], False)} | padded, no pad_lens: enc {d(a['enc'], c['enc'])} log_d {d(a['log_d'], c['log_d'])} e_pred {d(a['e'], c['e'])}")
