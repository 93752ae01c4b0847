"""Python host layer: the reference's call surface over the C ABI (include/cmtts_hip.h).

Mirrors, name for name, the callables the reference's inference path is made of:

    CMTotalTTS / CMDenoiserTTS / DurationPitchSpeakerNet   model/cm_tool/tts_net.py, model/cmtts.py
    KarrasDenoiser.denoise, karras_sample_tts               model/cm_tool/karras_diffusion.py
    Generator (HiFi-GAN), get_vocoder-style loading,
    vocoder_infer                                           hifigan/models.py, utils/model.py
    CMTotalTTSSynthesize.synthesize                         synthesize.py:35-153

PyTorch is plumbing only (device memory, streams, nn.Module container): every operator call goes
to hand-written HIP kernels through ctypes.  There is no eager/PyTorch fallback — a missing
``libcmtts_hip.so`` raises at construction time.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .config import CMTTSConfig, HifiGanConfig


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


def _i64(t, device):
    return t.to(device=device, dtype=torch.int64).contiguous()


def _push_state_dict(lib, setter, handle, sd):
    for name, v in sd.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
        if a.ndim == 0:
            a = a.reshape(1)
        shape = (C.c_int64 * a.ndim)(*a.shape)
        _lib.check(setter(handle, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))


_PG_NOTED = False


def _note_process_group(lib):
    """Once torch.distributed is initialised RCCL's kernels may share the GPU with the persistent denoiser grid, whose
    hand-offs need every workgroup resident: tell the library, which then launches that grid cooperatively (the runtime
    checks residency and fails the launch instead of letting it spin; cmtts_set_option("cooperative_launch", 0|1|2))."""
    global _PG_NOTED
    if not _PG_NOTED and torch.distributed.is_available() and torch.distributed.is_initialized():
        lib.cmtts_set_option(b"process_group", 1)
        _PG_NOTED = True


def _norm_device(device):
    """torch.device with an explicit index ("cuda" -> "cuda:<current>"): tensors report indexed devices, so an
    un-indexed one never compares equal to a buffer's and every workspace lookup would reallocate."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class _Workspace:
    """Caller-owned scratch buffers, cached per size (no allocation on the steady-state path)."""

    def __init__(self):
        self._bufs = {}

    def get(self, key, nbytes, device):
        device = _norm_device(device)
        # one set of buffers per HIP stream: groups running concurrently on different streams must not share scratch
        key = (key, torch.cuda.current_stream(device).cuda_stream)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes or buf.device != device:
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf


def get_mask_from_lengths(lengths, max_len=None):
    """utils/tools.py:275-283 — True = padding."""
    if max_len is None:
        max_len = int(lengths.max().item())
    if lengths.is_cuda and lengths.dtype == torch.int64:     # one small launch instead of arange + compare
        lengths = lengths.contiguous()
        mask = torch.empty(lengths.shape[0], int(max_len), dtype=torch.bool, device=lengths.device)
        with torch.cuda.device(lengths.device):
            _lib.check(_lib.load().cmtts_length_mask(_ptr(lengths), _ptr(mask), lengths.shape[0], int(max_len), _stream()))
        return mask
    ids = torch.arange(0, max_len, device=lengths.device).unsqueeze(0)
    return ids >= lengths.unsqueeze(1)


class CondFactors:
    """The phoneme-level factor of the conditioner projections that cmtts_frame_forward_sub returns beside the conditioning
    (include/cmtts_hip.h): with it the sampler expands cp from (p1, mel2ph, p_idx) instead of running the stacked conditioner GEMM
    over the frames.  Bound to ONE conditioning tensor: `matches` refuses a different or since-modified cond_ct (the sampler then
    takes the dense GEMM on whatever it was given)."""
    __slots__ = ("p1", "p1t", "p1_ld", "L", "mel2ph", "p_idx", "_ptr", "_version", "_shape", "_aux")

    def __init__(self, p1, p1_ld, L, mel2ph, p_idx, cond_ct, p1t=None):
        self.p1, self.p1_ld, self.L, self.mel2ph, self.p_idx = p1, int(p1_ld), int(L), mel2ph, p_idx
        self.p1t = p1t          # [B, res_layers, p1_ld, res_channels]: the same factor with the channels contiguous (cmtts_frame_forward_sub_t), or None
        self._ptr, self._version, self._shape = cond_ct.data_ptr(), cond_ct._version, tuple(cond_ct.shape)
        # mel2ph and p_idx are the very tensors the caller received (out["mel2ph"], out["p_predictions"]["p_idx"]) and p1 is reachable too:
        # an in-place edit of any of them after the duration net would change the mel on the factored path only — their
        # version counters are part of the match
        self._aux = (p1._version, mel2ph._version, p_idx._version, None if p1t is None else p1t._version)

    def matches(self, cond_ct):
        return (cond_ct.data_ptr() == self._ptr and cond_ct._version == self._version and tuple(cond_ct.shape) == self._shape and
                (self.p1._version, self.mel2ph._version, self.p_idx._version, None if self.p1t is None else self.p1t._version) == self._aux)


class CMTotalTTS(torch.nn.Module):
    """Drop-in for model/cm_tool/tts_net.py:40-183 (inference side).

    ``load_state_dict`` takes the reference checkpoint's flat state_dict (synthesize.py:79-83) and
    hands every tensor, under its original key and layout, to ``cmtts_set_tensor``.
    """

    def __init__(self, config: CMTTSConfig, device="cuda:0"):
        super().__init__()
        self.config = config
        self.device = _norm_device(device) if torch.cuda.is_available() else torch.device(device)
        self.lib = _lib.load()
        cs = _lib.CMTTSConfigStruct()
        for name, _ in cs._fields_:
            setattr(cs, name, type(getattr(cs, name))(getattr(config, name)))
        self._h = C.c_void_p()
        _lib.check(self.lib.cmtts_create(C.byref(cs), C.byref(self._h)))
        self._ready = False
        self._ws = _Workspace()
        self._cond_factors = True       # the duration net returns the conditioner factors with its conditioning (fp32 models)
        self.duration_pitch_energy_net = DurationPitchSpeakerNet(self)
        self.net = CMDenoiserTTS(self)
        self.decoder = FastspeechDecoder(self)      # usable when the loaded state dict carries decoder.* (else it raises)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and C is not None:      # C is None during interpreter teardown
            self.lib.cmtts_destroy(h)
            self._h = None

    def load_state_dict(self, state_dict, strict=True):
        """Takes the reference checkpoint's flat state dict.  The tensors are handed to the library under their original
        keys (host side: cmtts_set_tensor copies them); cmtts_finalize re-packs and uploads — at once when a GPU is present,
        otherwise on first use (so that host code which builds the object graph before a device is selected still runs)."""
        _push_state_dict(self.lib, self.lib.cmtts_set_tensor, self._h, state_dict)
        self._pending = True
        if self.device.type == "cuda" and torch.cuda.is_available():
            self._finalize()
        return self

    def _finalize(self):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cmtts_finalize(self._h))
        self._pending = False
        self._ready = True

    def eval(self):
        return self

    def set_precision(self, dtype="fp32"):
        """Operand precision of the denoiser residual blocks: "fp32" (reference), "bf16", "fp16", or "fp16x3" — every
        operand as two fp16 numbers (22 bits), every product as three fp16 MFMAs with fp32 accumulation: fp32-class
        results at a fraction of the fp32 matrix cost (large batches: the persistent stack; others run exact fp32)."""
        mode = {"fp32": 0, "f32": 0, "bf16": 1, "fp16": 2, "f16": 2, "fp16x3": 3}[dtype]
        _lib.check(self.lib.cmtts_set_precision(self._h, mode))
        self._precision_mode = mode
        return self

    def set_option(self, name, value):
        """Per-model numerics option (cmtts_model_set_option): "ffn2_split" 1 (default) | 0 — the FFN linear of the FFT blocks as
        eight K-segment partial GEMMs + one reduction, or as one launch (another fp32 summation order); "text16" 0 (default) | 1 — bf16 / fp16
        models: the in- / out-projections and FFN contractions of the FFT blocks and the variance predictors' convs with 16-bit operands as well (the integer stages — durations, pitch buckets, lengths —
        then depend on the precision mode); "winograd" 1 (default) | 2 | 0 — fp32 models, large batches: the gated k = 3 conv of the persistent
        denoiser stack as a Winograd convolution — 1: F(4,3) (half of the conv's MFMAs; fp32 rounding differences ~8e-6 on a mel; since round 5 / ABI revision 5),
        2: F(2,3) (2/3 of them; ~4e-6) — or (0) in the direct form (bit for bit the per-layer kernels of small batches).  Returns the previous value."""
        prev = self.lib.cmtts_model_set_option(self._h, name.encode() if isinstance(name, str) else name, int(value))
        if prev < 0:
            _lib.check(prev)
        return prev

    def _require(self):
        _note_process_group(self.lib)
        if not self._ready and getattr(self, "_pending", False):
            if not torch.cuda.is_available():
                raise RuntimeError("CMTotalTTS: no GPU — cmtts_amd has no CPU fallback (the weights are loaded, the kernels cannot run)")
            self.device = _norm_device(self.device if self.device.type == "cuda" else "cuda")
            self._finalize()
        if not self._ready:
            raise RuntimeError("CMTotalTTS: load_state_dict() first")

    def to(self, *args, **kwargs):
        """nn.Module.to: the weights live in the library's packed device buffers on the device given at construction;
        moving to that device (what synthesize.py:84 does) is a no-op, anything else is refused rather than ignored."""
        dev = kwargs.get("device", args[0] if args and isinstance(args[0], (str, torch.device, int)) else None)
        if dev is not None and self._ready and torch.cuda.is_available() and _norm_device(dev) != self.device:
            raise RuntimeError(f"CMTotalTTS was built on {self.device}; build another instance for {dev}")
        return self

    def get_segmentation_model(self):
        """tts_net.py:66-73 -> (duration_pitch_energy_net, denoise_fun)."""
        return self.duration_pitch_energy_net, self.net.forward

    def forward(self, x, timesteps, speakers=None, texts=None, src_lens=None, pitch=None, f0=None, uv=None, cwt_spec=None,
                f0_mean=None, f0_std=None, mel_lens=None, e_targets=None, d_targets=None, mel2phs=None, spker_embeds=None,
                p_control=1.0, e_control=1.0, d_control=1.0, **kwargs):
        """tts_net.py:75-183: re-runs the duration net with max_mel_len = x.size(2) (mels=x), then the denoiser.  The
        pitch target is assembled exactly like the reference does (:121-131): only when `pitch` is given, from
        pitch / f0 / uv / cwt_spec / f0_mean / f0_std.  The loss bookkeeping of :166-181 is training-only (not built)."""
        p_targets = None if pitch is None else {"pitch": pitch, "f0": f0, "uv": uv, "cwt_spec": cwt_spec,
                                                "f0_mean": f0_mean, "f0_std": f0_std}
        out = self.duration_pitch_energy_net(speakers=speakers, texts=texts, src_lens=src_lens, mels=x, mel_lens=mel_lens,
                                             p_targets=p_targets, e_targets=e_targets, d_targets=d_targets, mel2phs=mel2phs,
                                             spker_embeds=spker_embeds, p_control=p_control, e_control=e_control,
                                             d_control=d_control)
        return self.net(x, timesteps, out["cond"], out["speaker_emb"], out["mel_masks"])


class DurationPitchSpeakerNet(torch.nn.Module):
    """model/cmtts.py:10-122: the inference branch (targets None, controls 1) and the teacher-forced /
    controlled branches of VarianceAdaptor.forward (model/modules.py:331-412)."""

    def __init__(self, owner: CMTotalTTS):
        super().__init__()
        self.__dict__["_owner"] = owner

    def forward(self, speakers=None, texts=None, src_lens=None, mels=None, mel_lens=None, p_targets=None,
                e_targets=None, d_targets=None, mel2phs=None, spker_embeds=None,
                p_control=1.0, e_control=1.0, d_control=1.0, max_mel_len=None, **kwargs):
        """p_targets = {"cwt_spec" [B,T,10], "f0_mean" [B], "f0_std" [B], "uv" bool [B,T]}, e_targets [B,L],
        d_targets [B,L] as in the reference.  `mel2phs` is accepted and ignored: mel2ph is recomputed from
        d_targets (dur_to_mel2ph), which is what the reference's dataset stores."""
        o = self._owner
        o._require()
        cfg, lib, dev = o.config, o.lib, o.device
        vc, keep = None, []
        if p_control != 1.0 or e_control != 1.0 or p_targets is not None or e_targets is not None or d_targets is not None:
            vc = _lib.VarianceControlsStruct(p_control=float(p_control), e_control=float(e_control))
            if d_targets is not None:
                keep.append(_f32(d_targets, dev)); vc.d_target = _ptr(keep[-1])
            if e_targets is not None:
                keep.append(_f32(e_targets, dev)); vc.e_target = _ptr(keep[-1])
            if p_targets is not None:
                for k in ("cwt_spec", "f0_mean", "f0_std"):
                    keep.append(_f32(p_targets[k], dev)); setattr(vc, k, _ptr(keep[-1]))
                keep.append(p_targets["uv"].to(device=dev, dtype=torch.uint8).contiguous()); vc.uv = _ptr(keep[-1])
                if mels is None and max_mel_len is None:
                    max_mel_len = int(p_targets["cwt_spec"].shape[1])
        texts = _i64(texts, dev)
        src_lens = _i64(src_lens, dev)
        B, L = texts.shape
        table = cfg.multi_speaker and cfg.n_speaker > 0          # preprocess.yaml speaker_embedder "none": nn.Embedding(speakers)
        spk_in = _f32(spker_embeds, dev) if (cfg.multi_speaker and not table and spker_embeds is not None) else None
        if cfg.multi_speaker and not table and spk_in is None:
            raise AssertionError("Speaker embedding should not be None")
        spk_ids = None
        if table:
            if speakers is None:
                raise AssertionError("speakers (ids into the speaker_emb table) should not be None")
            # what nn.Embedding raises (model/cmtts.py:78), also for ids that already live on the device (the reference's
            # to_device has moved them): one tiny min/max read-back — the text path synchronises for mel_len anyway, and a
            # clamped id would silently synthesise another speaker
            if speakers.numel() and (int(speakers.min()) < 0 or int(speakers.max()) >= cfg.n_speaker):
                raise IndexError("index out of range in self")
            spk_ids = _i64(speakers, dev)
        H = cfg.hidden
        with torch.cuda.device(dev):
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            log_d, d_rounded, e_pred = f(B, L), f(B, L), f(B, L)
            mel_len = torch.empty(B, dtype=torch.int64, device=dev)
            e_idx = torch.empty(B, L, dtype=torch.int64, device=dev)
            enc_ct = f(B, H, L)
            spk = f(B, H) if cfg.multi_speaker else None
            nb = lib.cmtts_text_workspace_bytes(o._h, B, L)
            tws = o._ws.get("text", nb, dev)
            if vc is not None:
                _lib.check(lib.cmtts_set_variance_controls(o._h, C.byref(vc)))
            try:
                _lib.check(lib.cmtts_text_forward(o._h, _ptr(texts), _ptr(src_lens), _ptr(spk_in), _ptr(spk_ids), B, L, float(d_control),
                                                  _ptr(log_d), _ptr(d_rounded), _ptr(mel_len), _ptr(e_pred), _ptr(e_idx),
                                                  _ptr(enc_ct), _ptr(spk), _ptr(tws), nb, _stream()))
                if mels is not None:
                    T = int(mels.size(2))                       # model/cmtts.py:61-62
                elif max_mel_len is not None:
                    T = int(max_mel_len)
                else:
                    T = int(mel_len.max().item())               # the one host read-back (pad() batch max)
                O = cfg.cwt_out
                cond_ct = f(B, H, T)
                mel2ph = torch.empty(B, T, dtype=torch.int64, device=dev)
                cwt = f(B, T, O)
                f0 = f(B, T)
                p_idx = torch.empty(B, T, dtype=torch.int64, device=dev)
                stats = f(B, 2)
                nf = lib.cmtts_frame_workspace_bytes(o._h, B, T)
                fws = o._ws.get("frame", nf, dev)
                # fp32 models: the phoneme-level factor of the conditioner projections rides along (CondFactors)
                p1_ld = (L + 3) // 4 * 4
                p1 = f(B, cfg.res_layers * cfg.res_channels, p1_ld) if getattr(o, "_precision_mode", 0) == 0 and o._cond_factors else None
                p1t = None if p1 is None else f(B, cfg.res_layers, p1_ld, cfg.res_channels)      # channels contiguous: what the persistent stack gathers from
                _lib.check(lib.cmtts_frame_forward_sub_t(o._h, _ptr(tws), B, L, 0, B, T, _ptr(cond_ct), _ptr(mel2ph), _ptr(cwt),
                                                         _ptr(f0), _ptr(p_idx), _ptr(stats), _ptr(p1), _ptr(p1t), _ptr(fws), nf, _stream()))
            finally:
                if vc is not None:
                    lib.cmtts_set_variance_controls(o._h, None)              # back to the inference defaults
                    torch.cuda.current_stream(dev).synchronize()             # targets must outlive the kernels
        mel_masks = get_mask_from_lengths(mel_len, T)
        factors = None if p1 is None else CondFactors(p1, p1_ld, L, mel2ph, p_idx, cond_ct, p1t)
        cond_ct._cmtts_factors = factors       # rides along with THIS tensor object: sample_with_cond(cond_ct, ...) finds it (and re-checks it)
        return {
            "cond": cond_ct.transpose(1, 2),               # [B,T,H] view of the channel-major buffer
            "cond_ct": cond_ct,
            "cond_factors": factors,
            "p_targets": p_targets,
            "p_predictions": {"pitch_pred": None, "f0_denorm": f0, "cwt": cwt,
                              "f0_mean": stats[:, 0], "f0_std": stats[:, 1], "p_idx": p_idx},
            "e_predictions": e_pred,
            "e_idx": e_idx,
            "log_d_predictions": log_d,
            "d_rounded": d_rounded,
            "mel_lens": mel_len,
            "mel_masks": mel_masks,
            "mel2ph": mel2ph,
            "src_masks": get_mask_from_lengths(src_lens, L),
            "speaker_emb": spk,
            "src_lens": src_lens,
            "enc_out": enc_ct.transpose(1, 2),
        }


def _as_cond_ct(conditioner, dev):
    """[B,T,H] reference layout -> channel-major [B,H,T]; free when it is a view of a _ct buffer."""
    c = conditioner.to(device=dev, dtype=torch.float32).transpose(1, 2)
    return c if c.is_contiguous() else c.contiguous()


class FastspeechDecoder(torch.nn.Module):
    """model/modules.py:154-165 (FFTBlocks.forward :80-105 with the learnable pos_embed_alpha).  The reference defines
    this module and never instantiates it; it is served here from the same FFT-block kernels as
    the text encoder, over the frame axis, for checkpoints whose state dict carries `decoder.*`
    (cmtts_amd.weights.synth_decoder_state_dict lists the keys).  forward(x [B,T,256], padding_mask [B,T] | None)."""

    def __init__(self, owner: CMTotalTTS):
        super().__init__()
        self.__dict__["_owner"] = owner      # not a registered child: the owner holds this module (no module cycle)

    def forward(self, x, padding_mask=None, attn_mask=None, return_hiddens=False):
        if attn_mask is not None or return_hiddens:
            raise NotImplementedError("FastspeechDecoder: attn_mask / return_hiddens are not used on the inference path")
        o = self._owner
        o._require()
        dev, lib = o.device, o.lib
        x = _f32(x, dev)
        B, T, H = x.shape
        if padding_mask is None:
            padding_mask = x.abs().sum(-1).eq(0)            # modules.py:86
        padding_mask = padding_mask.to(device=dev, dtype=torch.bool)
        lens = (~padding_mask).sum(1).to(torch.int64)
        if not torch.equal(padding_mask, get_mask_from_lengths(lens, T)):
            raise NotImplementedError("FastspeechDecoder: the padding mask must mark a suffix of every row (a length mask)")
        x_ct = transpose_last2(x)
        with torch.cuda.device(dev):
            out_ct = torch.empty(B, H, T, dtype=torch.float32, device=dev)
            nb = lib.cmtts_decoder_workspace_bytes(o._h, B, T)
            ws = o._ws.get("dec", nb, dev)
            _lib.check(lib.cmtts_decoder_forward(o._h, _ptr(x_ct), _ptr(lens), B, T, _ptr(out_ct), _ptr(ws), nb, _stream()))
        return transpose_last2(out_ct)


class CMDenoiserTTS(torch.nn.Module):
    """model/cm_tool/tts_net.py:11-37: forward(x, timesteps, conditioner, speaker_emb, mask)."""

    def __init__(self, owner: CMTotalTTS):
        super().__init__()
        self.__dict__["_owner"] = owner

    def forward(self, x, timesteps, conditioner=None, speaker_emb=None, mask=None):
        o = self._owner
        o._require()
        dev, lib = o.device, o.lib
        x = _f32(x, dev)
        B, one, T, M = x.shape
        assert one == 1 and M == o.config.n_mels
        cond_ct = _as_cond_ct(conditioner, dev)
        t = _f32(timesteps, dev)
        spk = _f32(speaker_emb, dev) if speaker_emb is not None else None
        with torch.cuda.device(dev):
            out = torch.empty_like(x)
            nb = lib.cmtts_denoiser_workspace_bytes(o._h, B, T)
            ws = o._ws.get("den", nb, dev)
            _lib.check(lib.cmtts_denoiser_forward(o._h, _ptr(x), _ptr(t), _ptr(cond_ct), _ptr(spk), B, T, _ptr(out),
                                                  _ptr(ws), nb, _stream()))
        return out


class KarrasDenoiser:
    """model/cm_tool/karras_diffusion.py:35-102,392-407 (inference members only)."""

    def __init__(self, sigma_data=0.5, sigma_max=80.0, sigma_min=0.002, rho=7.0, weight_schedule="karras",
                 distillation=False, loss_norm="lpips", **kw):
        """Same defaults as the reference (karras_diffusion.py:36-45: distillation=False); the synthesize path builds
        it with distillation=True (synthesize.py:59-78 via create_model_and_diffusion_tts)."""
        self.sigma_data, self.sigma_max, self.sigma_min, self.rho = sigma_data, sigma_max, sigma_min, rho
        self.weight_schedule, self.distillation, self.loss_norm = weight_schedule, distillation, loss_norm

    def get_scalings(self, sigma):
        """karras_diffusion.py:81-85 (distillation=False)."""
        c_skip = self.sigma_data ** 2 / (sigma ** 2 + self.sigma_data ** 2)
        c_out = sigma * self.sigma_data / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        c_in = 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        return c_skip, c_out, c_in

    def get_scalings_for_boundary_condition(self, sigma):
        c_skip = self.sigma_data ** 2 / ((sigma - self.sigma_min) ** 2 + self.sigma_data ** 2)
        c_out = (sigma - self.sigma_min) * self.sigma_data / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        c_in = 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        return c_skip, c_out, c_in

    def denoise(self, model, x_t, sigmas, **model_kwargs):
        """Generic (unfused) form: host-side scalings around any model callable.  The fused
        path used by karras_sample_tts is cmtts_sample."""
        scal = self.get_scalings_for_boundary_condition if self.distillation else self.get_scalings
        c_skip, c_out, c_in = [v[(...,) + (None,) * (x_t.ndim - v.ndim)] for v in scal(sigmas)]
        rescaled_t = 1000 * 0.25 * torch.log(sigmas + 1e-44)
        model_output = model(c_in * x_t, rescaled_t, **model_kwargs)
        return model_output, c_out * model_output + c_skip * x_t


class DummyGenerator:
    """model/cm_tool/random_util.py:17-25."""

    def randn(self, *args, **kwargs):
        return torch.randn(*args, **kwargs)

    def randn_like(self, *args, **kwargs):
        return torch.randn_like(*args, **kwargs)


def check_async_error():
    """Raise if a persistent denoiser launch that has completed reported a neighbour-wait timeout (cmtts_poll_error)."""
    _lib.check(_lib.load().cmtts_poll_error())


def synchronize(device=None):
    """torch.cuda.synchronize + cmtts_poll_error: a failure of any launch completed so far is raised HERE, by the call
    that makes the results host-visible, not by whichever call comes next."""
    torch.cuda.synchronize(device)
    check_async_error()


def sample_with_cond(model: CMTotalTTS, cond_ct, speaker_emb, n_steps, noise, factors=None):
    """T-step consistency sampling on precomputed conditioning (cmtts_sample).
    noise: fp32 [n_noise,B,1,T,80] on device.  factors: the duration net's out["cond_factors"] for THIS cond_ct (default: the ones the
    duration net attached to the tensor object it returned; the conditioner projections are then expanded from them,
    cmtts_sample_factored — an unmodified, same-storage cond_ct only, else the dense GEMM).  Returns mel [B,T,80]."""
    model._require()
    lib, dev, cfg = model.lib, model.device, model.config
    B, H, T = cond_ct.shape
    n_noise = 1 if n_steps == 1 else n_steps + 1
    assert noise.shape[0] >= n_noise and tuple(noise.shape[1:]) == (B, 1, T, cfg.n_mels)
    sig = (C.c_float * n_steps)()
    std = (C.c_float * n_steps)()
    _lib.check(lib.cmtts_schedule(model._h, n_steps, sig, std))
    with torch.cuda.device(dev):
        mel = torch.empty(B, T, cfg.n_mels, dtype=torch.float32, device=dev)
        nb = lib.cmtts_denoiser_workspace_bytes(model._h, B, T)
        ws = model._ws.get("den", nb, dev)
        if factors is None:
            factors = getattr(cond_ct, "_cmtts_factors", None)
        if factors is not None and factors.matches(cond_ct):
            _lib.check(lib.cmtts_sample_factored_t(model._h, _ptr(noise), _ptr(cond_ct), _ptr(speaker_emb), B, T, n_steps, sig, std,
                                                   _ptr(mel), _ptr(ws), nb, _stream(), _ptr(factors.p1), _ptr(factors.p1t), factors.p1_ld, factors.L,
                                                   _ptr(factors.mel2ph), _ptr(factors.p_idx)))
        else:
            _lib.check(lib.cmtts_sample(model._h, _ptr(noise), _ptr(cond_ct), _ptr(speaker_emb), B, T, n_steps, sig, std,
                                        _ptr(mel), _ptr(ws), nb, _stream()))
    return mel


def sample_ragged(model: CMTotalTTS, groups, n_steps, tail_frames=0):
    """cmtts_sample_ragged: T-step consistency sampling of a RAGGED shard — every group a padded (B, T) batch with its own
    conditioning, results those of sample_with_cond on that batch — with the residual layers of ALL groups in one persistent
    launch per evaluation (buckets too small to fill the chip fill it together).
    groups: iterable of (cond_ct [B,H,T], speaker_emb [B,H] | None, noise [n_noise,B,1,T,80], active_frames | None[, CondFactors | None]);
    active_frames = host sequence of B ints (mel_len): the utterance is then only computed as far as those frames (+ tail_frames
    + the sampler's receptive field) need — they come out bit-identical to the untrimmed run in the direct and F(2,3) forms of the
    stack and within fp32 rounding of it (<= 1.5e-5: include/cmtts_hip.h, tests/conftest.py WINO_TRIM_TOL; ~2e-6 typical) in the default F(4,3) form, whose frame quads round every output from all
    six inputs of the quad; frames beyond the computed range are zeros.
    Returns the list of mels [B,T,80]."""
    model._require()
    lib, dev, cfg = model.lib, model.device, model.config
    n_noise = 1 if n_steps == 1 else n_steps + 1
    sig = (C.c_float * n_steps)()
    std = (C.c_float * n_steps)()
    _lib.check(lib.cmtts_schedule(model._h, n_steps, sig, std))
    groups = list(groups)
    arr = (_lib.SampleGroupStruct * len(groups))()
    mels, keep = [], []
    with torch.cuda.device(dev):
        for gi, grp in enumerate(groups):
            cond_ct, spk, noise, active = grp[:4]
            factors = grp[4] if len(grp) > 4 else None
            B, H, T = cond_ct.shape
            assert noise.shape[0] >= n_noise and tuple(noise.shape[1:]) == (B, 1, T, cfg.n_mels)
            mel = torch.empty(B, T, cfg.n_mels, dtype=torch.float32, device=dev)
            nb = lib.cmtts_denoiser_workspace_bytes(model._h, B, T)
            ws = model._ws.get(("den_ragged", gi), nb, dev)
            g = arr[gi]
            g.noise, g.cond_ct, g.speaker_emb = noise.data_ptr(), cond_ct.data_ptr(), (spk.data_ptr() if spk is not None else None)
            g.B, g.T, g.mel, g.ws, g.ws_bytes = B, T, mel.data_ptr(), ws.data_ptr(), nb
            if active is not None:
                act = (C.c_int64 * B)(*[int(v) for v in active])
                keep.append(act)
                g.active_frames = C.cast(act, C.c_void_p).value
            if factors is not None and factors.matches(cond_ct):
                g.cond_p1, g.p1_ld, g.L = factors.p1.data_ptr(), factors.p1_ld, factors.L
                g.mel2ph, g.p_idx = factors.mel2ph.data_ptr(), factors.p_idx.data_ptr()
                keep.append(factors)
            mels.append(mel)
            keep += [cond_ct, spk, noise, ws]
        _lib.check(lib.cmtts_sample_ragged(model._h, arr, len(groups), n_steps, sig, std, int(tail_frames), _stream()))
    return mels


def karras_sample_tts(diffusion, model, shape, steps=2, clip_denoised=False, progress=False, callback=None,
                      model_kwargs=None, device=None, sigma_min=0.002, sigma_max=80, rho=7.0, sampler="onestep",
                      generator=None, ts=None, **unused):
    """karras_diffusion.py:480-577.  "onestep" and "multistep" with steps=2, ts=(0,)*T+(1,) (what
    synthesize.py selects) run fused in cmtts_sample; "heun", "dpm", "euler", "ancestral", "our_multistep", any
    other multistep `ts` schedule, and everything under a diffusion with distillation=False (get_scalings instead of
    the boundary-condition scalings) run the reference's loops host-side around the denoiser kernels (s_churn,
    s_tmin, s_tmax, s_noise, T as keywords).  The duration net runs once (bit-identical to the reference's
    per-evaluation re-run) with max_mel_len = shape[2]."""
    if generator is None:
        generator = DummyGenerator()
    B, one, T, M = shape
    ode = sampler in ("heun", "dpm", "euler", "ancestral", "our_multistep", "progdist")
    distilled = getattr(diffusion, "distillation", False)
    # cmtts_schedule / cmtts_sample take sigma_min, sigma_max, sigma_data and rho from the model's config; a caller
    # whose arguments or diffusion object say otherwise gets the reference's host-side loop, which honours them
    cfg = model.config
    same = lambda a, b: abs(float(a) - float(b)) <= 1e-6 * max(abs(float(a)), abs(float(b)), 1e-30)
    if not (same(sigma_min, cfg.sigma_min) and same(sigma_max, cfg.sigma_max) and
            same(getattr(diffusion, "sigma_min", cfg.sigma_min), cfg.sigma_min) and
            same(getattr(diffusion, "sigma_max", cfg.sigma_max), cfg.sigma_max) and
            same(getattr(diffusion, "sigma_data", cfg.sigma_data), cfg.sigma_data) and
            same(getattr(diffusion, "rho", cfg.rho), cfg.rho)):
        distilled_fusable = False
    else:
        distilled_fusable = distilled
    if sampler == "onestep":
        n_steps = 1
        ode = not distilled_fusable    # cmtts_sample fuses the boundary-condition scalings (and the config's sigmas) only
    elif sampler == "multistep":
        fused = distilled_fusable and steps == 2 and ts is not None and tuple(ts[:-1]) == (0,) * (len(ts) - 1) and ts[-1] == 1
        n_steps = len(ts) - 1 if fused else 0
        ode = not fused                # any other `ts` schedule: the reference's loop, host-side
    elif not ode:
        raise KeyError(sampler)          # the reference's dispatch table has no other entry (karras_diffusion.py:536-545)
    dev = model.device
    kw = dict(model_kwargs or {})
    out = model.duration_pitch_energy_net(kw.get("speakers"), kw["texts"], kw["src_lens"],
                                          spker_embeds=kw.get("spker_embeds"), max_mel_len=T)
    if ode:
        # the reference's other loops (karras_diffusion.py:538-577), host-side around the denoiser kernels
        sigmas = get_sigmas_karras(steps + 1 if sampler == "progdist" else steps, sigma_min, sigma_max, rho)     # karras_diffusion.py:529-532
        x_T = _f32(generator.randn(*shape, device=dev), dev) * sigma_max
        dist = make_distiller(diffusion, model, out_cond(out), out["speaker_emb"])
        fn = {"heun": sample_heun, "dpm": sample_dpm, "euler": sample_euler, "ancestral": sample_euler_ancestral,
              "our_multistep": our_multistep, "onestep": sample_onestep, "multistep": stochastic_iterative_sampler,
              "progdist": sample_progdist}[sampler]
        args = {}
        if sampler in ("heun", "dpm"):
            args = {k: unused[k] for k in ("s_churn", "s_tmin", "s_tmax", "s_noise") if k in unused}
        elif sampler == "our_multistep":
            args = {"T": unused.get("T", 4)}
        elif sampler == "multistep":
            args = dict(ts=ts, t_min=sigma_min, t_max=sigma_max, rho=diffusion.rho, steps=steps)
        return fn(dist, x_T, sigmas, generator, **args)[:, 0]
    draws = [generator.randn(*shape, device=dev)]
    for _ in range(n_steps if n_steps > 1 else 0):
        draws.append(generator.randn_like(draws[0]))
    noise = torch.stack([_f32(d, dev) for d in draws], 0)
    return sample_with_cond(model, out["cond_ct"], out["speaker_emb"], n_steps, noise, factors=out.get("cond_factors"))


def out_cond(out):
    """[B,T,H] view of the duration net's channel-major conditioning (what CMDenoiserTTS.forward takes)."""
    return out["cond_ct"].transpose(1, 2)


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0, device="cpu"):
    """karras_diffusion.py:580-586 (fp32, trailing 0).  Kept on the host: the loops branch on its values."""
    ramp = torch.linspace(0, 1, n)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return torch.cat([(hi + ramp * (lo - hi)) ** rho, torch.zeros(1)])


def to_d(x, sigma, denoised):
    """karras_diffusion.py:589-591."""
    return (x - denoised) / float(sigma)


def get_ancestral_step(sigma_from, sigma_to):
    """karras_diffusion.py:594-601."""
    sigma_up = (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def sample_euler(denoiser, x, sigmas, generator=None, **kw):
    """karras_diffusion.py:743-771."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        d = to_d(x, sigmas[i], denoiser(x, float(sigmas[i]) * s_in))
        x = x + d * float(sigmas[i + 1] - sigmas[i])
    return x


def sample_euler_ancestral(denoiser, x, sigmas, generator, **kw):
    """karras_diffusion.py:605-632."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = denoiser(x, float(sigmas[i]) * s_in)
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1])
        d = to_d(x, sigmas[i], denoised)
        x = x + d * float(sigma_down - sigmas[i])
        x = x + generator.randn_like(x) * float(sigma_up)
    return x


def _churn(x, sigmas, i, generator, s_churn, s_tmin, s_tmax, s_noise):
    gamma = min(s_churn / (len(sigmas) - 1), 2 ** 0.5 - 1) if s_tmin <= sigmas[i] <= s_tmax else 0.0
    eps = generator.randn_like(x) * s_noise          # drawn every iteration, like the reference
    sigma_hat = sigmas[i] * (gamma + 1)
    if gamma > 0:
        x = x + eps * float((sigma_hat ** 2 - sigmas[i] ** 2) ** 0.5)
    return x, sigma_hat


def sample_progdist(denoiser, x, sigmas, generator=None, **kw):
    """karras_diffusion.py:856-888: Euler steps over the schedule WITHOUT its trailing zero (the progressive-distillation teacher's
    sampler; `steps + 1` sigmas are requested for it, :529-530)."""
    s_in = x.new_ones([x.shape[0]])
    sigmas = sigmas[:-1]
    for i in range(len(sigmas) - 1):
        d = to_d(x, sigmas[i], denoiser(x, float(sigmas[i]) * s_in))
        x = x + d * float(sigmas[i + 1] - sigmas[i])
    return x


def sample_heun(denoiser, x, sigmas, generator, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, **kw):
    """karras_diffusion.py:693-739 (Algorithm 2 of Karras et al. 2022)."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        x, sigma_hat = _churn(x, sigmas, i, generator, s_churn, s_tmin, s_tmax, s_noise)
        d = to_d(x, sigma_hat, denoiser(x, float(sigma_hat) * s_in))
        dt = float(sigmas[i + 1] - sigma_hat)
        if sigmas[i + 1] == 0:
            x = x + d * dt
        else:
            x_2 = x + d * dt
            d_2 = to_d(x_2, sigmas[i + 1], denoiser(x_2, float(sigmas[i + 1]) * s_in))
            x = x + (d + d_2) / 2 * dt
    return x


def sample_dpm(denoiser, x, sigmas, generator, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, **kw):
    """karras_diffusion.py:775-820 (midpoint on a rho=3 schedule)."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        x, sigma_hat = _churn(x, sigmas, i, generator, s_churn, s_tmin, s_tmax, s_noise)
        d = to_d(x, sigma_hat, denoiser(x, float(sigma_hat) * s_in))
        sigma_mid = ((sigma_hat ** (1 / 3) + sigmas[i + 1] ** (1 / 3)) / 2) ** 3
        x_2 = x + d * float(sigma_mid - sigma_hat)
        d_2 = to_d(x_2, sigma_mid, denoiser(x_2, float(sigma_mid) * s_in))
        x = x + d_2 * float(sigmas[i + 1] - sigma_hat)
    return x


def sample_onestep(distiller, x, sigmas, generator=None, **kw):
    """karras_diffusion.py:801-811 on the generic (unfused) denoise callable."""
    return distiller(x, sigmas[0] * x.new_ones([x.shape[0]]))


def our_multistep(distiller, x, sigmas, generator=None, T=4, **kw):
    """karras_diffusion.py:814-826: T evaluations at sigma_max without re-noising."""
    s_in = x.new_ones([x.shape[0]])
    for _ in range(T):
        x = distiller(x, sigmas[0] * s_in)
    return x


def stochastic_iterative_sampler(distiller, x, sigmas, generator, ts, t_min=0.002, t_max=80.0, rho=7.0, steps=40, **kw):
    """karras_diffusion.py:830-854, any `ts` schedule, on the generic denoise callable (SURVEY.md §8f item 3:
    host-side loops around the same denoiser kernels; the fused cmtts_sample covers synthesize.py's cases)."""
    t_max_rho, t_min_rho = t_max ** (1 / rho), t_min ** (1 / rho)
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(ts) - 1):
        t = (t_max_rho + ts[i] / (steps - 1) * (t_min_rho - t_max_rho)) ** rho
        x0 = distiller(x, t * s_in)
        next_t = (t_max_rho + ts[i + 1] / (steps - 1) * (t_min_rho - t_max_rho)) ** rho
        next_t = float(np.clip(next_t, t_min, t_max))
        x = x0 + generator.randn_like(x) * np.sqrt(next_t ** 2 - t_min ** 2) * 0.85
    return x


def make_distiller(diffusion: "KarrasDenoiser", model: CMTotalTTS, cond, speaker_emb):
    """denoiser(x_t, sigma) closure of karras_sample_tts (karras_diffusion.py:561-566) over precomputed
    conditioning: KarrasDenoiser.denoise around CMDenoiserTTS.forward."""
    def distiller(x_t, sigma):
        return diffusion.denoise(lambda xx, tt: model.net(xx, tt, cond, speaker_emb), x_t, sigma)[1]
    return distiller


class Generator(torch.nn.Module):
    """hifigan/models.py:112-174 — HiFi-GAN V1 generator; forward(x [B,80,T]) -> [B,1,256*T]."""

    def __init__(self, h: HifiGanConfig = HifiGanConfig(), device="cuda:0"):
        super().__init__()
        self.h = h
        self.device = _norm_device(device) if torch.cuda.is_available() else torch.device(device)
        self.lib = _lib.load()
        self._h = C.c_void_p()
        _lib.check(self.lib.cmtts_vocoder_create(C.byref(self._h)))
        self._ready = False
        self._ws = _Workspace()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and C is not None:
            self.lib.cmtts_vocoder_destroy(h)
            self._h = None

    def load_state_dict(self, state_dict, strict=True):
        """Accepts plain weights or weight_g/weight_v pairs (folded like remove_weight_norm)."""
        from .weights import fold_weight_norm
        sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in state_dict.items()}
        if any(k.endswith("weight_g") for k in sd):
            sd = fold_weight_norm(sd)
        _push_state_dict(self.lib, self.lib.cmtts_vocoder_set_tensor, self._h, sd)
        self._pending = True
        if self.device.type == "cuda" and torch.cuda.is_available():
            self._finalize()
        return self

    def _finalize(self):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cmtts_vocoder_finalize(self._h))
        self._pending = False
        self._ready = True

    def remove_weight_norm(self):
        return self

    def set_precision(self, dtype="fp32"):
        """Operand precision of the ResBlock convs: "fp32" (reference), "bf16", "fp16", or "fp16x3" (operands as hi + lo fp16
        pairs, three fp16 MFMAs per product, fp32 accumulate: fp32-class results, activations stay fp32 in HBM)."""
        mode = {"fp32": 0, "f32": 0, "bf16": 1, "fp16": 2, "f16": 2, "fp16x3": 3}[dtype]
        _lib.check(self.lib.cmtts_vocoder_set_precision(self._h, mode))
        return self

    def set_option(self, name, value):
        """Per-vocoder numerics option (cmtts_vocoder_set_option): "ups16" 1 (default) | 0 — in the 16-bit modes the upsamplers
        take 16-bit operands too, or stay fp32.  Returns the previous value.  "winograd" 1 (default) | 0 — fp32 generator, large batches: the ResBlock convs of the C >= 128 stages as Winograd
        convolutions (4 / 10 / 15 products per output pair instead of 6 / 14 / 22; <= 1.2e-6 on the waveform) or in the direct form."""
        prev = self.lib.cmtts_vocoder_set_option(self._h, name.encode() if isinstance(name, str) else name, int(value))
        if prev < 0:
            _lib.check(prev)
        return prev

    def eval(self):
        return self

    def forward(self, x):
        if not self._ready and getattr(self, "_pending", False):
            if not torch.cuda.is_available():
                raise RuntimeError("Generator: no GPU — cmtts_amd has no CPU fallback (the weights are loaded, the kernels cannot run)")
            self.device = _norm_device(self.device if self.device.type == "cuda" else "cuda")
            self._finalize()
        if not self._ready:
            raise RuntimeError("Generator: load_state_dict() first")
        dev = self.device
        x = _f32(x, dev)
        B, M, T = x.shape
        with torch.cuda.device(dev):
            wav = torch.empty(B, 1, T * self.h.hop, dtype=torch.float32, device=dev)
            nb = self.lib.cmtts_vocoder_workspace_bytes(self._h, B, T)
            ws = self._ws.get("voc", nb, dev)
            _lib.check(self.lib.cmtts_vocoder_forward(self._h, _ptr(x), B, T, _ptr(wav), _ptr(ws), nb, _stream()))
        return wav


def vocoder_infer_device(mels, vocoder, max_wav_value=32768.0):
    """The device half of vocoder_infer (utils/model.py:187-198): mels [B,80,T] -> int16 PCM [B, T*hop] still on the GPU
    (what shard.allgather_pcm collates across ranks before anything crosses PCIe)."""
    wavs = vocoder(mels).squeeze(1)
    pcm = torch.empty(wavs.shape, dtype=torch.int16, device=wavs.device)
    with torch.cuda.device(wavs.device):
        _lib.check(vocoder.lib.cmtts_wav_to_int16(_ptr(wavs), _ptr(pcm), wavs.numel(), float(max_wav_value), _stream()))
    return pcm


def vocoder_infer(mels, vocoder, model_config=None, preprocess_config=None, lengths=None, max_wav_value=32768.0):
    """utils/model.py:187-205: mels [B,80,T] -> list of int16 numpy arrays trimmed to `lengths`."""
    if preprocess_config is not None:
        max_wav_value = preprocess_config["preprocessing"]["audio"]["max_wav_value"]
    pcm = vocoder_infer_device(mels, vocoder, max_wav_value)
    out = [w for w in pcm.cpu().numpy()]
    check_async_error()          # the D2H copy synchronised: a timeout in the launches that produced `mels` is raised here
    if lengths is not None:
        out = [w[: int(lengths[i])] for i, w in enumerate(out)]
    return out


def synth_samples(args, targets, predictions, vocoder, model_config, preprocess_config, path, diffusion=None):
    """utils/tools.py:566-607 (same argument list), the part that is on the inference path: mel -> vocoder_infer -> one int16
    .wav per utterance at the dataset's sampling rate (22 050 Hz), written to <path>/<args.restore_step>/ under the
    reference's names: "<basename>_<speaker_id><tag>.wav" for a multi-speaker model in single mode, "<basename><tag>.wav"
    otherwise (tag = "_teacher_forced" when args.teacher_forced).  `targets` is the synthesize batch (ids first),
    `predictions` the out_put list of CMTotalTTSSynthesize.synthesize (mel [B,T,80] in slot 0, mel_lens in slot 11).  The
    mel-spectrogram PNGs of :573-594 are matplotlib host work and are not produced.  Returns the paths written."""
    import os
    from scipy.io import wavfile
    multi_speaker = model_config["multi_speaker"]
    tag = "_teacher_forced" if getattr(args, "teacher_forced", False) else ""
    basenames = targets[0]
    mel_predictions = predictions[0].transpose(1, 2)            # [B,80,T]
    lengths = predictions[11] * preprocess_config["preprocessing"]["stft"]["hop_length"]
    wav_predictions = vocoder_infer(mel_predictions, vocoder, model_config, preprocess_config, lengths=lengths.tolist())
    sampling_rate = preprocess_config["preprocessing"]["audio"]["sampling_rate"]
    out_dir = os.path.join(path, str(getattr(args, "restore_step", "")))
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for wav, basename in zip(wav_predictions, basenames):
        single = multi_speaker and getattr(args, "mode", None) == "single"
        name = "{}_{}{}.wav".format(basename, args.speaker_id, tag) if single else "{}{}.wav".format(basename, tag)
        out = os.path.join(out_dir, name)
        wavfile.write(out, sampling_rate, wav)
        written.append(out)
    return written


def transpose_last2(x):
    """[B,R,C] -> [B,C,R] on the HIP transpose kernel (cmtts_transpose)."""
    x = x.contiguous()
    B, R, Cn = x.shape
    out = torch.empty(B, Cn, R, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cmtts_transpose(_ptr(x), _ptr(out), B, R, Cn, _stream()))
    return out


class StreamPipelinedSynthesizer:
    """Throughput mode for a stream of batches: the phoneme/frame-level conditioning of batch i+1 (many small,
    latency-bound launches that cannot fill 256 CUs) runs on a second HIP stream underneath the MFMA-bound
    sampler of batch i.  Same kernels, same results per batch; only the issue order across batches changes."""

    def __init__(self, model: CMTotalTTS, n_steps=4):
        self.model, self.n_steps = model, n_steps
        self.side = torch.cuda.Stream(device=model.device)
        self.ready = None          # (event, out_dict) of the batch whose conditioning is already computed
        self._keep = []

    def prepare(self, texts, src_lens, spker_embeds=None, max_mel_len=None):
        main = torch.cuda.current_stream(self.model.device)
        self.side.wait_stream(main)                       # inputs / workspaces written on main are visible
        with torch.cuda.stream(self.side):
            out = self.model.duration_pitch_energy_net(None, texts, src_lens, spker_embeds=spker_embeds,
                                                       max_mel_len=max_mel_len)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.ready = (ev, out)

    def sample_and_prepare_next(self, noise, next_batch):
        """Sample the prepared batch on the current stream while `next_batch` = (texts, src_lens, spk,
        max_mel_len) is conditioned on the side stream.  Returns (mel, mel_lens) of the prepared batch."""
        ev, out = self.ready
        torch.cuda.current_stream(self.model.device).wait_event(ev)
        f = out.get("cond_factors")
        for v in (out["cond_ct"], out["speaker_emb"]) + ((f.p1, f.p1t, f.mel2ph, f.p_idx) if f is not None else ()):
            if v is not None:
                v.record_stream(torch.cuda.current_stream(self.model.device))
        self._keep = [out]
        if next_batch is not None:
            self.prepare(*next_batch)
        mel = sample_with_cond(self.model, out["cond_ct"], out["speaker_emb"], self.n_steps, noise, factors=out.get("cond_factors"))
        return mel, out["mel_lens"]


ATTN_SHORT_MAX = 192        # attention.hip: all keys in registers up to here, key-chunked online softmax above (not bitwise the same)


class CollatedShard:
    """A ragged shard's bucket groups collated for ONE phoneme-level call per attention class (cmtts_text_forward_ragged): the
    host-side counterpart of the reference's collate_fn padding (dataset.py), done once per shard.  `batches`: list of dicts with
    texts [Bt,Lt], src_lens [Bt], pad_lens [Bt] (each utterance's own group's padded phoneme count), spk [Bt,D] | None and
    `members` = [(group index, b0, n, Lg)]; `groups` keeps (noise, bucket) per group."""

    def __init__(self, groups, device):
        # a group may carry a sixth element: speakers int64 [n] (models whose speaker_emb is an nn.Embedding table, model/cmtts.py:77-78)
        self.groups = [(g[3], int(g[4])) for g in groups]
        self.n = [int(g[0].shape[0]) for g in groups]
        classes = {}
        for i, g in enumerate(groups):
            classes.setdefault(int(g[0].shape[1]) > ATTN_SHORT_MAX, []).append(i)      # groups that take the same attention kernel alone and batched
        self.batches = []
        for _, idx in sorted(classes.items()):
            Lt = max(int(groups[i][0].shape[1]) for i in idx)
            Bt = sum(int(groups[i][0].shape[0]) for i in idx)
            texts = torch.zeros(Bt, Lt, dtype=torch.int64, device=device)
            has_spk = groups[idx[0]][2] is not None
            has_ids = len(groups[idx[0]]) > 5 and groups[idx[0]][5] is not None
            src, pad, spk, ids, members, b0 = [], [], [], [], [], 0
            for i in idx:
                tx, ln, sp = groups[i][0], groups[i][1], groups[i][2]
                if has_ids:
                    ids.append(groups[i][5].to(device=device, dtype=torch.int64))
                n, Lg = tx.shape
                texts[b0:b0 + n, :Lg] = tx.to(device)
                src.append(ln.to(device=device, dtype=torch.int64))
                pad.append(torch.full((n,), Lg, dtype=torch.int64, device=device))
                if has_spk:
                    spk.append(sp.to(device=device, dtype=torch.float32))
                members.append((i, b0, n, Lg))
                b0 += n
            self.batches.append({"texts": texts, "src_lens": torch.cat(src), "pad_lens": torch.cat(pad),
                                 "spk": torch.cat(spk) if has_spk else None, "speakers": torch.cat(ids) if has_ids else None,
                                 "members": members})


def collate_groups(groups, device):
    """groups: [(texts [n,Lg], src_lens [n], spker_embeds [n,D] | None, noise, bucket)] -> CollatedShard."""
    return CollatedShard(list(groups), _norm_device(device))


def _text_forward_ragged(model, tb, key):
    """One cmtts_text_forward_ragged call for a collated text batch.  Returns (text workspace, B, L, mel_len [B], speaker_emb [B,H] | None)."""
    lib, dev, cfg = model.lib, model.device, model.config
    texts, src, pad, spk_in = tb["texts"], tb["src_lens"], tb["pad_lens"], tb["spk"]
    B, L = texts.shape
    table = cfg.multi_speaker and cfg.n_speaker > 0
    ids = tb.get("speakers") if table else None
    if table:
        if ids is None:
            raise AssertionError("speakers (ids into the speaker_emb table) should not be None")
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= cfg.n_speaker):
            raise IndexError("index out of range in self")
        spk_in = None
    elif cfg.multi_speaker and spk_in is None:
        raise AssertionError("Speaker embedding should not be None")
    with torch.cuda.device(dev):
        mel_len = torch.empty(B, dtype=torch.int64, device=dev)
        spk = torch.empty(B, cfg.hidden, dtype=torch.float32, device=dev) if cfg.multi_speaker else None
        nb = lib.cmtts_text_workspace_bytes(model._h, B, L)
        tws = model._ws.get(key, nb, dev)
        _lib.check(lib.cmtts_text_forward_ragged(model._h, _ptr(texts), _ptr(src), _ptr(pad), _ptr(spk_in), _ptr(ids), B, L, 1.0,
                                                 None, None, _ptr(mel_len), None, None, None, _ptr(spk), _ptr(tws), nb, _stream()))
    return tws, B, L, mel_len, spk


def _frame_forward_sub(model, tws, B_all, L_all, b0, n, T, key, want_factors=True):
    """cmtts_frame_forward_sub for utterances [b0, b0 + n) of a text workspace -> (cond_ct, CondFactors | None)."""
    lib, dev, cfg = model.lib, model.device, model.config
    with torch.cuda.device(dev):
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        cond_ct = f(n, cfg.hidden, T)
        mel2ph = torch.empty(n, T, dtype=torch.int64, device=dev)
        p_idx = torch.empty(n, T, dtype=torch.int64, device=dev)
        p1_ld = (L_all + 3) // 4 * 4
        p1 = f(n, cfg.res_layers * cfg.res_channels, p1_ld) if want_factors and getattr(model, "_precision_mode", 0) == 0 and model._cond_factors else None
        nf = lib.cmtts_frame_workspace_bytes(model._h, n, T)
        fws = model._ws.get(key, nf, dev)
        _lib.check(lib.cmtts_frame_forward_sub(model._h, _ptr(tws), B_all, L_all, b0, n, T, _ptr(cond_ct), _ptr(mel2ph), None, None,
                                               _ptr(p_idx), None, _ptr(p1), _ptr(fws), nf, _stream()))
    factors = None if p1 is None else CondFactors(p1, p1_ld, L_all, mel2ph, p_idx, cond_ct)
    cond_ct._cmtts_factors = factors
    return cond_ct, factors


class BucketedSynthesizer:
    """BASELINE.json configs[3] on one rank: a shard of ragged utterances dealt into static frame buckets
    (cmtts_amd.shard.frame_bucket).  A bucket group alone cannot fill 256 CUs.  mode "ragged" (default): the text side of every
    group runs on its own HIP stream (short launches that overlap), then ONE cmtts_sample_ragged call runs the residual layers of
    all groups in one persistent launch per evaluation, each utterance trimmed to the frames its mel_len (+ tail_frames) needs
    — the one host read-back of the shard (the reference's length regulator reads back per phoneme, model/modules.py:439-441).
    mode "streams": every group end to end on its own stream (round 2).  Either way each group's valid frames are those of
    running the group alone."""

    def __init__(self, model: CMTotalTTS, n_steps=4, n_streams=4, persistent=None, mode="ragged", tail_frames=16, trim=True, batch_text=None):
        """persistent: denoiser mode while the groups run in "streams" mode (cmtts_set_persistent_denoiser; None = leave the
        process setting alone).  tail_frames: frames beyond mel_len that must be exact in the padded mel (16 covers HiFi-GAN's
        receptive field; the frames beyond the computed range are zeros instead of denoised padding).  trim=False computes
        every padded frame (bit-identical to the per-group launches everywhere)."""
        self.model, self.n_steps, self.persistent, self.mode = model, n_steps, persistent, mode
        self.tail_frames, self.trim = int(tail_frames), bool(trim)
        self.streams = [torch.cuda.Stream(device=model.device) for _ in range(n_streams)]
        # round 4, mode "ragged": the phoneme-level half of ALL groups in one cmtts_text_forward_ragged call (one launch sequence for the
        # shard instead of one per group), then each group's frame-level half on its own stream; False = one text side per group (round 3)
        # None = automatic: batched once a process group exists (measured on MI355X, configs[3] shard: with RCCL loaded the one-call text side
        # runs the shard at 860-870 k frames/s against 824-834 k for one text side per group on four streams; without RCCL 870-880 k against
        # 877-881 k) — a CollatedShard passed to run() is always batched
        if batch_text is None:
            batch_text = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.batch_text = bool(batch_text)

    def _early_groups(self, sizes):
        """sizes: [(n, bucket)] -> indices of the small groups that run end to end on their own stream (see run())."""
        early = set()
        cap = torch.cuda.get_device_properties(self.model.device).multi_processor_count
        tiles = [n * ((bucket + 63) // 64) for n, bucket in sizes]
        order = sorted(range(len(sizes)), key=lambda i: tiles[i])
        rest, gone = sum(tiles), 0
        for i in order:
            if rest <= cap or gone + tiles[i] > 64:
                break
            early.add(i); rest -= tiles[i]; gone += tiles[i]
        if not (rest * 0.92 <= cap):       # would not fit one round even after trimming: let the library decide
            early = set()
        return early

    def _run_batched(self, coll):
        """mode "ragged", fp32: one phoneme-level call per attention class for the whole shard, the frame-level halves per group on
        the streams, set-aside candidates sampled end to end on theirs, ONE cmtts_sample_ragged for the rest."""
        model, lib, dev = self.model, self.model.lib, self.model.device
        main = torch.cuda.current_stream(dev)
        ng = len(coll.groups)
        early = self._early_groups([(coll.n[i], coll.groups[i][1]) for i in range(ng)]) if self.trim else set()
        conds, lens, done, late_events = [None] * ng, [None] * ng, {}, []
        text_done = []
        for k, tb in enumerate(coll.batches):          # the library's branch streams stay ON here: one call, nothing to compete with
            tws, Bt, Lt, mel_len, spk = _text_forward_ragged(model, tb, ("text_ragged", k))
            ev = torch.cuda.Event()
            ev.record(main)
            text_done.append((tb, tws, Bt, Lt, mel_len, spk, ev))
        prev_branch = lib.cmtts_set_option(b"branch_streams", 0)       # the groups overlap across the streams from here on
        try:
            order = []
            for tb, tws, Bt, Lt, mel_len, spk, ev in text_done:
                for (i, b0, n, Lg) in tb["members"]:
                    order.append((i not in early, i, b0, n, tws, Bt, Lt, mel_len, spk, ev))
            for _, i, b0, n, tws, Bt, Lt, mel_len, spk, ev in sorted(order, key=lambda t: t[:2]):      # the early groups are queued first
                noise, bucket = coll.groups[i]
                st = self.streams[i % len(self.streams)]
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    cond_ct, factors = _frame_forward_sub(model, tws, Bt, Lt, b0, n, bucket, ("frame_ragged", i))
                    spk_i = None if spk is None else spk[b0:b0 + n]
                    lens[i] = mel_len[b0:b0 + n]
                    if i in early:
                        prev_p = lib.cmtts_set_persistent_denoiser(0)
                        try:
                            done[i] = (sample_with_cond(model, cond_ct, spk_i, self.n_steps, noise, factors=factors), lens[i])
                        finally:
                            lib.cmtts_set_persistent_denoiser(prev_p)
                    else:
                        e2 = torch.cuda.Event()
                        e2.record(st)
                        late_events.append(e2)
                    conds[i] = (cond_ct, spk_i, factors)
        finally:
            lib.cmtts_set_option(b"branch_streams", prev_branch)
        for e2 in late_events:
            main.wait_event(e2)
        late = [i for i in range(ng) if i not in early]
        active = [None] * len(late)
        if self.trim and late:      # one device -> host copy for the whole shard
            flat = torch.cat([lens[i] for i in late]).cpu().tolist()
            k = 0
            for gi, i in enumerate(late):
                active[gi] = flat[k:k + coll.n[i]]
                k += coll.n[i]
        mels = sample_ragged(model, [(conds[i][0], conds[i][1], coll.groups[i][0], a, conds[i][2]) for i, a in zip(late, active)],
                             self.n_steps, self.tail_frames) if late else []
        for i, mel in zip(late, mels):
            done[i] = (mel, lens[i])
        for st in self.streams:
            main.wait_stream(st)
        return [done[i] for i in range(ng)]

    def run(self, groups):
        """groups: iterable of (texts, src_lens, spker_embeds | None, noise [n_steps+1,n,1,bucket,80], bucket), or the CollatedShard
        collate_groups() made of them (collation = input preparation, once per shard).
        Returns [(mel [n,bucket,80], mel_lens [n])] in the same order."""
        dev = self.model.device
        if not isinstance(groups, CollatedShard):
            groups = list(groups)
        table = self.model.config.multi_speaker and self.model.config.n_speaker > 0
        no_ids = table and not isinstance(groups, CollatedShard) and any(len(g) < 6 or g[5] is None for g in groups)
        if self.mode == "ragged" and (self.batch_text or isinstance(groups, CollatedShard)) and getattr(self.model, "_precision_mode", 0) == 0 and not no_ids:
            return self._run_batched(groups if isinstance(groups, CollatedShard) else collate_groups(groups, dev))
        if isinstance(groups, CollatedShard):
            raise ValueError("a CollatedShard needs mode='ragged' with batch_text on an fp32 model")
        main = torch.cuda.current_stream(dev)
        out = []
        lib = self.model.lib
        prev = lib.cmtts_set_persistent_denoiser(self.persistent) if self.persistent is not None and (self.mode != "ragged" or getattr(self.model, "_precision_mode", 0) != 0) else None
        # the groups already overlap across streams; the library's own side streams (independent branches of one group)
        # would only add streams competing for the few hardware queues HIP maps them onto (25 -> 31-39 ms measured)
        prev_branch = lib.cmtts_set_option(b"branch_streams", 0)
        groups = list(groups)
        mode = self.mode
        if mode == "ragged" and getattr(self.model, "_precision_mode", 0) != 0:
            mode = "streams"      # the one-launch form is the fp32 persistent kernel's: 16-bit models keep one stream per bucket group
        # "ragged": a shard whose padded tiles exceed the CU count needs a second round of the persistent launch unless a few SMALL
        # groups are left out (cmtts_sample_ragged sets them aside itself once the lengths are known).  When the padded sizes already
        # say so, those groups run END TO END on their own stream right away — their sampler (per-layer kernels) then overlaps the
        # text side of the large groups instead of following it.  The guess only moves work between streams: results are per group.
        early = set()
        if mode == "ragged" and self.trim:
            cap = torch.cuda.get_device_properties(dev).multi_processor_count
            tiles = [int(g[0].shape[0]) * ((int(g[4]) + 63) // 64) for g in groups]
            order = sorted(range(len(groups)), key=lambda i: tiles[i])
            rest, gone = sum(tiles), 0
            for i in order:
                if rest <= cap or gone + tiles[i] > 64:
                    break
                early.add(i); rest -= tiles[i]; gone += tiles[i]
            if not (rest * 0.92 <= cap):       # would not fit one round even after trimming: let the library decide
                early = set()
        try:
            conds = []
            done = {}
            late_events = []
            for i in sorted(range(len(groups)), key=lambda i: i not in early):      # the early groups are queued first
                texts, src_lens, spk, noise, bucket = groups[i][:5]
                ids = groups[i][5] if len(groups[i]) > 5 else None
                st = self.streams[i % len(self.streams)]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    o = self.model.duration_pitch_energy_net(ids, texts, src_lens, spker_embeds=spk, max_mel_len=bucket)
                    if mode != "ragged" or i in early:
                        prev_p = lib.cmtts_set_persistent_denoiser(0) if i in early else None
                        mel = sample_with_cond(self.model, o["cond_ct"], o["speaker_emb"], self.n_steps, noise, factors=o.get("cond_factors"))
                        if prev_p is not None:
                            lib.cmtts_set_persistent_denoiser(prev_p)
                        done[i] = (mel, o["mel_lens"])
                    if i not in early:      # main waits for THIS group's text side, whatever else shares its stream
                        ev = torch.cuda.Event()
                        ev.record(st)
                        late_events.append(ev)
                conds.append((i, o))
            conds = [o for _, o in sorted(conds, key=lambda t: t[0])]
            if mode != "ragged":
                out = [done[i] for i in range(len(groups))]
        finally:
            lib.cmtts_set_option(b"branch_streams", prev_branch)
            if prev is not None:
                lib.cmtts_set_persistent_denoiser(prev)
        # every late group's text side is ordered before main by ITS OWN event: with more groups than streams a late group can
        # share an early group's stream, whose full join only happens after the persistent launches are queued (below)
        for ev in late_events:
            main.wait_event(ev)
        early_streams = {id(self.streams[i % len(self.streams)]) for i in early}
        for st in self.streams:
            if id(st) not in early_streams:
                main.wait_stream(st)
        if mode == "ragged":
            late = [i for i in range(len(groups)) if i not in early]
            lens = [conds[i]["mel_lens"] for i in late]
            active = [None] * len(late)
            if self.trim and late:      # one device -> host copy for the whole shard
                flat = torch.cat(lens).cpu().tolist()
                k = 0
                for gi, l in enumerate(lens):
                    active[gi] = flat[k:k + l.numel()]
                    k += l.numel()
            mels = sample_ragged(self.model, [(conds[i]["cond_ct"], conds[i]["speaker_emb"], groups[i][3], a, conds[i].get("cond_factors"))
                                              for i, a in zip(late, active)],
                                 self.n_steps, self.tail_frames) if late else []
            for i, mel, l in zip(late, mels, lens):
                done[i] = (mel, l)
            out = [done[i] for i in range(len(groups))]
        for st in self.streams:
            if id(st) in early_streams:
                main.wait_stream(st)
        return out


class CMTotalTTSSynthesize:
    """synthesize.py:35-153 with the reference's constructor: CMTotalTTSSynthesize(model_path, model_step_num, args,
    preprocess_config, model_config, train_config, p_control, e_control, d_control).  The checkpoint
    <model_path>/CMDenoiserTTS/model{step:06d}.pt is read with torch.load and handed to CMTotalTTS.load_state_dict
    (cmtts_set_tensor + cmtts_finalize).  `from_model` keeps the model-object form (the reference reloads the checkpoint
    for every batch, :203-206; a serving host builds the model once)."""

    def __init__(self, model_path, model_step_num, args, preprocess_config, model_config, train_config,
                 p_control=1.0, e_control=1.0, d_control=1.0, device=None, generator=None, n_speaker=None):
        import os.path as osp
        self.device = _norm_device(device if device is not None else "cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.CMDenoiserTTS_path = osp.join(model_path, "CMDenoiserTTS", "model{:06d}.pt".format(int(model_step_num)))
        self.args = args
        self.train_config = train_config
        self.model, self.diffusion = self.load_cm_model(args, preprocess_config, model_config, train_config, n_speaker=n_speaker)
        self.duration_pitch_energy_net, self.denoise_net = self.model.get_segmentation_model()
        self.p_control, self.e_control, self.d_control = p_control, e_control, d_control
        self.generator = generator

    @classmethod
    def from_model(cls, model: CMTotalTTS, T=1, generator=None, p_control=1.0, e_control=1.0, d_control=1.0):
        """A synthesizer around an already loaded model (no checkpoint I/O); T = sampling steps (args.T)."""
        import argparse
        self = cls.__new__(cls)
        cfg = model.config
        self.device = model.device
        self.CMDenoiserTTS_path = None
        self.args = argparse.Namespace(T=int(T))
        self.train_config = {"cm": {"sigma_min": cfg.sigma_min, "sigma_max": cfg.sigma_max}}
        self.model = model
        self.diffusion = KarrasDenoiser(sigma_data=cfg.sigma_data, sigma_max=cfg.sigma_max, sigma_min=cfg.sigma_min,
                                        rho=cfg.rho, distillation=True)
        self.duration_pitch_energy_net, self.denoise_net = model.get_segmentation_model()
        self.p_control, self.e_control, self.d_control = p_control, e_control, d_control
        self.generator = generator
        return self

    def load_cm_model(self, args, preprocess_config, model_config, train_config, n_speaker=None):
        """synthesize.py:59-86: distillation from cm.training_mode, model + diffusion from the `cm` block
        (script_util.py:56-76: sigma_min / sigma_max / sigma_data / rho, weight_schedule, loss_norm), torch.load of the
        checkpoint -> load_state_dict -> to(device) -> eval()."""
        from .config import config_from_reference
        cm = train_config["cm"]
        mode = cm["training_mode"]
        if mode == "progdist":
            distillation = False
        elif "consistency" in mode:
            distillation = True
        else:
            raise ValueError(f"unknown training mode {mode}")
        cfg = config_from_reference(preprocess_config, model_config, train_config, n_speaker=n_speaker)
        model = CMTotalTTS(cfg, self.device)
        diffusion = KarrasDenoiser(sigma_data=cfg.sigma_data, sigma_max=cfg.sigma_max, sigma_min=cfg.sigma_min, rho=cfg.rho,
                                   weight_schedule=cm.get("weight_schedule", "karras"), distillation=distillation,
                                   loss_norm=cm.get("loss_norm", "lpips"))
        state = torch.load(self.CMDenoiserTTS_path, map_location="cpu")
        model.load_state_dict(state)
        model.to(self.device)
        model.eval()
        return model, diffusion

    def synthesize(self, batch):
        """batch = (ids, raw_texts, speakers, texts, src_lens, max_src_len, spker_embeds) after to_device (:88-153)."""
        kw = {"speakers": batch[2], "texts": batch[3], "src_lens": batch[4], "spker_embeds": batch[-1]}
        out_dict = self.duration_pitch_energy_net(**kw)
        B, T, _ = out_dict["cond"].shape
        cfg = self.model.config
        steps = int(self.args.T)
        if steps == 1:
            n_steps, draws = 1, 1
        elif steps in (2, 4):
            n_steps, draws = steps, steps + 1
        else:
            raise ValueError("T must be 1, 2 or 4 (synthesize.py:111-147)")
        cm = self.train_config["cm"]
        fusable = getattr(self.diffusion, "distillation", False) and \
            abs(float(cm.get("sigma_max", cfg.sigma_max)) - cfg.sigma_max) < 1e-6 * cfg.sigma_max and \
            abs(float(cm.get("sigma_min", cfg.sigma_min)) - cfg.sigma_min) < 1e-6 * cfg.sigma_min
        gen = self.generator or DummyGenerator()
        if not fusable:      # e.g. a progdist teacher: the reference's sampler loops, host-side (karras_sample_tts routes)
            sample = karras_sample_tts(self.diffusion, self.model, (B, 1, T, cfg.n_mels), steps=2, model_kwargs=kw,
                                       device=self.device, sigma_min=float(cm.get("sigma_min", cfg.sigma_min)),
                                       sigma_max=float(cm.get("sigma_max", cfg.sigma_max)),
                                       sampler="onestep" if steps == 1 else "multistep",
                                       ts=None if steps == 1 else (0,) * steps + (1,), generator=gen)
        else:                # the duration net ran once above; the reference's in-sampler re-runs are bit-identical
            x0 = gen.randn(B, 1, T, cfg.n_mels, device=self.model.device)
            noise = torch.stack([x0] + [gen.randn_like(x0) for _ in range(draws - 1)], 0).float()
            sample = sample_with_cond(self.model, out_dict["cond_ct"], out_dict["speaker_emb"], n_steps, noise,
                                      factors=out_dict.get("cond_factors"))
        out_put = [None] * 12
        out_put[0] = sample
        out_put[10] = kw["src_lens"]
        out_put[11] = out_dict["mel_lens"]
        return out_put


def get_vocoder(config, device, root="."):
    """utils/model.py:155-184: config["vocoder"] = {"model": "HiFi-GAN", "speaker": "LJSpeech" | "universal"} ->
    hifigan/config.json + hifigan/generator_<speaker>.pth.tar (ckpt["generator"], weight-norm pairs folded like
    remove_weight_norm) under `root` (the reference resolves them against the working directory).  MelGAN comes from
    torch.hub (network) in the reference and is not part of this path."""
    import json
    import os
    name, speaker = config["vocoder"]["model"], config["vocoder"]["speaker"]
    if name != "HiFi-GAN":
        raise NotImplementedError(f"vocoder {name!r}: only HiFi-GAN is on the hot path (MelGAN needs torch.hub)")
    if speaker not in ("LJSpeech", "universal"):
        raise ValueError(f"unknown vocoder speaker {speaker!r}")
    with open(os.path.join(root, "hifigan", "config.json")) as f:
        hc = json.load(f)
    h = HifiGanConfig(num_mels=hc.get("num_mels", 80), upsample_rates=tuple(hc["upsample_rates"]),
                      upsample_kernel_sizes=tuple(hc["upsample_kernel_sizes"]),
                      upsample_initial_channel=hc["upsample_initial_channel"],
                      resblock_kernel_sizes=tuple(hc["resblock_kernel_sizes"]),
                      resblock_dilation_sizes=tuple(tuple(d) for d in hc["resblock_dilation_sizes"]))
    if h != HifiGanConfig():
        raise NotImplementedError("the vocoder kernels are specialised for hifigan/config.json (V1 generator)")
    ckpt = torch.load(os.path.join(root, "hifigan", f"generator_{speaker}.pth.tar"), map_location="cpu")
    vocoder = Generator(h, device)
    vocoder.load_state_dict(ckpt["generator"])
    vocoder.eval()
    vocoder.remove_weight_norm()
    return vocoder
