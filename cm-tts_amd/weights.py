"""Synthetic checkpoint generator (no trained weights exist: .MISSING_LARGE_BLOBS in the reference).

Produces flat state_dicts with exactly the reference's key names and tensor layouts
(``CMTotalTTS.state_dict()`` as saved by model/cm_tool/train_util.py:890-899, and
``hifigan.Generator.state_dict()`` after ``remove_weight_norm()``, utils/model.py:175-181), so the
same dict can be (a) loaded into the imported reference in this container to make golden vectors,
(b) fed to the numpy oracle, (c) pushed through the C-ABI weight importer.

Every tensor is drawn from its own ``numpy.random.RandomState`` seeded by crc32(name) ^ seed — the
legacy generator's stream is frozen, so the values are identical on every machine.
Scales are fan-in normalised so that activations stay O(1) through all 20 residual layers and the
vocoder (the reference's own initialisers give a zero denoiser output — Denoiser.output_projection
is zero-initialised, model/modules.py:598 — which would make parity checks vacuous).
"""
import math
import zlib
from collections import OrderedDict

import numpy as np

from .config import CMTTSConfig, HifiGanConfig


def _rs(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def _normal(name, seed, shape, std, mean=0.0):
    return (_rs(name, seed).standard_normal(size=shape) * std + mean).astype(np.float32)


def synth_decoder_state_dict(cfg: CMTTSConfig, seed: int = 0, n_layers: int = 4) -> "OrderedDict[str, np.ndarray]":
    """State dict of a FastspeechDecoder (model/modules.py:154-165) registered as `decoder` — the keys
    `self.decoder = FastspeechDecoder(model_config)` would add to a CMTotalTTS checkpoint (CM-TTS defines the class but
    never instantiates it): decoder.layers.N.op.*, decoder.layer_norm.*, decoder.pos_embed_alpha (a learnable scalar;
    set away from its init value 1 here so that tests see it)."""
    H = cfg.hidden
    sd = OrderedDict()

    def N(name, shape, std, mean=0.0):
        sd[name] = _normal(name, seed, shape, std, mean)

    for i in range(n_layers):
        p = f"decoder.layers.{i}.op."
        for l in ("layer_norm1", "layer_norm2"):
            N(p + l + ".weight", (H,), 0.1, 1.0)
            N(p + l + ".bias", (H,), 0.1)
        N(p + "self_attn.in_proj_weight", (3 * H, H), 1.0 / math.sqrt(H))
        N(p + "self_attn.out_proj.weight", (H, H), 1.0 / math.sqrt(H))
        N(p + "ffn.ffn_1.weight", (4 * H, H, cfg.ffn_kernel), 1.5 / math.sqrt(H * cfg.ffn_kernel))
        N(p + "ffn.ffn_1.bias", (4 * H,), 0.1)
        N(p + "ffn.ffn_2.weight", (H, 4 * H), 1.0 / math.sqrt(4 * H))
        N(p + "ffn.ffn_2.bias", (H,), 0.1)
    N("decoder.layer_norm.weight", (H,), 0.1, 1.0)
    N("decoder.layer_norm.bias", (H,), 0.1)
    sd["decoder.pos_embed_alpha"] = np.asarray([0.7], np.float32)
    sd["decoder.embed_positions._float_tensor"] = np.zeros((1,), np.float32)
    return sd


def synth_cmtts_state_dict(cfg: CMTTSConfig, seed: int = 0, dur_frames: float = 6.0,
                           dur_spread: float = 0.02) -> "OrderedDict[str, np.ndarray]":
    """State dict of CMTotalTTS for one dataset variant.

    dur_frames/dur_spread steer the duration predictor's last Linear so that predicted durations
    centre on ``dur_frames`` frames per phoneme (bias = ln(dur_frames+1)); dur_spread=0 forces
    them all equal (the synthetic-benchmark setting).
    """
    H = cfg.hidden
    sd = OrderedDict()

    def N(name, shape, std, mean=0.0):
        sd[name] = _normal(name, seed, shape, std, mean)

    def ln(prefix):
        N(prefix + ".weight", (H,), 0.1, 1.0)
        N(prefix + ".bias", (H,), 0.1)

    enc = "duration_pitch_energy_net.text_encoder."
    for i in range(cfg.enc_layers):
        p = f"{enc}layers.{i}.op."
        ln(p + "layer_norm1")
        N(p + "self_attn.in_proj_weight", (3 * H, H), 1.0 / math.sqrt(H))
        N(p + "self_attn.out_proj.weight", (H, H), 1.0 / math.sqrt(H))
        ln(p + "layer_norm2")
        N(p + "ffn.ffn_1.weight", (4 * H, H, cfg.ffn_kernel), 1.5 / math.sqrt(H * cfg.ffn_kernel))
        N(p + "ffn.ffn_1.bias", (4 * H,), 0.1)
        N(p + "ffn.ffn_2.weight", (H, 4 * H), 1.0 / math.sqrt(4 * H))
        N(p + "ffn.ffn_2.bias", (H,), 0.1)
    ln(enc + "layer_norm")
    N(enc + "embed_tokens.weight", (cfg.n_symbols, H), H ** -0.5)
    sd[enc + "embed_tokens.weight"][0] = 0.0
    sd[enc + "embed_positions._float_tensor"] = np.zeros((1,), np.float32)

    va = "duration_pitch_energy_net.variance_adaptor."
    sd[va + "energy_bins"] = np.linspace(cfg.energy_min, cfg.energy_max, cfg.energy_bins - 1).astype(np.float32)

    def predictor(prefix, idim, n_layers, k, odim, lin_std, lin_bias):
        for li in range(n_layers):
            cin = idim if li == 0 else cfg.pred_filter
            N(f"{prefix}conv.{li}.1.weight", (cfg.pred_filter, cin, k), math.sqrt(2.0 / (cin * k)))
            N(f"{prefix}conv.{li}.1.bias", (cfg.pred_filter,), 0.1)
            N(f"{prefix}conv.{li}.3.weight", (cfg.pred_filter,), 0.1, 1.0)
            N(f"{prefix}conv.{li}.3.bias", (cfg.pred_filter,), 0.1)
        N(prefix + "linear.weight", (odim, cfg.pred_filter), lin_std)
        sd[prefix + "linear.bias"] = np.asarray(lin_bias, np.float32).reshape(odim)

    predictor(va + "duration_predictor.", H, cfg.dur_layers, cfg.dur_kernel, 1,
              dur_spread, [math.log(dur_frames + 1.0)])
    N(va + "cwt_predictor.0.weight", (cfg.cwt_hidden, H), 1.0 / math.sqrt(H))
    N(va + "cwt_predictor.0.bias", (cfg.cwt_hidden,), 0.1)
    sd[va + "cwt_predictor.1.pos_embed_alpha"] = np.asarray([0.9], np.float32)
    cwt_bias = _normal(va + "cwt_predictor.1.linear.bias", seed, (cfg.cwt_out,), 0.3)
    predictor(va + "cwt_predictor.1.", cfg.cwt_hidden, cfg.pred_layers, cfg.pred_kernel, cfg.cwt_out,
              0.06, cwt_bias)
    sd[va + "cwt_predictor.1.embed_positions._float_tensor"] = np.zeros((1,), np.float32)
    N(va + "cwt_stats_layers.0.weight", (cfg.cwt_hidden, H), 1.0 / math.sqrt(H))
    N(va + "cwt_stats_layers.0.bias", (cfg.cwt_hidden,), 0.1)
    N(va + "cwt_stats_layers.2.weight", (cfg.cwt_hidden, cfg.cwt_hidden), 1.0 / math.sqrt(cfg.cwt_hidden))
    N(va + "cwt_stats_layers.2.bias", (cfg.cwt_hidden,), 0.1)
    N(va + "cwt_stats_layers.4.weight", (2, cfg.cwt_hidden), 0.02)
    sd[va + "cwt_stats_layers.4.bias"] = np.asarray([5.0, 0.4], np.float32)   # ln(f0)~5 -> ~150 Hz
    N(va + "pitch_embed.weight", (cfg.pitch_bins, H), H ** -0.5)
    sd[va + "pitch_embed.weight"][0] = 0.0
    sd[va + "energy_predictor.pos_embed_alpha"] = np.asarray([1.1], np.float32)
    predictor(va + "energy_predictor.", H, cfg.pred_layers, cfg.pred_kernel, 1, 0.12, [3.0])
    sd[va + "energy_predictor.embed_positions._float_tensor"] = np.zeros((1,), np.float32)
    N(va + "energy_embedding.weight", (cfg.energy_bins, H), H ** -0.5)
    sd[va + "energy_embedding.weight"][0] = 0.0

    if cfg.multi_speaker and getattr(cfg, "n_speaker", 0) > 0:
        N("duration_pitch_energy_net.speaker_emb.weight", (cfg.n_speaker, H), 1.0)      # nn.Embedding: N(0, 1), no bias
    elif cfg.multi_speaker:
        N("duration_pitch_energy_net.speaker_emb.weight", (H, cfg.external_speaker_dim),
          1.0 / math.sqrt(cfg.external_speaker_dim))
        N("duration_pitch_energy_net.speaker_emb.bias", (H,), 0.1)

    C = cfg.res_channels
    N("net.input_projection.0.conv.weight", (C, cfg.n_mels, 1), math.sqrt(2.0 / cfg.n_mels))
    N("net.input_projection.0.conv.bias", (C,), 0.1)
    N("net.mlp.0.linear.weight", (4 * C, C), 1.0 / math.sqrt(C))
    N("net.mlp.2.linear.weight", (C, 4 * C), 1.0 / math.sqrt(4 * C))
    for i in range(cfg.res_layers):
        p = f"net.residual_layers.{i}."
        N(p + "conv_layer.conv.weight", (2 * C, C, 3), 1.2 / math.sqrt(3 * C))
        N(p + "conv_layer.conv.bias", (2 * C,), 0.1)
        N(p + "diffusion_projection.linear.weight", (C, C), 0.5 / math.sqrt(C))
        if cfg.multi_speaker:
            N(p + "speaker_projection.linear.weight", (C, H), 0.5 / math.sqrt(H))
        N(p + "conditioner_projection.conv.weight", (C, H, 1), 0.7 / math.sqrt(H))
        N(p + "conditioner_projection.conv.bias", (C,), 0.1)
        N(p + "output_projection.conv.weight", (2 * C, C, 1), 2.5 / math.sqrt(C))
        N(p + "output_projection.conv.bias", (2 * C,), 0.1)
    N("net.skip_projection.conv.weight", (C, C, 1), 1.0 / math.sqrt(C))
    N("net.skip_projection.conv.bias", (C,), 0.1)
    N("net.output_projection.conv.weight", (cfg.n_mels, C, 1), 1.0 / math.sqrt(C))
    N("net.output_projection.conv.bias", (cfg.n_mels,), 0.1)
    return sd


def synth_hifigan_state_dict(hcfg: HifiGanConfig = HifiGanConfig(), seed: int = 0):
    """State dict of hifigan.Generator with weight norm already folded (plain weight/bias)."""
    sd = OrderedDict()

    def N(name, shape, std):
        sd[name] = _normal("hifigan." + name, seed, shape, std)

    c0 = hcfg.upsample_initial_channel
    N("conv_pre.weight", (c0, hcfg.num_mels, 7), 1.0 / math.sqrt(hcfg.num_mels * 7))
    N("conv_pre.bias", (c0,), 0.05)
    ch = c0
    for i, (u, k) in enumerate(zip(hcfg.upsample_rates, hcfg.upsample_kernel_sizes)):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        # ConvTranspose1d weight layout is [C_in, C_out, k]; each output sample sees k/u taps
        N(f"ups.{i}.weight", (cin, cout, k), 1.0 / math.sqrt(cin * k / u))
        N(f"ups.{i}.bias", (cout,), 0.05)
        for j, (rk, dils) in enumerate(zip(hcfg.resblock_kernel_sizes, hcfg.resblock_dilation_sizes)):
            r = i * len(hcfg.resblock_kernel_sizes) + j
            for grp in ("convs1", "convs2"):
                for m in range(len(dils)):
                    N(f"resblocks.{r}.{grp}.{m}.weight", (cout, cout, rk), 0.6 / math.sqrt(cout * rk))
                    N(f"resblocks.{r}.{grp}.{m}.bias", (cout,), 0.05)
        ch = cout
    N("conv_post.weight", (1, ch, 7), 0.35 / math.sqrt(ch * 7))
    N("conv_post.bias", (1,), 0.05)
    return sd


def fold_weight_norm(sd):
    """Fold ``weight_g``/``weight_v`` pairs (torch.nn.utils.weight_norm, dim=0) into plain
    ``weight`` tensors: w = g * v / ||v||, the norm taken over all dims but 0 — what
    Generator.remove_weight_norm() (hifigan/models.py:167-174) does before inference."""
    out = OrderedDict()
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len("_g")]
            vv = np.asarray(sd[base + "_v"], np.float32)
            g = np.asarray(v, np.float32)
            nrm = np.sqrt((vv.reshape(vv.shape[0], -1) ** 2).sum(1)).reshape((-1,) + (1,) * (vv.ndim - 1))
            out[base] = (g.reshape(nrm.shape) * vv / nrm).astype(np.float32)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = np.asarray(v, np.float32)
    return out
