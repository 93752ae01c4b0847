"""CPU oracle for the CM-TTS inference hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-numpy (fp32) restatement of the reference's algorithm for every row of SURVEY.md §8(a):
phoneme ids (+ speaker vector) -> FFT-block encoder -> variance adaptor / length regulator ->
T-step consistency denoiser -> HiFi-GAN generator -> int16 wav.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file; the
shipped path (``cm-tts_amd/``) never does and fails loudly if the HIP library is missing.

Parity pinning: the reference has no tests.  This oracle is pinned against golden
vectors produced by importing the reference's own modules in the build container
(tests/golden/make_golden.py, which needs /root/reference; the .npz fixtures it writes are
committed) — see tests/test_oracle_golden.py.

All tensors use the reference's layouts: activations [B, L, C] / [B, T, C], conv weights
[C_out, C_in, k], ConvTranspose1d weights [C_in, C_out, k], mel [B, T, 80].
Every function names the reference file:line it restates (paths relative to the reference root).
"""
import math

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32          # the working dtype of every function below; `precision("f64")` rebinds it


class precision:
    """``with precision("f64"):`` runs the same restatement in float64 (inputs are cast on entry by every function,
    fp32 weights promote).  The float64 result is the yardstick the tests measure fp32 / fp16 / bf16 errors against:
    the reference itself only ever computes in fp32, so its golden vectors carry fp32 rounding of their own."""

    def __init__(self, name):
        self.dtype = {"f32": np.float32, "f64": np.float64}[name]

    def __enter__(self):
        global F32
        self.prev = F32
        F32 = self.dtype
        _TAP_CACHE.clear()
        return self

    def __exit__(self, *exc):
        global F32
        F32 = self.prev
        _TAP_CACHE.clear()
        return False


_OPERAND16 = None          # None | "bf16" | "fp16": see operands16
_TEXT16 = False            # operands16(..., text=True): the library's opt-in "text16" (16-bit in- / out-projection and FFN contractions in the FFT blocks, 16-bit variance-predictor convs)


class operands16:
    """``with operands16("bf16"):`` restates the library's reduced-precision scheme (BASELINE.json configs[2]/[4];
    the reference has no such mode): the MFMA operands of the denoiser's residual-block convs (u, z, the conditioner input and the three
    weight sets) and of the HiFi-GAN ResBlock convs (leaky_relu(x), leaky_relu(xt), weights) are rounded to 16 bits
    (round-to-nearest-even from their fp32 value), products and sums stay in the working precision, and everything
    else (biases, gate, residual arithmetic, conv_pre / transposed convs / conv_post) is untouched."""

    def __init__(self, mode, text=False):
        assert mode in (None, "fp32", "bf16", "fp16", "fp16x3")
        self.mode = None if mode == "fp32" else mode
        self.text = bool(text) and self.mode in ("bf16", "fp16")    # cmtts_model_set_option(m, "text16", 1): bf16 / fp16 models only

    def __enter__(self):
        global _OPERAND16, _TEXT16
        self.prev = (_OPERAND16, _TEXT16)
        _OPERAND16, _TEXT16 = self.mode, self.text
        return self

    def __exit__(self, *exc):
        global _OPERAND16, _TEXT16
        _OPERAND16, _TEXT16 = self.prev
        return False


def quant16(a, mode=None):
    """Round to bf16 / fp16 (RNE, via the fp32 value like v_cvt_pk_{bf16,f16}_f32) and return in the working dtype."""
    mode = mode or _OPERAND16
    if mode is None:
        return a
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    if mode == "fp16":
        return a32.astype(np.float16).astype(F32)
    if mode == "fp16x3":      # hi = fp16(v), lo = fp16(v - hi): the operand the library carries is hi + lo (22 significant bits);
        hi = a32.astype(np.float16).astype(np.float32)          # its three-MFMA product omits only lo*lo (2^-22 relative)
        lo = (a32 - hi).astype(np.float16).astype(np.float32)
        return hi.astype(F32) + lo.astype(F32)
    u = a32.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(F32)


# ----------------------------------------------------------------------------- primitives

# Dense primitives can run on two CPU back ends with identical semantics: "numpy" (OpenBLAS GEMM per tap,
# the default and what the golden tests pin) and "torch" (torch.nn.functional on CPU tensors that alias
# the numpy arrays — oneDNN/MKL kernels, i.e. what the reference's own CPU path executes).  bench.py's
# cpu_baseline uses "torch" when available because it is the faster, fairer CPU number.
_BACKEND = "numpy"


def set_backend(name):
    global _BACKEND
    assert name in ("numpy", "torch")
    _BACKEND = name


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=F32))


_TAP_CACHE = {}


def _taps(w):
    """[Cout,Cin,K] -> contiguous [K,Cout,Cin] (BLAS-friendly per-tap matrices), cached per array."""
    key = (id(w), w.shape)
    hit = _TAP_CACHE.get(key)
    if hit is None or hit[0] is not w:
        if len(_TAP_CACHE) > 512:
            _TAP_CACHE.clear()
        hit = (w, np.ascontiguousarray(np.asarray(w, F32).transpose(2, 0, 1)))
        _TAP_CACHE[key] = hit
    return hit[1]


def conv1d(x, w, b=None, padding=0, dilation=1):
    """torch.nn.functional.conv1d, stride 1.  x [B,Cin,T], w [Cout,Cin,K] -> [B,Cout,T']."""
    x = np.asarray(x, F32)
    if _BACKEND == "torch":
        import torch
        with torch.no_grad():
            return torch.nn.functional.conv1d(_t(x), _t(w), None if b is None else _t(b), padding=padding,
                                              dilation=dilation).numpy()
    B, Cin, T = x.shape
    Cout, _, K = w.shape
    xp = np.zeros((B, Cin, T + 2 * padding), F32)
    xp[:, :, padding:padding + T] = x
    To = T + 2 * padding - dilation * (K - 1)
    y = np.zeros((B, Cout, To), F32)
    wt = _taps(w)
    for k in range(K):
        y += np.matmul(wt[k], np.ascontiguousarray(xp[:, :, k * dilation:k * dilation + To]))
    if b is not None:
        y += b[None, :, None]
    return y


def conv_transpose1d(x, w, b, stride, padding):
    """torch.nn.functional.conv_transpose1d.  x [B,Cin,T], w [Cin,Cout,K] -> [B,Cout,(T-1)s-2p+K]."""
    x = np.asarray(x, F32)
    if _BACKEND == "torch":
        import torch
        with torch.no_grad():
            return torch.nn.functional.conv_transpose1d(_t(x), _t(w), _t(b), stride=stride, padding=padding).numpy()
    B, Cin, T = x.shape
    _, Cout, K = w.shape
    full = np.zeros((B, Cout, (T - 1) * stride + K), F32)
    wt = _taps(w)                                    # [K, Cin, Cout]
    for k in range(K):
        full[:, :, k:k + (T - 1) * stride + 1:stride] += np.matmul(np.ascontiguousarray(wt[k].T), x)
    y = full[:, :, padding:full.shape[2] - padding]
    return y + b[None, :, None]


def linear(x, w, b=None):
    if _BACKEND == "torch":
        import torch
        with torch.no_grad():
            return torch.nn.functional.linear(_t(x), _t(w), None if b is None else _t(b)).numpy()
    y = np.matmul(np.asarray(x, F32), w.T)
    return y if b is None else y + b


def layer_norm(x, g, b, eps):
    """torch.nn.LayerNorm over the last axis (biased variance)."""
    m = x.mean(-1, keepdims=True, dtype=F32)
    v = ((x - m) ** 2).mean(-1, keepdims=True, dtype=F32)
    return ((x - m) / np.sqrt(v + F32(eps))).astype(F32) * g + b


def gelu_erf(x):
    return (x * F32(0.5) * (F32(1.0) + _erf(x * F32(math.sqrt(0.5))).astype(F32))).astype(F32)


def sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def mish(x):
    """model/blocks.py:621-623: x * tanh(softplus(x)); softplus with torch's threshold 20."""
    sp = np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, F32(20.0))))).astype(F32)
    return (x * np.tanh(sp)).astype(F32)


def leaky_relu(x, slope):
    return np.where(x > 0, x, x * F32(slope)).astype(F32)


def get_mask_from_lengths(lengths, max_len):
    """utils/tools.py:275-283 — True marks padding."""
    return np.arange(max_len)[None, :] >= np.asarray(lengths)[:, None]


def sinusoid_table(n_pos, dim):
    """model/blocks.py:45-62 SinusoidalPositionalEmbedding.get_embedding (padding row 0 zeroed):
    row p = [sin(p*w_j) | cos(p*w_j)], w_j = exp(-j*ln(1e4)/(dim/2-1)), halves concatenated."""
    half = dim // 2
    e = F32(math.log(10000) / (half - 1))
    w = np.exp(np.arange(half, dtype=F32) * -e).astype(F32)
    ang = (np.arange(n_pos, dtype=F32)[:, None] * w[None, :]).astype(F32)
    tab = np.concatenate([np.sin(ang), np.cos(ang)], 1).astype(F32)
    tab[0, :] = 0
    return tab


def make_positions(nonpad):
    """utils/tools.py:810-822 with padding_idx 0: cumsum(nonpad)*nonpad (1-based, pad -> 0)."""
    nonpad = nonpad.astype(np.int64)
    return np.cumsum(nonpad, 1) * nonpad


def positional_embedding(ref_values, dim):
    """SinusoidalPositionalEmbedding.forward (model/blocks.py:64-81): positions from `value != 0`."""
    pos = make_positions(ref_values != 0)
    tab = sinusoid_table(int(pos.max()) + 2, dim)
    return tab[pos]


# ----------------------------------------------------------------------------- text encoder

def encoder_embedding(sd, texts, hidden=256):
    """FastspeechEncoder.forward_embedding, model/modules.py:145-151: sqrt(H)*E[tok] + PE[pos]."""
    E = sd["duration_pitch_energy_net.text_encoder.embed_tokens.weight"]
    x = F32(math.sqrt(hidden)) * E[texts]
    return (x + positional_embedding(texts, hidden)).astype(F32)


def multihead_self_attention(x, in_w, out_w, key_pad, n_heads):
    """model/blocks.py:266-312 -> F.multi_head_attention_forward, no biases (bias=False, :585).
    x [B,L,C]; heads are contiguous channel slices; padded keys get -inf before softmax."""
    B, L, C = x.shape
    d = C // n_heads
    qt = quant16 if _TEXT16 else (lambda a: a)              # "text16": in- and out-projection operands in 16 bits (scores, softmax, P V stay as they are)
    qkv = linear(qt(x.astype(F32)), qt(in_w))              # [B,L,3C]
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    q = q * F32(1.0 / math.sqrt(d))

    def heads(t):
        return t.reshape(B, L, n_heads, d).transpose(0, 2, 1, 3)   # [B,H,L,d]

    s = np.matmul(heads(q), heads(k).transpose(0, 1, 3, 2))        # [B,H,L,L]
    s = np.where(key_pad[:, None, None, :], F32(-np.inf), s)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = (p / p.sum(-1, keepdims=True, dtype=F32)).astype(F32)
    o = np.matmul(p, heads(v)).transpose(0, 2, 1, 3).reshape(B, L, C)
    return linear(qt(o.astype(F32)), qt(out_w))


def enc_sa_layer(sd, prefix, x, pad_mask, n_heads, kernel):
    """EncSALayer.forward model/blocks.py:594-618 + TransformerFFNLayer :539-552 (eval mode)."""
    keep = (~pad_mask).astype(F32)[:, :, None]
    h = layer_norm(x, sd[prefix + "layer_norm1.weight"], sd[prefix + "layer_norm1.bias"], 1e-12)
    h = multihead_self_attention(h, sd[prefix + "self_attn.in_proj_weight"],
                                 sd[prefix + "self_attn.out_proj.weight"], pad_mask, n_heads)
    x = (x + h) * keep
    h = layer_norm(x, sd[prefix + "layer_norm2.weight"], sd[prefix + "layer_norm2.bias"], 1e-12)
    qt = quant16 if _TEXT16 else (lambda a: a)          # "text16": the two FFN contractions take 16-bit operands (csrc/cmtts_api.hip fft_stack)
    h = conv1d(qt(h.transpose(0, 2, 1).astype(F32)), qt(sd[prefix + "ffn.ffn_1.weight"]), sd[prefix + "ffn.ffn_1.bias"],
               padding=kernel // 2)
    h = gelu_erf(h * F32(kernel ** -0.5)).transpose(0, 2, 1)
    h = linear(qt(h.astype(F32)), qt(sd[prefix + "ffn.ffn_2.weight"]), sd[prefix + "ffn.ffn_2.bias"])
    return ((x + h) * keep).astype(F32)


def text_encoder(sd, cfg, texts, src_mask):
    """FastspeechEncoder.forward / FFTBlocks.forward, model/modules.py:80-105,133-143."""
    pre = "duration_pitch_energy_net.text_encoder."
    keep = (~src_mask).astype(F32)[:, :, None]
    x = encoder_embedding(sd, texts, cfg.hidden) * keep
    for i in range(cfg.enc_layers):
        x = enc_sa_layer(sd, f"{pre}layers.{i}.op.", x, src_mask, cfg.enc_heads, cfg.ffn_kernel) * keep
    x = layer_norm(x, sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], 1e-5) * keep
    return x.astype(F32)


def fastspeech_decoder(sd, cfg, x, pad_mask=None, prefix="decoder.", n_layers=4):
    """FastspeechDecoder.forward = FFTBlocks.forward with use_pos_embed=True, model/modules.py:80-105,154-165:
    pad = all-zero rows unless given; x + alpha*PE[positions(x[..., 0] != 0)]; masked; FFT blocks; LayerNorm 1e-5."""
    x = np.asarray(x, F32)
    if pad_mask is None:
        pad_mask = np.abs(x).sum(-1) == 0
    keep = (~pad_mask).astype(F32)[:, :, None]
    x = (x + sd[prefix + "pos_embed_alpha"].astype(F32) * positional_embedding(x[..., 0], x.shape[-1])).astype(F32)
    x = x * keep
    for i in range(n_layers):
        x = enc_sa_layer(sd, f"{prefix}layers.{i}.op.", x, pad_mask, cfg.enc_heads, cfg.ffn_kernel) * keep
    x = layer_norm(x, sd[prefix + "layer_norm.weight"], sd[prefix + "layer_norm.bias"], 1e-5) * keep
    return x.astype(F32)


# ----------------------------------------------------------------------------- variance adaptor

def _pred_convs(sd, prefix, xs, n_layers, kernel, mask=None):
    """conv stack shared by Duration/Pitch/Energy predictors (model/modules.py:477-487,527-537):
    zero-pad, Conv1d, ReLU, LayerNorm over channels (eps 1e-12); duration also masks per layer."""
    h = xs.transpose(0, 2, 1)
    for li in range(n_layers):
        qt = quant16 if _TEXT16 else (lambda a: a)          # "text16": the predictor convs' operands in 16 bits too
        h = conv1d(qt(h.astype(F32)), qt(sd[f"{prefix}conv.{li}.1.weight"]), sd[f"{prefix}conv.{li}.1.bias"],
                   padding=(kernel - 1) // 2)
        h = np.maximum(h, 0)
        h = layer_norm(h.transpose(0, 2, 1), sd[f"{prefix}conv.{li}.3.weight"],
                       sd[f"{prefix}conv.{li}.3.bias"], 1e-12).transpose(0, 2, 1)
        if mask is not None:
            h = h * (~mask).astype(F32)[:, None, :]
    return h.transpose(0, 2, 1).astype(F32)


def duration_predictor(sd, cfg, x, src_mask):
    """DurationPredictor.forward model/modules.py:498-509 -> log-durations [B,L]."""
    p = "duration_pitch_energy_net.variance_adaptor.duration_predictor."
    h = _pred_convs(sd, p, x, cfg.dur_layers, cfg.dur_kernel, src_mask)
    y = linear(h, sd[p + "linear.weight"], sd[p + "linear.bias"])
    return (y * (~src_mask).astype(F32)[:, :, None])[..., 0]


def pitch_style_predictor(sd, prefix, cfg, xs):
    """PitchPredictor/EnergyPredictor.forward model/modules.py:542-556: adds alpha*PE at positions
    counted from `xs[...,0] != 0` (a float-zero test), conv stack without masking, Linear."""
    pe = positional_embedding(xs[..., 0], xs.shape[-1])
    xs = (xs + sd[prefix + "pos_embed_alpha"] * pe).astype(F32)
    h = _pred_convs(sd, prefix, xs, cfg.pred_layers, cfg.pred_kernel, None)
    return linear(h, sd[prefix + "linear.weight"], sd[prefix + "linear.bias"])


def bucketize(v, bins):
    """torch.bucketize(right=False): first i with bins[i] >= v (== numpy side='left')."""
    return np.searchsorted(bins, v, side="left").astype(np.int64)


def durations_from_log(log_d, d_control=1.0):
    """model/modules.py:369-372: clamp(round(exp(log_d)-1)*d_control, min=0); half-to-even round."""
    return np.maximum(np.rint(np.exp(log_d.astype(F32)) - F32(1.0)) * F32(d_control), 0).astype(F32)


def dur_to_mel2ph(dur, src_mask, T=None):
    """utils/tools.py:768-798: 1-based phoneme id per frame, 0 for padding (int64 [B,T])."""
    d = np.rint(dur).astype(np.int64) * (~src_mask).astype(np.int64)
    cum = np.cumsum(d, 1)
    width = int(cum[:, -1].max()) if T is None else T
    t = np.arange(width)[None, :]
    # number of phonemes whose cumulative end is <= t  -> index of the covering phoneme
    idx = (cum[:, None, :] <= t[:, :, None]).sum(-1)
    return np.where(t < cum[:, -1:], idx + 1, 0).astype(np.int64)


def length_regulate(x, dur, max_len=None):
    """LengthRegulator model/modules.py:421-448 + pad utils/tools.py:724-742: repeat phoneme i
    int(d_i) times, zero-pad (or truncate) to max_len.  Returns ([B,T,C], mel_len int64[B])."""
    B, L, C = x.shape
    d = np.maximum(dur.astype(np.int64), 0)
    mel_len = d.sum(1)
    T = int(mel_len.max()) if max_len is None else int(max_len)
    out = np.zeros((B, T, C), F32)
    for b in range(B):
        rep = np.repeat(x[b], d[b], axis=0)[:T]
        out[b, :rep.shape[0]] = rep
    return out, mel_len.astype(np.int64)


_F0_MEL_MIN = 1127 * np.log(1 + 50.0 / 700)     # utils/pitch_tools.py:19-23
_F0_MEL_MAX = 1127 * np.log(1 + 1100.0 / 700)


def f0_to_coarse(f0, f0_bin=256):
    """utils/pitch_tools.py:26-35 (torch branch): mel-scale bucket in [1, 255], (mel+0.5) truncated."""
    mel = (F32(1127) * np.log(F32(1) + f0 / F32(700))).astype(F32)
    pos = mel > 0
    mel = np.where(pos, (mel - F32(_F0_MEL_MIN)) * F32(f0_bin - 2) / F32(_F0_MEL_MAX - _F0_MEL_MIN) + F32(1), mel)
    mel = np.where(mel <= 1, F32(1), mel)
    mel = np.where(mel > f0_bin - 1, F32(f0_bin - 1), mel).astype(F32)
    return (mel + F32(0.5)).astype(np.int64), mel


def cwt_to_pitch_index(cwt_out, mean, std, cfg):
    """VarianceAdaptor.get_pitch_embedding cwt branch, model/modules.py:274-300, with
    cwt2f0_norm/inverse_cwt_torch/norm_f0/denorm_f0 (utils/pitch_tools.py:244-279,38-47,64-78).
    cwt_out [B,T,10|11], mean/std [B] (std already * cwt_std_scale)."""
    spec = cwt_out[:, :, :10]
    scale = ((np.arange(10, dtype=F32) + F32(1) + F32(2.5)) ** F32(-2.5)).astype(F32)
    r = (spec * scale[None, None, :]).sum(-1, dtype=F32)
    r = ((r - r.mean(-1, keepdims=True, dtype=F32)) / r.std(-1, ddof=1, keepdims=True, dtype=F32)).astype(F32)
    f0 = np.exp(r * std[:, None] + mean[:, None]).astype(F32)
    f0n = np.log2(f0 + F32(cfg.pitch_norm_eps)).astype(F32)          # norm_f0 'log'
    f0d = np.exp2(f0n).astype(F32)                                    # denorm_f0: 2 ** f0
    if cfg.use_uv:
        uv = cwt_out[:, :, -1] > 0
        f0d = np.where(uv, F32(0), f0d)
    idx, mel = f0_to_coarse(f0d)
    return idx, f0d, mel


def variance_adaptor(sd, cfg, enc_out, src_mask, speaker_emb=None, max_len=None, p_control=1.0, e_control=1.0,
                     d_control=1.0, d_target=None, e_target=None, pitch_target=None):
    """VarianceAdaptor.forward, model/modules.py:331-412.  Defaults = the inference branch (targets None,
    controls 1.0).  Teacher-forced branches: d_target [B,L] (:365-367), e_target [B,L] (get_energy_embedding
    :318-328), pitch_target = dict(cwt_spec [B,T,10], f0_mean [B], f0_std [B], uv bool [B,T]) (:379-390 with
    cwt2f0_norm utils/pitch_tools.py:268-273: target statistics, no cwt_std_scale; the predictors still run
    and their outputs are returned).  Controls multiply the predictor outputs (:270,326,369).
    Returns a dict of every intermediate the parity tests pin."""
    va = "duration_pitch_energy_net.variance_adaptor."
    x = enc_out
    if speaker_emb is not None:
        x = (x + speaker_emb[:, None, :]).astype(F32)
    log_d = duration_predictor(sd, cfg, x, src_mask)
    e_pred = pitch_style_predictor(sd, va + "energy_predictor.", cfg, x)[..., 0]
    if e_target is not None:
        e_idx = bucketize(np.asarray(e_target, F32), sd[va + "energy_bins"])
    else:
        e_pred = (e_pred * F32(e_control)).astype(F32)
        e_idx = bucketize(e_pred, sd[va + "energy_bins"])
    out1 = (x + sd[va + "energy_embedding.weight"][e_idx]).astype(F32)
    d_rounded = durations_from_log(log_d, d_control) if d_target is None else np.asarray(d_target, F32)
    x_lr, mel_len = length_regulate(out1, d_rounded, max_len)
    T = x_lr.shape[1]
    mel2ph = dur_to_mel2ph(d_rounded, src_mask)
    h = linear(x_lr, sd[va + "cwt_predictor.0.weight"], sd[va + "cwt_predictor.0.bias"])
    cwt_out = (pitch_style_predictor(sd, va + "cwt_predictor.1.", cfg, h) * F32(p_control)).astype(F32)
    s = np.maximum(linear(out1[:, 0, :], sd[va + "cwt_stats_layers.0.weight"], sd[va + "cwt_stats_layers.0.bias"]), 0)
    s = np.maximum(linear(s, sd[va + "cwt_stats_layers.2.weight"], sd[va + "cwt_stats_layers.2.bias"]), 0)
    s = linear(s, sd[va + "cwt_stats_layers.4.weight"], sd[va + "cwt_stats_layers.4.bias"])
    mean, std_raw = s[:, 0], s[:, 1]                                  # f0_mean / f0_std as returned
    if pitch_target is None:
        p_idx, f0d, f0_mel = cwt_to_pitch_index(cwt_out, mean, (std_raw * F32(cfg.cwt_std_scale)).astype(F32), cfg)
    else:
        spec = np.asarray(pitch_target["cwt_spec"], F32)
        if cfg.use_uv:      # the target's uv mask replaces the predicted uv logit
            spec = np.concatenate([spec, np.where(pitch_target["uv"], F32(1), F32(-1))[..., None].astype(F32)], -1)
        p_idx, f0d, f0_mel = cwt_to_pitch_index(spec, np.asarray(pitch_target["f0_mean"], F32),
                                                np.asarray(pitch_target["f0_std"], F32), cfg)
    cond = (x_lr + sd[va + "pitch_embed.weight"][p_idx]).astype(F32)
    return dict(cond=cond, log_d=log_d, d_rounded=d_rounded, mel_len=mel_len, mel2ph=mel2ph,
                e_pred=e_pred, e_idx=e_idx, out1=out1, x_lr=x_lr, cwt_out=cwt_out,
                f0_mean=mean, f0_std=std_raw, f0_denorm=f0d, f0_mel=f0_mel, p_idx=p_idx,
                mel_mask=get_mask_from_lengths(mel_len, T))


def duration_pitch_speaker_net(sd, cfg, texts, src_lens, spker_embeds=None, max_mel_len=None, speakers=None, **va_kwargs):
    """DurationPitchSpeakerNet.forward model/cmtts.py:44-122 (va_kwargs: controls / targets of
    variance_adaptor).  Multi-speaker: Linear(512 -> 256) of the external vector (:39-42,81), or — preprocess.yaml
    speaker_embedder "none", cfg.n_speaker > 0 — the row of the nn.Embedding table that `speakers` names (:26-38,78)."""
    B, L = texts.shape
    src_mask = get_mask_from_lengths(src_lens, L)
    enc = text_encoder(sd, cfg, texts, src_mask)
    spk = None
    if cfg.multi_speaker and getattr(cfg, "n_speaker", 0) > 0:
        spk = np.asarray(sd["duration_pitch_energy_net.speaker_emb.weight"], F32)[np.asarray(speakers, np.int64)]
    elif cfg.multi_speaker:
        spk = linear(spker_embeds, sd["duration_pitch_energy_net.speaker_emb.weight"],
                     sd["duration_pitch_energy_net.speaker_emb.bias"]).astype(F32)
    out = variance_adaptor(sd, cfg, enc, src_mask, spk, max_mel_len, **va_kwargs)
    out.update(enc_out=enc, speaker_emb=spk, src_mask=src_mask)
    return out


# ----------------------------------------------------------------------------- denoiser + sampler

def diffusion_embedding(t, dim):
    """DiffusionEmbedding.forward model/blocks.py:633-640."""
    half = dim // 2
    e = F32(math.log(10000) / (half - 1))
    w = np.exp(np.arange(half, dtype=F32) * -e).astype(F32)
    a = (t.astype(F32)[:, None] * w[None, :]).astype(F32)
    return np.concatenate([np.sin(a), np.cos(a)], -1).astype(F32)


def denoiser_forward(sd, cfg, x, t, cond, speaker_emb):
    """CMDenoiserTTS.forward (model/cm_tool/tts_net.py:29-37) = Denoiser.forward
    (model/modules.py:600-639) + ResidualBlock.forward (model/blocks.py:667-686).
    x [B,1,T,80] (already scaled by c_in), t [B] (= 250*ln sigma), cond [B,T,256],
    speaker_emb [B,256] or None  ->  [B,1,T,80]."""
    C = cfg.res_channels
    h = x[:, 0].transpose(0, 2, 1)                                   # [B,80,T]
    h = np.maximum(conv1d(h, sd["net.input_projection.0.conv.weight"], sd["net.input_projection.0.conv.bias"]), 0)
    e = diffusion_embedding(t, C)
    e = linear(mish(linear(e, sd["net.mlp.0.linear.weight"])), sd["net.mlp.2.linear.weight"])
    c = cond.transpose(0, 2, 1)
    skip_sum = None
    for i in range(cfg.res_layers):
        p = f"net.residual_layers.{i}."
        d = linear(e, sd[p + "diffusion_projection.linear.weight"])[:, :, None]
        # 16-bit modes (round 3): the conditioner projection takes 16-bit operands like the block's other two contractions (csrc/cond_gemm16.hip)
        cp = conv1d(quant16(c.astype(F32)), quant16(sd[p + "conditioner_projection.conv.weight"]), sd[p + "conditioner_projection.conv.bias"])
        r = (h + d).astype(F32)
        u = r + cp
        if cfg.multi_speaker:
            u = u + linear(speaker_emb, sd[p + "speaker_projection.linear.weight"])[:, :, None]
        y = conv1d(quant16(u.astype(F32)), quant16(sd[p + "conv_layer.conv.weight"]), sd[p + "conv_layer.conv.bias"], padding=1)
        z = (sigmoid(y[:, :C]) * np.tanh(y[:, C:])).astype(F32)
        o = conv1d(quant16(z), quant16(sd[p + "output_projection.conv.weight"]), sd[p + "output_projection.conv.bias"])
        h = ((o[:, :C] + r) / F32(math.sqrt(2.0))).astype(F32)
        skip_sum = o[:, C:] if skip_sum is None else skip_sum + o[:, C:]
    s = (skip_sum / F32(math.sqrt(cfg.res_layers))).astype(F32)
    s = np.maximum(conv1d(s, sd["net.skip_projection.conv.weight"], sd["net.skip_projection.conv.bias"]), 0)
    out = conv1d(s, sd["net.output_projection.conv.weight"], sd["net.output_projection.conv.bias"])
    return out.transpose(0, 2, 1)[:, None].astype(F32)


def boundary_scalings(sigma, cfg):
    """KarrasDenoiser.get_scalings_for_boundary_condition, karras_diffusion.py:87-102."""
    sd2 = cfg.sigma_data ** 2
    c_skip = sd2 / ((sigma - cfg.sigma_min) ** 2 + sd2)
    c_out = (sigma - cfg.sigma_min) * cfg.sigma_data / (sigma ** 2 + sd2) ** 0.5
    c_in = 1 / (sigma ** 2 + sd2) ** 0.5
    return c_skip, c_out, c_in


def edm_scalings(sigma, cfg):
    """KarrasDenoiser.get_scalings, karras_diffusion.py:81-85 (distillation=False: no sigma_min shift)."""
    sd2 = cfg.sigma_data ** 2
    c_skip = sd2 / (sigma ** 2 + sd2)
    c_out = sigma * cfg.sigma_data / (sigma ** 2 + sd2) ** 0.5
    c_in = 1 / (sigma ** 2 + sd2) ** 0.5
    return c_skip, c_out, c_in


def karras_denoise(sd, cfg, x_t, sigma, cond, speaker_emb, distillation=True):
    """KarrasDenoiser.denoise karras_diffusion.py:392-407: boundary-condition scalings when distillation (what
    synthesize.py:59-64 selects for consistency models), get_scalings otherwise (progdist teachers).
    sigma: fp32 [B].  Returns the denoised sample [B,1,T,80]."""
    sigma = np.asarray(sigma, F32)
    scal = boundary_scalings if distillation else edm_scalings
    c_skip, c_out, c_in = [np.asarray(v, F32)[:, None, None, None] for v in scal(sigma, cfg)]
    t = (F32(1000 * 0.25) * np.log(sigma + F32(1e-44))).astype(F32)
    f = denoiser_forward(sd, cfg, (c_in * x_t).astype(F32), t, cond, speaker_emb)
    return (c_out * f + c_skip * x_t).astype(F32)


def multistep_schedule(n_steps, cfg, ts=None, steps=2):
    """synthesize.py:111-147 + stochastic_iterative_sampler karras_diffusion.py:830-854.
    T=1 -> onestep at sigma_max.  T=2/4 -> ts=(0,)*T+(1,), steps=2: every evaluation happens at
    sigma_max; the re-noising std after evaluation i is sqrt(next_t^2 - sigma_min^2)*0.85 with
    next_t = sigma_max except after the last evaluation, where next_t = sigma_min -> std 0.
    An explicit `ts` (with its `steps`) gives the general stochastic_iterative_sampler schedule instead.
    Returns (eval_sigmas float64[n], renoise_std float64[n]); onestep has renoise None."""
    if ts is None:
        if n_steps == 1:
            return [cfg.sigma_max], [None]
        ts = (0,) * n_steps + (1,)
        steps = 2
    tmax, tmin = cfg.sigma_max ** (1 / cfg.rho), cfg.sigma_min ** (1 / cfg.rho)
    sig, std = [], []
    for i in range(len(ts) - 1):
        t = (tmax + ts[i] / (steps - 1) * (tmin - tmax)) ** cfg.rho
        nt = (tmax + ts[i + 1] / (steps - 1) * (tmin - tmax)) ** cfg.rho
        nt = float(np.clip(nt, cfg.sigma_min, cfg.sigma_max))
        sig.append(t)
        std.append(float(np.sqrt(nt ** 2 - cfg.sigma_min ** 2) * 0.85))
    return sig, std


def karras_sample_tts(sd, cfg, cond, speaker_emb, n_steps, noise, ts=None, steps=2):
    """karras_sample_tts karras_diffusion.py:480-577 with explicit noise (the reference draws
    x_T then one randn_like per multistep iteration, random_util.py:17-25).
    noise: list of [B,1,T,80] N(0,1) arrays — noise[0] -> x_T, noise[1+i] -> re-noise after eval i.
    Returns mel [B,T,80]."""
    B = cond.shape[0]
    sig, std = multistep_schedule(n_steps, cfg, ts, steps)
    x = (noise[0] * F32(cfg.sigma_max)).astype(F32)
    for i, s in enumerate(sig):
        x0 = karras_denoise(sd, cfg, x, np.full((B,), s, F32), cond, speaker_emb)
        if std[i] is None:
            x = x0
        else:
            # randn_like(x) * np.sqrt(next_t**2 - t_min**2) * 0.85: two fp32 multiplies (karras_diffusion.py:852)
            x = (x0 + (noise[1 + i] * F32(std[i] / 0.85)).astype(F32) * F32(0.85)).astype(F32)
    return x[:, 0]


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0):
    """karras_diffusion.py:580-586: the Karras et al. (2022) schedule, n sigmas + a trailing 0, fp32 like
    th.linspace."""
    ramp = np.linspace(0, 1, n, dtype=F32)
    lo, hi = F32(sigma_min ** (1 / rho)), F32(sigma_max ** (1 / rho))
    sig = ((hi + ramp * (lo - hi)).astype(F32) ** F32(rho)).astype(F32)
    return np.concatenate([sig, np.zeros(1, F32)])


def ode_samplers(denoiser, x, sigmas, noise, sampler, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0):
    """The ODE/SDE sampler loops of karras_diffusion.py around any denoiser(x, sigma[B]) -> x0:
    "euler" :743-771, "heun" :693-739, "dpm" :775-820, "ancestral" :605-632.  `noise` is the list of
    N(0,1) draws handed out in call order (heun/dpm draw one per iteration even with s_churn = 0;
    ancestral adds one per iteration).  fp32 throughout, like the reference's tensors."""
    x = x.astype(F32)
    s_in = np.ones((x.shape[0],), F32)
    n = len(sigmas) - 1
    draws = iter(noise)
    to_d = lambda xx, sg, den: ((xx - den) / F32(sg)).astype(F32)
    for i in range(n):
        if sampler == "euler":
            d = to_d(x, sigmas[i], denoiser(x, sigmas[i] * s_in))
            x = (x + d * F32(sigmas[i + 1] - sigmas[i])).astype(F32)
        elif sampler == "ancestral":
            den = denoiser(x, sigmas[i] * s_in)
            sf, st = F32(sigmas[i]), F32(sigmas[i + 1])
            up = F32((st ** 2 * (sf ** 2 - st ** 2) / sf ** 2) ** F32(0.5))
            down = F32((st ** 2 - up ** 2) ** F32(0.5))
            d = to_d(x, sf, den)
            x = (x + d * F32(down - sf)).astype(F32)
            x = (x + next(draws).astype(F32) * up).astype(F32)
        elif sampler in ("heun", "dpm"):
            gamma = min(s_churn / n, 2 ** 0.5 - 1) if s_tmin <= sigmas[i] <= s_tmax else 0.0
            eps = (next(draws).astype(F32) * F32(s_noise)).astype(F32)
            sigma_hat = F32(sigmas[i] * F32(gamma + 1))
            if gamma > 0:
                x = (x + eps * F32((sigma_hat ** 2 - sigmas[i] ** 2) ** F32(0.5))).astype(F32)
            d = to_d(x, sigma_hat, denoiser(x, sigma_hat * s_in))
            if sampler == "heun":
                dt = F32(sigmas[i + 1] - sigma_hat)
                if sigmas[i + 1] == 0:
                    x = (x + d * dt).astype(F32)
                else:
                    x2 = (x + d * dt).astype(F32)
                    d2 = to_d(x2, sigmas[i + 1], denoiser(x2, sigmas[i + 1] * s_in))
                    x = (x + ((d + d2) / F32(2)).astype(F32) * dt).astype(F32)
            else:
                third = F32(1.0 / 3.0)
                mid = F32(((sigma_hat ** third + F32(sigmas[i + 1]) ** third) / F32(2)) ** F32(3))
                x2 = (x + d * F32(mid - sigma_hat)).astype(F32)
                d2 = to_d(x2, mid, denoiser(x2, mid * s_in))
                x = (x + d2 * F32(sigmas[i + 1] - sigma_hat)).astype(F32)
        else:
            raise ValueError(sampler)
    return x


def karras_sample_tts_ode(sd, cfg, cond, speaker_emb, sampler, steps, noise, **kw):
    """karras_sample_tts (karras_diffusion.py:480-577) with sampler in {"euler", "heun", "dpm", "ancestral"}:
    sigmas = get_sigmas_karras(steps), x_T = noise[0] * sigma_max, remaining draws feed the loop.
    Returns mel [B,T,80]."""
    distillation = kw.pop("distillation", True)
    sig = get_sigmas_karras(steps, cfg.sigma_min, cfg.sigma_max, cfg.rho)
    den = lambda x, s: karras_denoise(sd, cfg, x, s, cond, speaker_emb, distillation)
    x = (noise[0] * F32(cfg.sigma_max)).astype(F32)
    return ode_samplers(den, x, sig, noise[1:], sampler, **kw)[:, 0]


def karras_sample_tts_torch(sd, cfg, cond, speaker_emb, n_steps, noise):
    """Same computation as karras_sample_tts/denoiser_forward, written end-to-end in stock torch CPU ops
    (multi-threaded conv/GEMM *and* element-wise kernels) — the layer graph a PyTorch-CPU run of the
    reference executes.  Used for the timed cpu_baseline and cross-checked against the numpy path in
    tests/test_oracle_golden.py; the weights are converted once per call."""
    import torch
    import torch.nn.functional as Fn
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=F32))
    C = cfg.res_channels
    W = {k: tt(v) for k, v in sd.items() if k.startswith("net.")}
    c = tt(cond).transpose(1, 2).contiguous()
    spk = None if speaker_emb is None else tt(speaker_emb)
    B = c.shape[0]
    half = C // 2
    omega = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    sig, std = multistep_schedule(n_steps, cfg)

    def denoise(x_bt, t):
        h = Fn.relu(Fn.conv1d(x_bt[:, 0].transpose(1, 2), W["net.input_projection.0.conv.weight"],
                              W["net.input_projection.0.conv.bias"]))
        a = t[:, None] * omega[None, :]
        e = torch.cat([a.sin(), a.cos()], -1)
        e = Fn.linear(e, W["net.mlp.0.linear.weight"])
        e = Fn.linear(e * torch.tanh(Fn.softplus(e)), W["net.mlp.2.linear.weight"])
        skip = None
        for i in range(cfg.res_layers):
            p = f"net.residual_layers.{i}."
            d = Fn.linear(e, W[p + "diffusion_projection.linear.weight"])[:, :, None]
            cp = Fn.conv1d(c, W[p + "conditioner_projection.conv.weight"], W[p + "conditioner_projection.conv.bias"])
            r = h + d
            u = r + cp
            if cfg.multi_speaker:
                u = u + Fn.linear(spk, W[p + "speaker_projection.linear.weight"])[:, :, None]
            y = Fn.conv1d(u, W[p + "conv_layer.conv.weight"], W[p + "conv_layer.conv.bias"], padding=1)
            z = torch.sigmoid(y[:, :C]) * torch.tanh(y[:, C:])
            o = Fn.conv1d(z, W[p + "output_projection.conv.weight"], W[p + "output_projection.conv.bias"])
            h = (o[:, :C] + r) / math.sqrt(2.0)
            skip = o[:, C:] if skip is None else skip + o[:, C:]
        s_ = skip / math.sqrt(cfg.res_layers)
        s_ = Fn.relu(Fn.conv1d(s_, W["net.skip_projection.conv.weight"], W["net.skip_projection.conv.bias"]))
        out = Fn.conv1d(s_, W["net.output_projection.conv.weight"], W["net.output_projection.conv.bias"])
        return out.transpose(1, 2)[:, None]

    with torch.no_grad():
        x = tt(noise[0]) * cfg.sigma_max
        for i, sg in enumerate(sig):
            sigma = torch.full((B,), sg, dtype=torch.float32)
            c_skip, c_out, c_in = [v[:, None, None, None] for v in boundary_scalings(sigma, cfg)]
            f = denoise(c_in * x, 1000 * 0.25 * torch.log(sigma + 1e-44))
            x0 = c_out * f + c_skip * x
            x = x0 if std[i] is None else x0 + tt(noise[1 + i]) * (std[i] / 0.85) * 0.85
    return x[:, 0].numpy()


# ----------------------------------------------------------------------------- HiFi-GAN

def hifigan_generator(hsd, hcfg, mel_ct):
    """hifigan.Generator.forward hifigan/models.py:149-165, ResBlock.forward :96-103.
    mel_ct [B,80,T] -> wav [B,1,256*T] in (-1,1)."""
    x = conv1d(mel_ct, hsd["conv_pre.weight"], hsd["conv_pre.bias"], padding=3)
    nk = len(hcfg.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(hcfg.upsample_rates, hcfg.upsample_kernel_sizes)):
        x = leaky_relu(x, hcfg.lrelu_slope)
        # bf16 / fp16 vocoder modes carry the upsamplers' operands in 16 bits too (convT_xl16_kernel); fp16x3 keeps them fp32
        qu = quant16 if _OPERAND16 in ("bf16", "fp16") else (lambda a: a)
        x = conv_transpose1d(qu(x), qu(hsd[f"ups.{i}.weight"]), hsd[f"ups.{i}.bias"], u, (k - u) // 2)
        xs = None
        for j, (rk, dils) in enumerate(zip(hcfg.resblock_kernel_sizes, hcfg.resblock_dilation_sizes)):
            r = i * nk + j
            xr = x
            for m, dil in enumerate(dils):
                xt = quant16(leaky_relu(xr, hcfg.lrelu_slope))
                xt = conv1d(xt, quant16(hsd[f"resblocks.{r}.convs1.{m}.weight"]), hsd[f"resblocks.{r}.convs1.{m}.bias"],
                            padding=(rk * dil - dil) // 2, dilation=dil)
                xt = quant16(leaky_relu(xt, hcfg.lrelu_slope))
                xt = conv1d(xt, quant16(hsd[f"resblocks.{r}.convs2.{m}.weight"]), hsd[f"resblocks.{r}.convs2.{m}.bias"],
                            padding=(rk - 1) // 2)
                xr = (xt + xr).astype(F32)
            xs = xr if xs is None else xs + xr
        x = (xs / F32(nk)).astype(F32)
    x = leaky_relu(x, hcfg.final_lrelu_slope)
    x = conv1d(x, hsd["conv_post.weight"], hsd["conv_post.bias"], padding=3)
    return np.tanh(x).astype(F32)


def wav_to_int16(wav, max_wav_value=32768.0):
    """vocoder_infer utils/model.py:195-198: (wav*32768).astype('int16') — truncation toward zero;
    +1.0 maps to 32768 which wraps to -32768 (C cast via int32 on x86)."""
    v = (wav * F32(max_wav_value)).astype(F32)
    return v.astype(np.int32).astype(np.int16)


def vocoder_infer(hsd, hcfg, mel_btc, mel_lens, cfg):
    """vocoder_infer utils/model.py:187-205 as called from synth_samples utils/tools.py:595-600:
    mel [B,T,80] -> list of int16 arrays trimmed to mel_len*hop."""
    wav = hifigan_generator(hsd, hcfg, mel_btc.transpose(0, 2, 1))[:, 0]
    pcm = wav_to_int16(wav, cfg.max_wav_value)
    return [pcm[i, : int(mel_lens[i]) * cfg.hop_length] for i in range(len(pcm))], wav


# ----------------------------------------------------------------------------- end to end

def synthesize(sd, cfg, texts, src_lens, spker_embeds, n_steps, noise, max_mel_len=None, torch_sampler=False):
    """CMTotalTTSSynthesize.synthesize synthesize.py:88-153: duration net once (it is bit-identical
    to re-running it every step with mels=x), then the T-step sampler.
    Returns (mel [B,T,80], mel_lens, stage dict)."""
    st = duration_pitch_speaker_net(sd, cfg, texts, src_lens, spker_embeds, max_mel_len)
    sampler = karras_sample_tts_torch if torch_sampler else karras_sample_tts
    mel = sampler(sd, cfg, st["cond"], st["speaker_emb"], n_steps, noise)
    return mel, st["mel_len"], st
