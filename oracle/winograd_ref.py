"""TEST INFRASTRUCTURE (like everything under oracle/): a numpy restatement of the Winograd forms the fp32 kernels use since round 4, so that the
algebra the HIP kernels rely on is checked on the CPU against the plain convolution they replace.

  * denoiser_persist.hip, WINO instances (cm-tts_amd/csrc/cmtts_api.hip: to_wino_fragments): the gated k = 3, dilation-1 conv of
    ResidualBlock.forward (reference model/blocks.py:672) as F(2,3) over frame pairs;
  * conv_xlq.hip: conv_xlq_kernel (round 5): the dilation-1 ResBlock convs of HiFi-GAN as F(4,3) tap groups over output quads
    (conv1d_f43_taps below);
  * denoiser_persist.hip, WINO == 2 instances (round 5; cmtts_api.hip: to_wino43_fragments): the same conv as F(4,3) over frame quads
    (conv1d_f43 below: the kernel's transforms in the kernel's operation order);
  * resblock_pair.hip: conv_xlw_kernel (cm-tts_amd/csrc/resblock_pair.h: WinoTab<k>, cmtts_api.hip: to_wino_iter_fragments): the k = 3 / 7 / 11 dilated convs of
    hifigan ResBlock1 (reference hifigan/models.py:96-103) over output pairs one dilation apart — groups of three taps as F(2,3), a remainder of two taps
    as F(2,2), a single remaining tap directly.

Nothing here is imported by the product path (cm-tts_amd/, bench.py's timed region)."""
import numpy as np

# (accumulator, a, b, sgn, weight kind, tau): M[acc] += W_kind(tau) * (X(a) + sgn * X(b)); X(m) = the input m dilated taps to the right of the
# pair's first output's leftmost tap;  y(t) = (M0 + M1) + M2,  y(t + dil) = (M1 - M2) - M3          (resblock_pair.h: WinoEntry)
def _f23(t):
    return [(0, t, t + 2, -1, 0, t), (1, t + 1, t + 2, +1, 1, t), (2, t + 2, t + 1, -1, 2, t), (3, t + 1, t + 3, -1, 3, t)]


def _f22(t):
    return [(0, t, t + 1, -1, 0, t), (1, t + 1, 0, 0, 5, t), (3, t + 1, t + 2, -1, 6, t)]


def _one(t):
    return [(0, t, 0, 0, 0, t), (3, t + 1, 0, 0, 4, t)]


WINO_TAB = {3: _f23(0), 7: _f23(0) + _f23(3) + _one(6), 9: _f23(0) + _f23(3) + _f23(6), 11: _f23(0) + _f23(3) + _f23(6) + _f22(9)}      # 9: the FFT blocks' FFN conv (conv_xres.hip, WQ == 2)


def wino_weight(g, kind, tau):
    """g: [..., k] taps (last axis) -> the transformed weight of one table entry (cmtts_api.hip: to_wino_iter_fragments_k)."""
    t = lambda i: g[..., i]
    return {0: lambda: t(tau), 1: lambda: 0.5 * (t(tau) + t(tau + 1) + t(tau + 2)), 2: lambda: 0.5 * (t(tau) - t(tau + 1) + t(tau + 2)),
            3: lambda: t(tau + 2), 4: lambda: -t(tau), 5: lambda: t(tau) + t(tau + 1), 6: lambda: t(tau + 1)}[kind]()


def conv1d_direct(x, w, dil=1):
    """x [Cin][T], w [Cout][Cin][k] -> y [Cout][T]: torch.nn.functional.conv1d(padding=dil * (k - 1) // 2, dilation=dil) (cross-correlation)."""
    cout, cin, k = w.shape
    T = x.shape[1]
    pad = dil * (k - 1) // 2
    xp = np.pad(x, ((0, 0), (pad, pad)))
    y = np.zeros((cout, T), x.dtype)
    for tap in range(k):
        y += w[:, :, tap] @ xp[:, tap * dil: tap * dil + T]
    return y


def conv1d_winograd(x, w, dil=1):
    """The same conv through the kernels' table: outputs in pairs (t, t + dil)."""
    cout, cin, k = w.shape
    T = x.shape[1]
    pad = dil * (k - 1) // 2
    tab = WINO_TAB[k]
    # first outputs of the pairs: blocks of 2 dil columns, the first dil of each block
    Tp = -(-T // (2 * dil)) * (2 * dil)
    xp = np.pad(x, ((0, 0), (pad, pad + (Tp - T) + dil * (k + 1))))
    t_first = np.asarray([q * 2 * dil + r for q in range(Tp // (2 * dil)) for r in range(dil)])
    X = lambda m: xp[:, t_first + m * dil]                      # [Cin][pairs]
    M = [np.zeros((cout, t_first.size), x.dtype) for _ in range(4)]
    for acc, a, b, sgn, kind, tau in tab:
        v = X(a) + sgn * X(b) if sgn else X(a)
        M[acc] = M[acc] + wino_weight(w, kind, tau).astype(x.dtype) @ v
    y = np.zeros((cout, Tp + dil), x.dtype)
    y[:, t_first] = (M[0] + M[1]) + M[2]
    y[:, t_first + dil] = (M[1] - M[2]) - M[3]
    return y[:, :T]


def f43_weights(w):
    """w [Cout][Cin][3] -> the six transformed weights U_p [Cout][Cin] (points 0, +-1, +-2, inf; cmtts_api.hip: to_wino43_fragments forms them in
    double and rounds once)."""
    g0, g1, g2 = (w[..., i].astype(np.float64) for i in range(3))
    return [g0 / 4.0, -(g0 + g1 + g2) / 6.0, -(g0 - g1 + g2) / 6.0, g0 / 24.0 + g1 / 12.0 + g2 / 6.0, g0 / 24.0 - g1 / 12.0 + g2 / 6.0, g2]


def conv1d_f43(x, w):
    """The k = 3, dilation-1, padding-1 conv through F(4,3): outputs in quads 4q .. 4q + 3 from inputs d0 .. d5 = x(4q - 1 .. 4q + 4)
    (denoiser_persist.hip, WINO == 2: transform4 / out4, same expressions)."""
    cout, cin, k = w.shape
    assert k == 3
    T = x.shape[1]
    Tq = -(-T // 4) * 4
    xp = np.pad(x, ((0, 0), (1, Tq - T + 4)))
    q0 = np.arange(0, Tq, 4)
    d = [xp[:, q0 + i] for i in range(6)]                      # [Cin][quads] each
    dt = x.dtype.type
    t0, t1 = d[4] - dt(4) * d[2], d[3] - dt(4) * d[1]
    t2, t3 = d[4] - d[2], d[3] - d[1]
    V = [dt(4) * d[0] + (d[4] - dt(5) * d[2]), t0 + t1, t0 - t1, t2 + dt(2) * t3, t2 - dt(2) * t3, dt(4) * d[1] + (d[5] - dt(5) * d[3])]
    U = [u.astype(x.dtype) for u in f43_weights(w)]
    m = [U[p] @ V[p] for p in range(6)]
    s12, d12, s34, d34 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
    y = np.zeros((cout, Tq), x.dtype)
    y[:, q0] = (m[0] + s12) + s34
    y[:, q0 + 1] = d12 + dt(2) * d34
    y[:, q0 + 2] = s12 + dt(4) * s34
    y[:, q0 + 3] = (d12 + dt(8) * d34) + m[5]
    return y[:, :T]


# conv_xlq.hip: QTab<k> — per k the entries (kind, tap offset): F(4,3) groups of three taps (a tap beyond the kernel is zero), k = 7's seventh tap alone
# (k = 9: the FFT blocks' FFN conv, conv_xres.hip WQ instances — three groups; k = 5: the frame-level pitch predictor, conv_k5q.hip — two groups, the sixth tap zero)
F43_TAPS = {3: [("f43", 0)], 5: [("f43", 0), ("f43", 3)], 7: [("f43", 0), ("f43", 3), ("one", 6)], 9: [("f43", 0), ("f43", 3), ("f43", 6)],
            11: [("f43", 0), ("f43", 3), ("f43", 6), ("f43", 9)]}


def conv1d_f43_taps(x, w):
    """The k = 3 / 5 / 7 / 9 / 11, dilation-1, padding-(k-1)/2 conv through conv_xlq_kernel's (k = 9: conv_xres_kernel<WQ>'s) products (cm-tts_amd/csrc/conv_xlq.hip; weights as
    cmtts_api.hip: to_wino43_iter_fragments forms them): all tap groups into six transform-domain accumulators, one output transform."""
    cout, cin, k = w.shape
    T = x.shape[1]
    Tq = -(-T // 4) * 4
    pad = (k - 1) // 2
    xp = np.pad(x, ((0, 0), (pad, Tq - T + pad + 2)))
    q0 = np.arange(0, Tq, 4)
    dt = x.dtype.type
    M = [np.zeros((cout, q0.size), x.dtype) for _ in range(6)]
    wz = np.concatenate([w, np.zeros((cout, cin, 2), w.dtype)], axis=2)
    for kind, o in F43_TAPS[k]:
        if kind == "f43":
            d = [xp[:, q0 + o + i] for i in range(6)]
            t0, t1 = d[4] - dt(4) * d[2], d[3] - dt(4) * d[1]
            t2, t3 = d[4] - d[2], d[3] - d[1]
            V = [dt(4) * d[0] + (d[4] - dt(5) * d[2]), t0 + t1, t0 - t1, t2 + dt(2) * t3, t2 - dt(2) * t3, dt(4) * d[1] + (d[5] - dt(5) * d[3])]
            U = [u.astype(x.dtype) for u in f43_weights(wz[:, :, o:o + 3])]
            for p in range(6):
                M[p] = M[p] + U[p] @ V[p]
        else:
            xs = [xp[:, q0 + o + i] for i in range(4)]
            g = wz[:, :, o].astype(np.float64)
            for p, (u, v) in zip((0, 1, 2, 5), ((g, xs[0] - xs[2]), (0.5 * g, xs[1] + xs[2]), (0.5 * g, xs[2] - xs[1]), (g, xs[3] - xs[1]))):
                M[p] = M[p] + u.astype(x.dtype) @ v
    s12, d12, s34, d34 = M[1] + M[2], M[1] - M[2], M[3] + M[4], M[3] - M[4]
    y = np.zeros((cout, Tq), x.dtype)
    y[:, q0] = (M[0] + s12) + s34
    y[:, q0 + 1] = d12 + dt(2) * d34
    y[:, q0 + 2] = s12 + dt(4) * s34
    y[:, q0 + 3] = (d12 + dt(8) * d34) + M[5]
    return y[:, :T]
