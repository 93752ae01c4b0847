"""TEST INFRASTRUCTURE (like everything under oracle/): a numpy restatement of the Winograd forms the fp32 kernels use since round 4, so that the
algebra the HIP kernels rely on is checked on the CPU against the plain convolution they replace.

  * denoiser_persist.hip, WINO instances (cm-tts_amd/csrc/cmtts_api.hip: to_wino_fragments): the gated k = 3, dilation-1 conv of
    ResidualBlock.forward (reference model/blocks.py:672) as F(2,3) over frame pairs;
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


WINO_TAB = {3: _f23(0), 7: _f23(0) + _f23(3) + _one(6), 11: _f23(0) + _f23(3) + _f23(6) + _f22(9)}


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
