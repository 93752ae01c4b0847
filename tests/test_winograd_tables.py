"""CPU: the Winograd tables the fp32 kernels use since round 4 (oracle/winograd_ref.py restates cm-tts_amd/csrc/resblock_pair.h: WinoTab and the two
host-side weight transforms) reproduce the plain convolution — exactly in float64 up to rounding, and to fp32 rounding in float32."""
import numpy as np
import pytest

from oracle import winograd_ref as W


@pytest.mark.parametrize("k,dil", [(3, 1), (3, 3), (3, 5), (7, 1), (7, 3), (7, 5), (9, 1), (11, 1), (11, 3), (11, 5)])
def test_winograd_table_equals_direct_conv(k, dil):
    rs = np.random.RandomState(100 * k + dil)
    for T in (1, 2, 59, 60, 61, 64, 137):
        x = rs.standard_normal((6, T))
        w = rs.standard_normal((5, 6, k))
        ref = W.conv1d_direct(x, w, dil)
        got = W.conv1d_winograd(x, w, dil)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()), (k, dil, T)


def test_winograd_products_per_pair():
    # 4 / 10 / 15 products per output pair instead of 6 / 14 / 22 (k = 9, the FFT blocks' FFN conv: 12 instead of 18)
    assert {k: len(t) for k, t in W.WINO_TAB.items()} == {3: 4, 7: 10, 9: 12, 11: 15}
    for k, tab in W.WINO_TAB.items():
        assert all(0 <= acc <= 3 and max(a, b) <= k for acc, a, b, sgn, kind, tau in tab)


def test_winograd_fp32_error_is_rounding_level():
    rs = np.random.RandomState(7)
    x = rs.standard_normal((128, 300)).astype(np.float32)
    for k, dil in ((3, 1), (7, 3), (9, 1), (11, 5)):
        w = (rs.standard_normal((64, 128, k)) / np.sqrt(128 * k)).astype(np.float32)
        ref64 = W.conv1d_direct(x.astype(np.float64), w.astype(np.float64), dil)
        e_dir = np.abs(W.conv1d_direct(x, w, dil) - ref64).max()
        e_win = np.abs(W.conv1d_winograd(x, w, dil) - ref64).max()
        assert e_win < 4 * e_dir + 1e-6, (k, dil, e_win, e_dir)


@pytest.mark.parametrize("T", [1, 3, 4, 63, 64, 65, 130])
def test_f43_equals_direct_conv(T):
    """Round 5: the F(4,3) form of the denoiser's gated conv (denoiser_persist.hip, WINO == 2) is the k = 3 conv — exactly, in float64, at
    lengths that are not multiples of the quad."""
    rs = np.random.RandomState(T)
    x = rs.standard_normal((16, T))
    w = rs.standard_normal((8, 16, 3))
    assert np.abs(W.conv1d_f43(x, w) - W.conv1d_direct(x, w, 1)).max() < 1e-12


def test_f43_fp32_error_and_its_growth_with_the_input_scale():
    """fp32: within an order of magnitude of the direct form on one conv (the kernels measure 1.5e-5 against 8e-6 for F(2,3) on one network evaluation);
    the error is RELATIVE to the inputs' magnitude in every form, F(4,3) with the largest constant (transform coefficients up to 5 and 8) —
    what test_winograd_and_fp16x3_stress_statistics[near_fp16_max] sees on the GPU."""
    rs = np.random.RandomState(11)
    w = (rs.standard_normal((64, 256, 3)) / np.sqrt(3 * 256)).astype(np.float32)
    for scale in (1.0, 3e4):
        x = (rs.standard_normal((256, 256)) * scale).astype(np.float32)
        ref64 = W.conv1d_direct(x.astype(np.float64), w.astype(np.float64), 1)
        e_dir = np.abs(W.conv1d_direct(x, w, 1) - ref64).max()
        e_23 = np.abs(W.conv1d_winograd(x, w, 1) - ref64).max()
        e_43 = np.abs(W.conv1d_f43(x, w) - ref64).max()
        assert e_23 < 4 * e_dir and e_43 < 16 * e_dir, (scale, e_dir, e_23, e_43)      # measured: 1.5x and 8x on the MAXIMUM of one conv (rms: 2x)
        assert e_43 < 3e-5 * scale


@pytest.mark.parametrize("k", [3, 5, 7, 9, 11])
def test_f43_tap_groups_equal_direct_conv(k):
    """conv_xlq_kernel's products (HiFi-GAN dilation-1 convs, round 5; k = 9: the FFT blocks' FFN conv in conv_xres.hip): F(4,3) groups of three taps
    + k = 7's single tap + k = 11's zero twelfth tap, all into six accumulators, reproduce the k-tap conv — exactly in float64 at ragged lengths, to fp32 rounding in float32."""
    rs = np.random.RandomState(k)
    for T in (1, 2, 5, 63, 64, 66, 131):
        x = rs.standard_normal((16, T))
        w = rs.standard_normal((8, 16, k))
        assert np.abs(W.conv1d_f43_taps(x, w) - W.conv1d_direct(x, w, 1)).max() < 1e-11, (k, T)
    x = rs.standard_normal((128, 300)).astype(np.float32)
    w = (rs.standard_normal((64, 128, k)) / np.sqrt(128 * k)).astype(np.float32)
    ref64 = W.conv1d_direct(x.astype(np.float64), w.astype(np.float64), 1)
    e_dir = np.abs(W.conv1d_direct(x, w, 1) - ref64).max()
    e_q = np.abs(W.conv1d_f43_taps(x, w) - ref64).max()
    assert e_q < 16 * e_dir + 1e-6, (k, e_q, e_dir)
