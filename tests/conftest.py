import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # a fresh checkout has no libcmtts_hip.so (build artefacts are git-ignored): build it once (hipcc cross-compiles
    # gfx950 without a GPU; the GPU box only ever sees the prebuilt file)
    lib = os.path.join(ROOT, "cm-tts_amd", "libcmtts_hip.so")
    if not os.path.exists(lib):
        import shutil
        import subprocess
        if shutil.which("hipcc"):
            subprocess.run(["make", "-C", os.path.join(ROOT, "cm-tts_amd", "csrc"), "-j4"], check=True,
                           stdout=subprocess.DEVNULL)


# Lines the GPU tests want in the run's summary even under `-q` without `-s` (flip counts, dtype error ladders):
# collected here, printed by pytest_terminal_summary.
REPORT = []


def report(line):
    REPORT.append(line)
    print(line)


def pytest_terminal_summary(terminalreporter):
    if REPORT:
        terminalreporter.write_sep("-", "parity report")
        for line in REPORT:
            terminalreporter.write_line(line)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def golden_noise(seed, shape, n):
    """Same draw as tests/golden/make_golden.py:draw_noise."""
    return [np.random.RandomState(seed + 1000 * i).standard_normal(size=shape).astype(np.float32)
            for i in range(n)]


def pitch_margin_mask(f0_denorm, thr=2e-3):
    """True where the golden pre-rounding pitch bucket value is at least `thr` away from a
    rounding boundary (float64 restatement of f0_to_coarse's scaling)."""
    f0 = f0_denorm.astype(np.float64)
    mel = 1127 * np.log(1 + f0 / 700)
    lo, hi = 1127 * np.log(1 + 50.0 / 700), 1127 * np.log(1 + 1100.0 / 700)
    sc = np.where(mel > 0, (mel - lo) * 254 / (hi - lo) + 1, mel)
    sc = np.clip(sc, 1, 255)
    frac = sc + 0.5 - np.floor(sc + 0.5)
    return np.minimum(frac, 1 - frac) > thr


# Pitch-bucket flips: `(mel + 0.5).long()` (utils/pitch_tools.py:26-35) is discontinuous, and the value it rounds is
# computed from the cwt predictor's fp32 output, which differs from the reference's CPU result in the last bits
# (different accumulation order).  A frame whose pre-rounding value sits within that drift of a bucket boundary may land
# in the neighbouring bucket.  The tests COUNT those frames exactly, require every one of them to sit on a boundary
# (golden margin < FLIP_MARGIN of a bucket), and pin the count per golden: a change of the count is a change of the
# kernels' numerics and must be looked at.  Measured on MI355X (round 2); 0 everywhere means no masking takes place.
FLIP_MARGIN = 2e-3
KNOWN_PITCH_FLIPS = {
    "cmtts_LJSpeech": 0, "cmtts_VCTK": 0, "cmtts_LibriTTS": 0,
    "controls_VCTK:ctl": 0, "controls_VCTK:tf": 0, "cmtts_VCTK_table": 0,
}


def pitch_flips(p_idx, golden_p_idx, golden_f0_denorm, tag):
    """Exact flip accounting.  Returns the boolean map of agreeing frames; prints `PITCH_FLIPS <tag> n/N`."""
    same = np.asarray(p_idx) == np.asarray(golden_p_idx)
    n = int((~same).sum())
    line = f"PITCH_FLIPS {tag}: {n} of {same.size} frames (pinned: {KNOWN_PITCH_FLIPS[tag]})"
    if line not in REPORT:
        report(line)
    on_boundary = ~pitch_margin_mask(golden_f0_denorm, FLIP_MARGIN)
    assert not (~same & ~on_boundary).any(), f"{tag}: a pitch bucket differs away from a rounding boundary"
    assert (np.abs(np.asarray(p_idx) - np.asarray(golden_p_idx)) <= 1).all(), f"{tag}: a pitch bucket is off by more than one"
    assert n <= KNOWN_PITCH_FLIPS[tag], f"{tag}: {n} pitch-bucket flips, {KNOWN_PITCH_FLIPS[tag]} known"
    return same


def near_flip_mask(same, reach=24):
    """Frames farther than the denoiser's receptive field (+-20 frames) from any flipped frame."""
    near = np.zeros_like(same)
    for b, t in zip(*np.nonzero(~same)):
        near[b, max(0, t - reach): t + reach + 1] = True
    return ~near


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


# The fp32 persistent denoiser stack runs its gated k = 3 conv as a Winograd convolution by default (F(2,3) in round 4, F(4,3) since
# round 5; model option "winograd", process-wide A/B switch "persist_wino"): 2/3 resp. 1/2 of the conv's MFMAs, NOT bitwise the direct form
# of the per-layer kernels (measured: F(2,3) <= 9e-6 / F(4,3) <= 1.6e-5 on one network evaluation, 4e-6 / 8e-6 on a T = 2 mel).  Tests that compare the persistent stack with the per-layer kernels
# run in both forms: "direct" keeps every assertion bitwise, "winograd" keeps the persistent-vs-persistent assertions bitwise and
# bounds the persistent-vs-per-layer ones by WINO_TOL.
WINO_TOL = 3e-5


@pytest.fixture(params=["direct", "winograd"])
def conv_form(request):
    from cmtts_amd import _lib
    import os
    # "winograd" = the default Winograd stack (3: the 8-wave F(4,3) instances; CMTTS_TEST_WINO=1 runs the F(2,3) instances through every conv_form test)
    prev = _lib.internal_set("persist_wino", 0 if request.param == "direct" else int(os.environ.get("CMTTS_TEST_WINO", "3")))
    try:
        yield request.param
    finally:
        _lib.internal_set("persist_wino", prev)


# F(4,3) forms every output of a frame QUAD from all six inputs of the quad: the inputs outside an output's own three taps cancel
# exactly in real arithmetic but not in its rounding, so a frame near the end of a TRIMMED utterance (cmtts_sample_ragged: frames beyond
# the computed range are zeros instead of denoised padding) can differ from the untrimmed run in the last bit although no trimmed frame
# is inside its receptive field (measured 1.1e-6 after one evaluation, a few 1e-6 after four).  Direct and F(2,3) (each output from its own taps only) stay bit for bit.
WINO_TRIM_TOL = 1.5e-5


def trim_exact(form):
    import os
    return form == "direct" or os.environ.get("CMTTS_TEST_WINO", "3") in ("1", "2")


def same_trimmed(a, b, form):
    import torch
    if trim_exact(form):
        return torch.equal(a, b)
    return a.shape == b.shape and float((a.float() - b.float()).abs().max()) <= WINO_TRIM_TOL


def same_result(a, b, form, strict=False):
    """torch.equal(a, b) for the direct form (and wherever `strict`: both sides come from the persistent stack); max|a - b| <= WINO_TOL
    for the Winograd form against the per-layer kernels."""
    import torch
    if form == "direct" or strict:
        return torch.equal(a, b)
    return a.shape == b.shape and float((a.float() - b.float()).abs().max()) <= WINO_TOL


# The fp32 HiFi-GAN generator runs the ResBlock convs of its C >= 128 stages in a Winograd form (conv_xlw_kernel, round 4; vocoder option
# "winograd", A/B switch "voc_wino") once a launch has >= 1024 column tiles: large batches differ from the direct form — which small
# batches keep — by fp32 rounding (measured <= 1.1e-6 on the waveform).  Tests of batch independence run in both forms.
VOC_WINO_TOL = 5e-6


@pytest.fixture(params=["direct", "winograd"])
def voc_form(request):
    from cmtts_amd import _lib
    prev = _lib.internal_set("voc_wino", 0 if request.param == "direct" else 1)
    try:
        yield request.param
    finally:
        _lib.internal_set("voc_wino", prev)


def same_wav(a, b, form):
    """Waveforms (float tensors): bit for bit in the direct form, within VOC_WINO_TOL in the Winograd form."""
    import torch
    if form == "direct":
        return torch.equal(a, b)
    return a.shape == b.shape and float((a.float() - b.float()).abs().max()) <= VOC_WINO_TOL


def same_pcm(a, b, form):
    """int16 PCM (numpy arrays or tensors): identical in the direct form, within one LSB in the Winograd form (a 1e-6 difference of the
    waveform moves a sample across a rounding boundary now and then)."""
    a = np.asarray(a.cpu() if hasattr(a, "cpu") else a).astype(np.int32)
    b = np.asarray(b.cpu() if hasattr(b, "cpu") else b).astype(np.int32)
    if a.shape != b.shape:
        return False
    if form == "direct":
        return bool(np.array_equal(a, b))
    return bool(np.abs(a - b).max() <= 1)
