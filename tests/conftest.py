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


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
