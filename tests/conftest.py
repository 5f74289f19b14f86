import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
HOSTSIM = os.path.join(ROOT, "tests", "hostsim", "_build", "hostsim")
HOSTSIM_LEAN = os.path.join(ROOT, "tests", "hostsim", "_build", "hostsim_lean")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "yart_ref")
ORACLE_BIN = os.path.join(ROOT, "oracle", "_build", "yart_oracle")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native targets exist (build() is idempotent and cheap when up to date)."""
    if not (os.path.exists(os.path.join(ROOT, "yart_amd", "libyart_hip.so")) and os.path.exists(HOSTSIM) and os.path.exists(HOSTSIM_LEAN)):
        import __graft_entry__
        __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def hostsim(built):
    return HOSTSIM


@pytest.fixture(scope="session")
def hostsim_lean(built):
    """hostsim whose path tracer runs the lean kernels' hand-over logic (TRAV_FAST walk first, general walk for the rays it hands
    over) and reports on stderr every ray it keeps whose result differs from the general walk's."""
    return HOSTSIM_LEAN


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, text=True, **kw)


# The CPU checkers of the `-m gpu` tests. They are git-ignored build products that travel to the GPU box prebuilt; a GPU run
# without them must FAIL, not skip — a green run in which every oracle comparison silently skipped proves nothing
# (set YART_ALLOW_MISSING_CHECKER=1 to turn the failure back into a skip on a development box without the reference).
def _checker(path, what, build=None):
    if not os.path.exists(path) and build is not None:
        subprocess.run(build, capture_output=True)
    if not os.path.exists(path):
        msg = f"{what} ({os.path.relpath(path, ROOT)}) is missing: the GPU parity tests compare against it"
        if os.environ.get("YART_ALLOW_MISSING_CHECKER"):
            pytest.skip(msg)
        pytest.fail(msg)
    return path


@pytest.fixture(scope="session")
def ref_bin():
    return _checker(REF_BIN, "the compiled reference")


@pytest.fixture(scope="session")
def oracle_bin():
    return _checker(ORACLE_BIN, "the oracle restatement", ["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])


@pytest.fixture(scope="session")
def any_checker():
    if os.path.exists(REF_BIN):
        return REF_BIN
    return _checker(ORACLE_BIN, "a CPU checker (reference or oracle)", ["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])


def bit_identical_or_drift(img, ref, tag=""):
    """The golden bar: the frame IS the reference's, bit for bit (the device evaluates glibc's own libm algorithms,
    DESIGN §1). YART_ALLOW_LIBM_DRIFT=1 — for a box whose glibc selects other libm variants — falls back to north_star's
    RMSE < 1e-3 with most pixels identical."""
    import numpy as np
    same = float(np.mean(np.all(img.view(np.uint32) == ref.view(np.uint32), axis=-1)))
    e = float(np.sqrt(np.mean((np.nan_to_num(img[..., :3]).astype(np.float64) - np.nan_to_num(ref[..., :3]).astype(np.float64)) ** 2)))
    print(f"{tag}: rmse={e:.3e} identical_pixels={same:.6f}")
    if os.environ.get("YART_ALLOW_LIBM_DRIFT"):
        assert e < 1e-3 and same > 0.5, (tag, e, same)
    else:
        assert same == 1.0 and e == 0.0, f"{tag}: not bit-identical to the reference (rmse {e:.3e}, identical pixels {same:.6f}); YART_ALLOW_LIBM_DRIFT=1 relaxes this to RMSE < 1e-3"
    return same, e
