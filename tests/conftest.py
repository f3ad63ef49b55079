import os

# exercise the weight-gradient-derived norm sums (csrc/wgrad_sums.hip) on every layer that qualifies, not only on the
# >= 256 MB tensors where the product path switches them on
os.environ.setdefault("TEM_OPT_WGRAD_SUMS_MIN_MB", "0")  # applied through tem_set_option() when the library loads
# DefaultTrainer(mixed_precision=True) -- the reference's default flag -- selects the fp16 mixed mode since round 6 (as in
# the reference).  The suite checks the trainers against fp32 / float64 oracles at 1e-3 unless a test names a dtype, so it
# pins the bare flag to the parity-grade fp32-class path; tests/test_gpu_trainer.py::test_mixed_precision_default_flag
# removes the pin and checks the product default.
os.environ.setdefault("TEM_MIXED_PRECISION", "0")
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_err(a, b, floor=0.0):
    """max |a-b| / max(|b|) -- the 'relative fp32 tolerance' of the north star.
    `floor` bounds the denominator from below for quantities that are mathematically zero
    (e.g. the bias gradient of a conv that feeds an InstanceNorm is pure round-off, ~1e-8)."""
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor, 1e-30))


def check_grads(named_grads, ref_grads, tol, what="", zero_keys=("decoder.samplers.", ".conv.bias")):
    """Every parameter gradient within `tol` (relative to that tensor's max) plus an absolute
    round-off allowance of 1e-6 x the largest gradient of the model.
    Gradients that are MATHEMATICALLY ZERO -- the bias of the sampler's 1x1 conv, whose output goes
    (through the concat) straight into a per-channel norm that removes any constant -- are sums of
    millions of cancelling terms: both sides hold only fp32 summation noise (~1e-7 * sum|terms|), so
    they are checked to be small relative to the model's gradient scale (1e-4) instead."""
    import numpy as np
    ref = {k: np.asarray(v, dtype=np.float64) for k, v in ref_grads.items()}
    gscale = max(float(np.abs(v).max()) for v in ref.values())
    worst = 0.0
    for k, g in named_grads.items():
        a = np.asarray(g, dtype=np.float64)
        diff = float(np.abs(a - ref[k]).max())
        rmax = float(np.abs(ref[k]).max())
        if all(z in k for z in zero_keys) and rmax < 1e-4 * gscale:
            assert float(np.abs(a).max()) < 1e-4 * gscale, f"{what}{k}: should be ~0, got {np.abs(a).max():.3e}"
            continue
        bound = tol * rmax + 1e-6 * gscale
        assert diff <= bound, f"{what}{k}: |diff|={diff:.3e} > {bound:.3e} (|ref|max={rmax:.3e}, gscale={gscale:.3e})"
        worst = max(worst, diff / max(rmax, 1e-6 * gscale / tol))
    return worst
