"""The level-synchronous form of the adaptive grid sampler equals the reference's stack walk.

CPU-only: both forms are small Python models (tests/adaptive_model.py); the device kernels
follow levelsync() and are compared with the oracle's stack walk in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import adaptive_model


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_level_synchronous_form_equals_the_stack_walk(seed):
    rng = np.random.default_rng(seed)
    subdivided = 0
    for _ in range(14):
        W0, H0 = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        D = int(rng.integers(0, 4))
        thr = float(rng.choice([.02, .05, .2]))
        div = 1 << D
        ny, nx = div * H0 + 1, div * W0 + 1
        # piecewise smooth RGBA: a ramp plus noise, and a blob with sharp edges
        base = rng.random((ny, nx, 4))
        yy, xx = np.mgrid[0:ny, 0:nx]
        blob = (xx - rng.integers(0, nx)) ** 2 + (yy - rng.integers(0, ny)) ** 2 < (div * 1.3) ** 2
        img = .03 * np.sin(xx / 3.)[:, :, None] + .02 * base + blob[:, :, None] * base * float(rng.choice([.1, 1.]))

        def colour(x, y):
            return img[y, x]
        a, traced_a = adaptive_model.sequential(W0, H0, D, thr, colour)
        b, traced_b = adaptive_model.levelsync(W0, H0, D, thr, colour)
        assert np.array_equal(traced_a, traced_b), (W0, H0, D, thr)
        assert np.array_equal(a, b), (W0, H0, D, thr)
        subdivided += int(traced_a.sum() > (W0 + 1) * (H0 + 1))
    assert subdivided >= 4            # the trials do exercise subdivision
