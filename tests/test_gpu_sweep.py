"""GPU: a fixed-seed slice of the randomised parity sweep (tests/fuzz_parity.py) -- random (N, image size, colour / covariance mode,
SH degree and coefficient count, splat size, camera, opacity shift, culling on / off, split SH) draws, including 1x1 images, N = 1
and ragged last waves; lists bit-exact (or an ordered sub-list with culling), images and every gradient against the C oracle."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_randomised_parity_sweep(seed):
    from tests.fuzz_parity import run_draws
    n, worst = run_draws(seed=seed, n_draws=80, budget_s=240)
    strict = worst.pop("_strict_draws", 0)
    print(f"\n  seed {seed}: {n} draws, {strict} with every image and gradient within 1e-4 outright (zero threshold flips); "
          "worst max-relative gradient errors: " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    assert n >= 40, f"only {n} draws finished inside the time budget"
    # every draw holds the Gaussians away from its flipped pixels to 1e-4 (tests/common.py check_grads_isolating_flips, asserted per draw);
    # most draws have no flip at all and meet the bar on every entry
    assert worst.get("_far_from_flips", 0.0) < 1e-4
    assert strict >= 0.8 * n, f"only {strict} of {n} draws met the 1e-4 bar outright"
