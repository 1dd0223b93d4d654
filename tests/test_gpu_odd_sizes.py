"""GPU (-m gpu): batch sizes around the kernels' internal boundaries (one instance, fewer instances than lanes in a wave /
workgroup, one more than the persistent raster grid of 14,336 workgroups, not a multiple of anything)."""
import pytest

from gpu_parity import run_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 5, 257, 14337])
@pytest.mark.parametrize("env_id", ["MortarMayhem-Grid-v0", "Endless-MysteryPath-v0", "Endless-SearingSpotlights-v0", "MysteryPath-Grid-v0"])
def test_odd_batch_sizes(env_id, n):
    run_parity(env_id, None, n=n, steps=24 if n > 1000 else 90, check_every=1 if n < 1000 else 6)
