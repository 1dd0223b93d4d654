"""GPU (-m gpu): batch sizes around the kernels' internal boundaries (one instance, fewer instances than lanes in a wave /
workgroup, one more than the persistent raster grid of 14,336 workgroups, not a multiple of anything)."""
import pytest

from gpu_parity import run_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 5, 257, 14337])
@pytest.mark.parametrize("env_id", ["MortarMayhem-Grid-v0", "Endless-MysteryPath-v0", "Endless-SearingSpotlights-v0", "MysteryPath-Grid-v0"])
def test_odd_batch_sizes(env_id, n):
    run_parity(env_id, None, n=n, steps=24 if n > 1000 else 90, check_every=1 if n < 1000 else 6)


# The spotlight family changes kernels with the launch size (mg_raster.hpp raster_nt() / raster_grid(), mg_spot.hip FUSE_MAX):
# plain stores up to 16,384 frames, non-temporal beyond; 9,728 workgroups up to 24,576 frames, 14,336 beyond; the finite
# variant serves its resets inside the raster launch up to 65,536 instances.  One size on either side of every switch, every
# instance against the oracle.
@pytest.mark.parametrize("env_id,n", [
    ("Endless-SearingSpotlights-v0", 16385), ("Endless-SearingSpotlights-v0", 24577),
    ("SearingSpotlights-v0", 16385), ("SearingSpotlights-v0", 65536),
    pytest.param("SearingSpotlights-v0", 65537, marks=pytest.mark.slow),  # (the other side of FUSE_MAX: the same kernels as 65,536 / 16,385)
])
def test_spotlight_launch_size_switches(env_id, n):
    if n > 60000:  # (every instance truncated in steps 8 and 16: the resets of both arrangements at their largest / smallest size)
        run_parity(env_id, dict(max_steps=8), n=n, steps=18, check_every=8)
    else:
        run_parity(env_id, None, n=n, steps=36, check_every=9)
