"""Host-side logic that needs no GPU: slab partition, level count, synthetic inputs, start vector."""
import numpy as np
import pytest


def test_partition_covers_domain():
    from ma_amd.slab import Partition
    for n, w in ((4096, 8), (4096, 1), (70, 4), (55, 3)):
        parts = [Partition(n, w, r) for r in range(w)]
        assert parts[0].j0 == 0 and parts[-1].j1 == n
        for a, b in zip(parts[:-1], parts[1:]):
            assert a.j1 == b.j0
        assert max(p.nl for p in parts) - min(p.nl for p in parts) <= 1
    with pytest.raises(ValueError):
        Partition(8, 8, 0)


def test_num_levels_matches_reference_formula():
    from ma_amd.drivers import num_levels
    from oracle.operators import num_levels as ref
    for m, n in ((64, 48), (200, 200), (1024, 1024), (4096, 4096), (70, 66), (33, 500)):
        assert num_levels(m, n) == ref(m, n)


def test_synthetic_images_deterministic_and_normalised():
    from ma_amd import synth
    a = synth.dynamic_test_image(40, 36, 3).numpy()
    b = synth.dynamic_test_image(40, 36, 3).numpy()
    assert np.array_equal(a, b) and a.shape == (3, 40, 36)
    assert a.min() == 0.0 and a.max() == 1.0
    u = synth.smooth_displacement(40, 36, 3).numpy()
    assert u.shape == (3, 2, 36, 40)


def test_power_iteration_start_vector_is_exact():
    from oracle.drivers import normest_start
    x = normest_start(1000)
    assert x.min() >= -1.0 and x.max() < 1.0
    assert x[1] == 2654435761 / 2147483648.0 - 1.0
    assert len(np.unique(x)) == 1000
