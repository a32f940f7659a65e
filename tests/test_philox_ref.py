"""CPU-only: the NumPy restatement of the device noise generator reproduces the published Philox4x32-10
known-answer vectors (Random123 kat_vectors), so the GPU test that compares the kernel with it pins the kernel to
the published algorithm."""
import numpy as np

import philox_ref as PR

KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox4x32_10_known_answers():
    for ctr, key, want in KAT:
        out = PR.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], *key)
        assert tuple(int(o[0]) for o in out) == want


def test_normals_are_standard_and_offsets_are_consistent():
    z = PR.normals(0, 400000, 215)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    np.testing.assert_array_equal(PR.normals(1003, 500, 215), z[1003:1503])
    m = PR.nan_mask(0, 200000, 3, 0.3)
    assert abs(m.mean() - 0.3) < 5e-3
