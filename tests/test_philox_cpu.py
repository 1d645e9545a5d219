"""CPU: the numpy restatement of the in-kernel noise generator (oracle/philox.py) against the known-answer vectors
published with Random123 (kat_vectors: philox4x32 10 rounds), and the statistics of its Box-Muller normals."""
import numpy as np

from oracle.philox import philox4x32_10, step_noise

# (counter, key, expected) -- Random123 examples/kat_vectors, "philox4x32 10"
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox4x32_10_known_answers():
    for ctr, key, want in KAT:
        got = philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert tuple(int(v) for v in got) == want


def test_normals_moments_and_independence():
    z = step_noise(seed=1234567, base=0, step=3, n=200_000, seq_len=4).astype(np.float64)
    assert np.isfinite(z).all()
    assert abs(z.mean()) < 5e-3 and abs(z.var() - 1.0) < 1e-2
    assert abs((z ** 4).mean() - 3.0) < 0.06                        # kurtosis of a unit normal
    c = np.corrcoef(z.T)
    assert np.abs(c - np.eye(4)).max() < 1e-2                        # the four positions of one counter
    z2 = step_noise(seed=1234567, base=0, step=4, n=200_000, seq_len=4).astype(np.float64)
    assert abs(np.corrcoef(z[:, 0], z2[:, 0])[0, 1]) < 1e-2          # consecutive steps
    assert abs(np.corrcoef(z[:-1, 0], z[1:, 0])[0, 1]) < 1e-2        # neighbouring latents


def test_counter_is_the_global_latent_index():
    a = step_noise(seed=9, base=0, step=7, n=64, seq_len=16)
    b = step_noise(seed=9, base=40, step=7, n=24, seq_len=16)
    assert np.array_equal(a[40:], b)
    big = step_noise(seed=9, base=(1 << 32) - 2, step=0, n=4, seq_len=4)   # the index carries into the second counter word
    assert np.isfinite(big).all() and len({tuple(r) for r in big.tolist()}) == 4
