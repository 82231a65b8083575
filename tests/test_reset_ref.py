"""The NumPy restatement of the device's counter-based generator (tests/reset_ref.py) against Random123's published
known-answer vectors (kat_vectors: `philox4x32 R ctr[4] key[2] -> out[4]`) for both round counts the device uses: R = 10 (the
per-episode day / hour / roll draws, the actor's sampler) and R = 7 (the normals of a reset's weather walk).  The device code is
pinned against this restatement on the GPU (tests/test_gpu_reset_pin.py)."""
import numpy as np
import pytest

from tests import reset_ref as RR

PI = (0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)
KAT = {
    7: [((0,) * 6, (0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48)),
        ((0xFFFFFFFF,) * 6, (0x5207DDC2, 0x45165E59, 0x4D8EE751, 0x8C52F662)),
        (PI, (0x4DFCCABA, 0x190A87F0, 0xC47362BA, 0xB6B5242A))],
    10: [((0,) * 6, (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
         ((0xFFFFFFFF,) * 6, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
         (PI, (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))],
}


@pytest.mark.parametrize("rounds", [7, 10])
def test_philox4x32_known_answers(rounds):
    for inp, want in KAT[rounds]:
        got = RR.philox4x32(*inp, rounds=rounds)
        assert tuple(int(np.asarray(g).reshape(-1)[0]) for g in got) == want, (rounds, inp)
    # the 10-round alias the draws use
    assert [int(np.asarray(g).reshape(-1)[0]) for g in RR.philox4x32_10(*KAT[10][2][0])] == list(KAT[10][2][1])


def test_device_normals_are_standard_normal():
    """Moments of the restated Box-Muller normals of one (env, episode): 35 040 samples."""
    z = RR.device_normals(0x1234ABCD5, 1007, 3).astype(np.float64)
    assert z.shape == (RR.TL,)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    assert abs(((z - z.mean()) ** 3).mean()) < 0.05 and abs(((z - z.mean()) ** 4).mean() - 3.0) < 0.15
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 0.02
