"""Tests-only NumPy restatements around SustainDC.reset's weather pipeline (TEST INFRASTRUCTURE, never imported by the
product):

  * `reference_weather_reset` -- what `Weather_Manager.reset` does with its draws (utils/managers.py:596-613): add the
    coherent noise to dry / wet bulb, np.roll by whole days, clip to [0, 45], 30-day min / max from the cursor.  Pinned
    against the reference itself by tests/golden/weather_resets.npz (tests/test_weather_fixture.py).
  * `coherent_noise_legacy` -- `CoherentNoise.generate` (managers.py:35-48) on NumPy's legacy global MT19937 stream.
  * `device_reset_expected` -- the DEVICE's draw scheme restated operation for operation (csrc/sdc_reset.hip): Philox4x32-10
    keyed on (seed, global env index, episode), multiply-shift ranges for day / hour / roll, four fp32 Box-Muller normals
    per Philox4x32-7 block, fp64 random walk, population std -> the same arithmetic as above.
"""
from __future__ import annotations

import numpy as np

TL = 35040
M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Philox4x32-R (Salmon et al., SC'11) on uint32 arrays (broadcast); returns 4 uint32 arrays.  Pinned by Random123's
    known-answer vectors for R = 7 and R = 10 (tests/test_reset_ref.py)."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & MASK for x in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    return philox4x32(c0, c1, c2, c3, k0, k1, 10)


def coherent_noise_legacy(seed, weight=0.02, desired_std=0.75, n=TL):
    """np.random.seed(seed); CoherentNoise(base=0, weight, desired_std).generate(n); then the roll draw that
    Weather_Manager.reset makes right after (np.random.randint(0, 14))."""
    rs = np.random.RandomState(seed)          # the same MT19937 stream as the legacy global functions after seed()
    steps = rs.normal(loc=0, scale=1, size=n)
    walk = np.cumsum(weight * steps)
    noise = (walk / np.std(walk)) * desired_std
    roll_days = int(rs.randint(0, 14))
    return noise, roll_days


def reference_weather_reset(T_orig, WB_orig, noise, roll_days, c0, window):
    """-> dict(T, WB [window] from the cursor, t_min, t_max over 30 days from the cursor)."""
    T = np.clip(np.roll(T_orig + noise, roll_days * 96), 0, 45)
    WB = np.clip(np.roll(WB_orig + noise, roll_days * 96), 0, 45)
    seg = T[c0:c0 + 2880]
    return dict(T=T[c0:c0 + window], WB=WB[c0:c0 + window], t_min=seg.min(), t_max=seg.max(), T_full=T, WB_full=WB)


def device_draws(seed, env_global, episode, day_lo, day_hi, max_roll_days=14):
    x, y, z, _ = philox4x32_10(0, env_global, episode, 0xD4A7, seed & 0xFFFFFFFF, seed >> 32)
    span = (np.asarray(day_hi, dtype=np.uint64) - np.asarray(day_lo, dtype=np.uint64) + np.uint64(1))
    day = np.asarray(day_lo, dtype=np.int64) + ((x.astype(np.uint64) * span) >> np.uint64(32)).astype(np.int64)
    hour = ((y.astype(np.uint64) * np.uint64(24)) >> np.uint64(32)).astype(np.int64)
    roll = ((z.astype(np.uint64) * np.uint64(max_roll_days)) >> np.uint64(32)).astype(np.int64)
    return day, hour, roll


def device_normals(seed, env_global, episode, n=TL):
    """The n standard normals of one (env, episode): Philox4x32-7 block c -> samples 4c .. 4c+3, fp32 Box-Muller."""
    nb = (n + 3) // 4
    x, y, z, w = philox4x32(np.arange(nb, dtype=np.uint64), env_global, episode, 0x7E47, seed & 0xFFFFFFFF, seed >> 32, 7)
    # (fp32(word) + 0.5) * 2^-32 with ONE rounding after the conversion's own (a fused multiply-add on the device): exact in fp64
    u1, u2, u3, u4 = ((v.astype(np.float32).astype(np.float64) * 2.0 ** -32 + 2.0 ** -33).astype(np.float32) for v in (x, y, z, w))
    c = np.float32(-1.3862943611198906)
    r1 = np.sqrt(c * np.log2(u1), dtype=np.float32)
    r2 = np.sqrt(c * np.log2(u3), dtype=np.float32)
    two_pi = 2.0 * np.pi
    out = np.empty((nb, 4), dtype=np.float32)
    out[:, 0] = r1 * np.cos(two_pi * u2.astype(np.float64)).astype(np.float32)
    out[:, 1] = r1 * np.sin(two_pi * u2.astype(np.float64)).astype(np.float32)
    out[:, 2] = r2 * np.cos(two_pi * u4.astype(np.float64)).astype(np.float32)
    out[:, 3] = r2 * np.sin(two_pi * u4.astype(np.float64)).astype(np.float32)
    return out.reshape(-1)[:n]


def device_reset_expected(tables, seed, env_global, episode, day_lo, day_hi, episode_steps, noise_std=0.75,
                          noise_weight=0.02, max_roll_days=14):
    """Everything sdc_reset_kernel derives for one env: day, hour, roll, cursor, t_win / wb_win [episode_steps + 18],
    t_min / t_den, ci_min / ci_den."""
    day, hour, roll = (int(v) for v in device_draws(seed, env_global, episode, day_lo, day_hi, max_roll_days))
    last_ok = TL - 1 - (episode_steps + 17)            # year-end fence (csrc/sdc_reset.hip)
    if day * 96 + hour * 4 > last_ok:
        c = max(last_ok, 0)
        day, hour = c // 96, (c % 96) // 4
    c0 = day * 96 + hour * 4
    nz = device_normals(seed, env_global, episode).astype(np.float64)
    walk = np.cumsum(noise_weight * nz)
    # the device scales by noise_std / std (one division per reset) and accumulates the walk with fused multiply-adds:
    # the same values to within fp64 rounding (~1e-15), far inside the 2e-6 C of its fp32 transcendentals
    noise = walk * (noise_std / np.sqrt(np.mean(walk * walk) - np.mean(walk) ** 2)) if noise_std > 0 else np.zeros(TL)
    lw = episode_steps + 18
    r = reference_weather_reset(tables["T"], tables["WB"], noise, roll, c0, lw)
    Cseg = tables["C"][c0:c0 + 2880]
    return dict(day=day, hour=hour, roll=roll, c0=c0, t_win=r["T"], wb_win=r["WB"], t_min=r["t_min"],
                t_den=r["t_max"] - r["t_min"], ci_min=Cseg.min(), ci_den=Cseg.max() - Cseg.min(), noise=noise)
