"""The keyed generator's per-draw mixer (two multiplications by 32-bit constants between three folds of
the high half into the low one, behind the Weyl step on a SplitMix64-mixed sample key) — known answers, identity of the three
statements of the spec (oracle, host build of the product core, numpy here), and the statistics a path tracer needs of it:
uniformity of the leading, middle and trailing bits of the 53-bit uniforms, pairwise independence of every two draws a path makes
(the counters it really uses: 13 bounces x 26 slots), serial independence over adjacent samples of a pixel, and the acceptance
rate of the unit-ball rejection loop (vec3.rs:149-160).  The reference draws from rand's ChaCha stream seeded by the OS
(`thread_rng()`): there is no stream to be equal to, only a distribution."""
import ctypes as C

import numpy as np
import pytest

M = (1 << 64) - 1
GAMMA = 0x9E3779B97F4A7C15


def mix64_int(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def mixd_int(z):
    z = ((z ^ (z >> 32)) * 0x9E3779B1) & M
    z = ((z ^ (z >> 32)) * 0x85EBCA6B) & M
    return z ^ (z >> 32)


def sample_key_int(seed, pixel, sample):
    k0 = mix64_int((seed + GAMMA) & M)
    k1 = mix64_int((k0 + pixel * 0xD1B54A32D192ED03) & M)
    return mix64_int((k1 + sample * 0x8CB92BA72F3D8DD7) & M)


def word_int(seed, pixel, sample, block, slot):
    return mixd_int((sample_key_int(seed, pixel, sample) + (block * 1024 + slot + 1) * GAMMA) & M)


def test_known_answers_and_the_three_statements_agree(oracle, hostsim):
    # worked by hand from the definition (python integers): folds by 32 around * 0x9E3779B1 and * 0x85EBCA6B
    assert mixd_int(0) == 0
    assert mixd_int(1) == (0x85EBCA6B * 0x9E3779B1) ^ ((0x85EBCA6B * 0x9E3779B1) >> 32) == 0x52C48C3FC3740AC4   # 1 -> C1 -> (C1 < 2^32: fold is a no-op) C1 C2 -> fold
    assert mixd_int(GAMMA) == 0x503BEA2C745BFCDF
    rng = np.random.default_rng(3)
    for _ in range(300):
        seed, pixel, sample = int(rng.integers(1 << 62)), int(rng.integers(1 << 30)), int(rng.integers(1 << 20))
        block, slot = int(rng.integers(52)), int(rng.integers(64))
        w = word_int(seed, pixel, sample, block, slot)
        assert oracle.probe_word(seed, pixel, sample, block, slot) == w
        assert oracle.probe_uniform(seed, pixel, sample, block, slot) == (w >> 11) * 2.0 ** -53
    # the product core's statement of the same draws (host build): the unit-ball draw of 2000 keys against this file's
    keys = np.array([sample_key_int(5, p, 7) for p in range(2000)], dtype=np.uint64)
    out64 = np.zeros((2000, 3)); out32 = np.zeros((2000, 3), dtype=np.float32)
    hostsim.lib.hostsim_ball.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    assert hostsim.lib.hostsim_ball(2000, keys.ctypes.data, 2, out64.ctypes.data, out32.ctypes.data) == 0
    for i in range(0, 2000, 97):
        k, it = int(keys[i]), 0
        while True:
            base = 3 * 1024 + 32 + 4 * it
            h, second, third = (mixd_int((k + (base + j + 1) * GAMMA) & M) for j in range(3))
            f = [h >> 43, (h >> 22) & 0x1FFFFF, (h >> 1) & 0x1FFFFF]
            lo = [second >> 32, second & 0xFFFFFFFF, third >> 32]
            v = np.array([2.0 * (((a << 32) | b) * 2.0 ** -53) - 1.0 for a, b in zip(f, lo)])
            if v @ v < 1.0:
                break
            it += 1
        assert np.array_equal(out64[i], v)


def _words(keys, ctr):
    with np.errstate(over="ignore"):
        z = keys[:, None] + (ctr[None, :] + np.uint64(1)) * np.uint64(GAMMA)
        z = (z ^ (z >> np.uint64(32))) * np.uint64(0x9E3779B1)
        z = (z ^ (z >> np.uint64(32))) * np.uint64(0x85EBCA6B)
        return z ^ (z >> np.uint64(32))


def _keys(n, seed):
    with np.errstate(over="ignore"):
        def mix64(z):
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))
        k1 = mix64(mix64(np.uint64(seed) + np.uint64(GAMMA)) + np.uint64(1234) * np.uint64(0xD1B54A32D192ED03))   # one pixel ...
        return mix64(k1 + np.arange(n, dtype=np.uint64) * np.uint64(0x8CB92BA72F3D8DD7))                           # ... its samples 0 .. n-1


def test_the_draws_of_a_path_are_uniform_and_independent():
    slots = np.array([0, 1, 2, 8, 9, 16] + list(range(32, 52)), dtype=np.uint64)            # jitter, time, lens, medium 0/1, dielectric, five ball iterations
    ctr = (np.arange(13, dtype=np.uint64)[:, None] * np.uint64(1024) + slots[None, :]).ravel()
    n_keys = 1 << 15
    W = _words(_keys(n_keys, 99), ctr)
    U = (W >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    n = U.size                                                                               # 11 million draws
    assert abs(U.mean() - 0.5) * np.sqrt(12 * n) < 4.5 and abs(U.var() * 12 - 1) < 3e-3
    for shift in (56, 40, 24, 11):                                                           # eight bits from the top, the middle, the end of the mantissa
        c = np.bincount(((W >> np.uint64(shift)) & np.uint64(255)).ravel().astype(np.int64), minlength=256).astype(np.float64)
        chi = ((c - n / 256) ** 2 / (n / 256)).sum()
        assert abs(chi - 255) / np.sqrt(2 * 255) < 4.5, (shift, chi)
    Z = (U - 0.5) * np.sqrt(12)
    corr = (Z.T @ Z) / n_keys                                                                # every pair of the 338 counters, over the keys
    np.fill_diagonal(corr, 0)
    assert np.abs(corr).max() * np.sqrt(n_keys) < 5.6                                        # (the maximum of 57 000 standard normals is ~4.7)
    a, b = Z[:-1].ravel(), Z[1:].ravel()                                                     # adjacent samples of the pixel, same counter
    assert abs((a * b).mean()) * np.sqrt(a.size) < 4.5
    i, j = (U[:, :-1] * 64).astype(np.int64), (U[:, 1:] * 64).astype(np.int64)               # consecutive counters: 64 x 64 cells
    c = np.bincount((i * 64 + j).ravel(), minlength=4096).astype(np.float64)
    chi = ((c - c.sum() / 4096) ** 2 / (c.sum() / 4096)).sum()
    assert abs(chi - 4095) / np.sqrt(2 * 4095) < 4.5
    # the unit-ball loop's acceptance rate: pi / 6 of the candidates of iteration 0 (slots 32, 33, 34 as the draw spec packs them)
    h, second, third = W[:, 6::26], W[:, 7::26], W[:, 8::26]
    f = [h >> np.uint64(43), (h >> np.uint64(22)) & np.uint64(0x1FFFFF), (h >> np.uint64(1)) & np.uint64(0x1FFFFF)]
    lo = [second >> np.uint64(32), second & np.uint64(0xFFFFFFFF), third >> np.uint64(32)]
    v = [2.0 * (((x << np.uint64(32)) | y).astype(np.float64) * 2.0 ** -53) - 1.0 for x, y in zip(f, lo)]
    acc = (v[0] ** 2 + v[1] ** 2 + v[2] ** 2) < 1.0
    p = np.pi / 6
    assert abs(acc.mean() - p) / np.sqrt(p * (1 - p) / acc.size) < 4.5
    for x in v:                                                                              # the accepted vectors are centred
        assert abs(x[acc].mean()) * np.sqrt(acc.sum() / 0.2) < 4.5
