"""Jump-ahead polynomials of MT19937 for the segmented z generator (csrc/gs_zgen_device.hip: `mt_jump_kernel`).

A stream of the reference's latents is one MT19937 sequence per seed (models/wrappers.py:167-174), serial by construction.
To split ONE stream over several workgroups, segment i has to start from the generator's state after i * L blocks of 624
draws.  With phi the characteristic polynomial of the word recurrence (degree 19937) and

    c(x) = x^J mod phi(x)                      over GF(2),  J = i * L * 624,

the word sequence w_0, w_1, ... of a stream satisfies  w_{J+n} = XOR_{k : c_k = 1} w_{k+n}  for every n >= 0 - the state at
offset J is an XOR of windows of the first 19937 + 624 words (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer: "Efficient
jump ahead for F2-linear random number generators", INFORMS J. Comput. 2008; the windows-of-the-output form used here needs
no Horner scheme over states).  The polynomials depend on J alone, not on the seed.

This script derives phi from the generator itself (Berlekamp-Massey on one output bit plane of NumPy's MT19937), computes
c^(i) for i = 1 .. SEGMENTS - 1, checks each of them against NumPy (the state RandomState reaches after J draws) and writes

    ganspace_amd/data/mt19937_jump_L<L>.npz     polys: uint32 [SEGMENTS - 1, 624] (bit k of the polynomial = bit k % 32 of
                                                 word k // 32), block_len = L, phi: uint32 [624]

    python tools/make_mt_jump.py [L=2048] [SEGMENTS=64]            (about a minute of pure Python big-integer arithmetic)
"""
import os
import sys

import numpy as np

N, DEG = 624, 19937


def word_sequence(seed, blocks):
    """The first `blocks` * 624 untempered state words a RandomState(seed) draws from (block 1 = the first regenerated
    state; the seeded state itself is never drawn from)."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(blocks):
        rs.random_sample(N // 2)                  # 312 doubles = 624 draws = one block
        out.append(np.array(rs.get_state()[1], dtype=np.uint32))
    return np.concatenate(out)


def berlekamp_massey(bits):
    """Connection polynomial C (int: bit i = coefficient of x^i, C_0 = 1) and linear complexity of a GF(2) sequence:
    s_n = XOR_{i=1..L} C_i s_{n-i}."""
    C, B, L, m, win = 1, 1, 0, 1, 0
    for n, s in enumerate(bits):
        win = (win << 1) | s                      # bit i of win = s_{n-i}
        if (C & win).bit_count() & 1:             # discrepancy
            T = C
            C ^= B << m
            if 2 * L <= n:
                L, B, m = n + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    return C, L


def reverse_bits(p, nbits):
    return int(bin(p)[2:].zfill(nbits)[::-1], 2)


def square(p):
    return int("0".join(bin(p)[2:]), 2)           # b_k ... b_0 -> b_k 0 ... 0 b_0: sum b_i x^(2i)


def reduce_mod(p, phi):
    d = phi.bit_length() - 1
    while True:
        k = p.bit_length() - 1
        if k < d:
            return p
        p ^= phi << (k - d)


def mulmod(a, b, phi):
    acc = 0
    while b:
        acc ^= a << ((b & -b).bit_length() - 1)
        b &= b - 1
    return reduce_mod(acc, phi)


def powx(J, phi):
    """x^J mod phi."""
    r = 1
    for bit in bin(J)[2:]:
        r = reduce_mod(square(r), phi)
        if bit == "1":
            r = reduce_mod(r << 1, phi)
    return r


def poly_words(p):
    return np.frombuffer(p.to_bytes(N * 4, "little"), dtype="<u4").astype(np.uint32)


def apply_jump(words, poly):
    """State block at the polynomial's offset: XOR of the windows words[k : k + 624] over the set bits k."""
    bits = np.unpackbits(poly.view(np.uint8), bitorder="little")[:DEG]
    ks = np.nonzero(bits)[0]
    out = np.zeros(N, dtype=np.uint32)
    for lo in range(0, len(ks), 2048):
        idx = ks[lo:lo + 2048, None] + np.arange(N)[None, :]
        out ^= np.bitwise_xor.reduce(words[idx], axis=0)
    return out


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    segments = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    need_blocks = (DEG + N + N - 1) // N          # 33 blocks hold every window
    w = word_sequence(5489, 2 * DEG // N + 2)
    C, lin = berlekamp_massey([int(v) & 1 for v in w[:2 * DEG + 64]])
    assert lin == DEG, lin
    phi = reverse_bits(C, DEG + 1)                # characteristic polynomial: s_{n+L} = XOR_{j<L} phi_j s_{n+j}
    assert phi >> DEG == 1 and phi & 1
    print(f"phi: degree {phi.bit_length() - 1}, weight {phi.bit_count()}", flush=True)

    step = powx(L * N, phi)
    polys, cur = [], 1
    for i in range(1, segments):
        cur = mulmod(cur, step, phi)
        polys.append(poly_words(cur))
        if i <= 3 or i == segments - 1:           # against NumPy, two seeds
            for seed in (12345, 2 ** 31 - 7):
                words = word_sequence(seed, need_blocks)
                rs = np.random.RandomState(seed)
                rs.random_sample((i * L + 1) * (N // 2))
                want = np.array(rs.get_state()[1], dtype=np.uint32)
                got = apply_jump(words, polys[-1])
                assert np.array_equal(got, want), (i, seed)
            print(f"segment {i}: x^{i * L * N} mod phi, weight {cur.bit_count()}: matches NumPy", flush=True)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ganspace_amd", "data")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, f"mt19937_jump_L{L}.npz")
    np.savez_compressed(path, polys=np.stack(polys), block_len=np.int64(L), phi=poly_words(phi ^ (1 << DEG)))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
