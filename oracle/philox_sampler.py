"""TEST INFRASTRUCTURE -- bit-exact numpy specification of the DEVICE negative sampler.

This is not a restatement of the reference (its sampler draws from numpy's global Mersenne state,
`src/neg_sampling.py:9,13`, which no device RNG can reproduce; `oracle/tip_oracle.py` restates
that one for distribution checks).  It pins `tipk_typed_negative_sampling` (include/tipk.h
section 5) so the HIP kernel can be tested for exact equality:

  position e of relation r (counter c = e + pos_offset[r]), attempt a = 0, 1, ...:

  n^2 < 2^32 (every graph whose node ids fit 16 bits; round 5) -- ONE Philox call serves FOUR positions:
      (x0, x1, x2, x3) = Philox4x32-10(counter = (q & 0xffffffff, q >> 32, a, 0), q = c >> 2,
                                       key = (seed & 0xffffffff, seed >> 32));   x = x[c & 3]
      m = x * n^2 (64 bits);  cand = m >> 32;  lo = m & 0xffffffff
      the draw is REJECTED if lo < (2^32 - n^2) mod n^2   (Lemire's test: what is left is exactly uniform on [0, n^2),
                                                           p(reject) < n^2 / 2^32 = 1e-4 at BioSNAP)
      or if cand is a positive key u*n+v of relation r; the first draw that is not rejected is kept (at most 64
      attempts, the 64th is kept regardless);  u = cand // n, v = cand % n.
      (Round 3 drew one 64-bit candidate per call: 20 wide multiplies per position, two thirds of the sampler's issue
      slots; now 5 + 1.)

  n^2 >= 2^32:
      (x0, x1, x2, x3) = Philox4x32-10(counter = (c & 0xffffffff, c >> 32, a, 0), key as above)
      cand = (x0 | x1 << 32) * n^2 >> 64,  rejected while it is a positive key of relation r, as above.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)
MAX_ATTEMPTS = 64


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 on uint32 arrays (arithmetic carried in uint64)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK32 for c in (c0, c1, c2, c3))
    k0, k1 = int(k0), int(k1)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def call_key(seed, n):
    """Philox key of call n of a sampler stream = splitmix64(seed + (n + 1) * golden); identical to
    `call_key` in tip_amd/csrc/tipk_negsample.hip and tip_amd/neg_sampling.py."""
    m = (1 << 64) - 1
    z = (seed + (n + 1) * 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def _mulhi64(a, b):
    """floor(a * b / 2^64) for uint64 arrays a and python int b < 2^64."""
    a_hi, a_lo = a >> np.uint64(32), a & MASK32
    b_hi, b_lo = np.uint64(b >> 32), np.uint64(b & 0xFFFFFFFF)
    t = a_lo * b_lo
    mid1 = a_hi * b_lo + (t >> np.uint64(32))
    mid2 = a_lo * b_hi + (mid1 & MASK32)
    return a_hi * b_hi + (mid1 >> np.uint64(32)) + (mid2 >> np.uint64(32))


def typed_negative_sampling_spec(pos_edge_index, num_nodes, rel_ptr, seed, pos_offset=None):
    """pos_edge_index: int64 [2, E] numpy; rel_ptr: [R+1]; returns int64 [2, E].
    pos_offset (optional, [R]): the Philox counter of position e of relation r is e + pos_offset[r]."""
    pos = np.asarray(pos_edge_index, dtype=np.int64)
    rel_ptr = np.asarray(rel_ptr, dtype=np.int64)
    n = int(num_nodes)
    E = pos.shape[1]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    out = np.zeros(E, dtype=np.uint64)
    for r in range(rel_ptr.size - 1):
        a, b = int(rel_ptr[r]), int(rel_ptr[r + 1])
        if a == b:
            continue
        keys = np.unique(pos[0, a:b] * n + pos[1, a:b]).astype(np.uint64)
        todo = np.arange(a, b, dtype=np.uint64)
        off = np.uint64(0 if pos_offset is None else int(pos_offset[r]))
        nn = n * n
        thresh = np.uint64(((1 << 32) - nn) % nn) if nn < (1 << 32) else None
        for attempt in range(MAX_ATTEMPTS):
            ctr = todo + off
            if thresh is not None:
                q = ctr >> np.uint64(2)
                xs = philox4x32_10(q & MASK32, q >> np.uint64(32), np.full(todo.size, attempt), 0, k0, k1)
                x = np.choose((ctr & np.uint64(3)).astype(np.int64), xs)
                m = x * np.uint64(nn)
                cand = m >> np.uint64(32)
                bad = (m & MASK32) < thresh
            else:
                x0, x1, _, _ = philox4x32_10(ctr & MASK32, ctr >> np.uint64(32), np.full(todo.size, attempt), 0, k0, k1)
                cand = _mulhi64(x0 | (x1 << np.uint64(32)), nn)
                bad = np.zeros(todo.size, dtype=bool)
            out[todo.astype(np.int64)] = cand
            bad = bad | np.isin(cand, keys)
            todo = todo[bad]
            if todo.size == 0:
                break
    out = out.astype(np.int64)
    return np.stack([out // n, out % n])
