"""An exact PARALLEL evaluation of a sequential f32 sum of non-negative terms -- the model of what k_describe's photometric
normalisation could run instead of its 2 x 1257 dependent adds.  Test infrastructure / groundwork: nothing
in the product uses it yet.

  s_0 = 0, s_i = fl(s_{i-1} + x_i)   (f32, round to nearest even), all x_i >= 0.

While the running sum stays in one binade [2^e, 2^(e+1)) its ulp u = 2^(e-23) is constant, s = m u with an integer m, and an
addition is s + round_u(x): the exact sum is rounded to a multiple of u, so with x = q u + r (0 <= r < u)
    r < u/2 : m += q        r > u/2 : m += q + 1        r = u/2 (a tie): m += q + ((m + q) odd)  -- the result is EVEN.
Because a tie leaves m even, the corrections of the ties need no sequential pass: with b'_i = q_i (+1 when r_i > u/2) and
cp_i = parity(m_0 + sum_{j<=i} b'_j) the correction of tie k is  c_k = cp_k xor cp_{previous tie}  (xor 0 for the first), and
    m_i = m_0 + prefix(b')_i + prefix(c)_i :
two prefix sums and a "value at the previous tie" scan.  The binade ends at the first term whose exact sum reaches 2^(e+1)
(m_{i-1} + q_i >= 2^24: that one addition is done in f32, its rounding unit is coarser) or whose rounded sum lands on it; the
tail is then re-evaluated with the new unit -- a sum that grows from its first terms to n of them crosses ~log2(n) binades, most
of them within the first few terms."""
import numpy as np


def sequential_sum(x):
    s = np.float32(0)
    for v in np.asarray(x, np.float32):
        s = np.float32(s + v)
    return s


def parallel_sum(x, stats=None):
    x = np.asarray(x, np.float32)
    assert (x >= 0).all() and np.isfinite(x).all()
    n = len(x)
    i0 = 0
    s = np.float32(0)
    segments = 0
    xd = x.astype(np.float64)
    while i0 < n:
        if s == 0 or not np.isfinite(s) or s < np.float32(2.0 ** -100):
            s = np.float32(s + x[i0]); i0 += 1      # no binade yet (zeros / the first term): one plain addition
            continue
        segments += 1
        e = int(np.floor(np.log2(np.float64(s))))
        if np.float64(s) >= 2.0 ** (e + 1): e += 1
        if np.float64(s) < 2.0 ** e: e -= 1
        u = 2.0 ** (e - 23)
        m0 = int(round(np.float64(s) / u))
        assert (1 << 23) <= m0 < (1 << 24) and m0 * u == np.float64(s)
        t = xd[i0:] / u                              # exact: a power-of-two scaling
        q = np.floor(t)
        r = t - q                                    # exact: t has at most 24 significant bits
        tie = r == 0.5
        bp = (q + (r > 0.5)).astype(np.int64)        # b': ties contribute q
        P = np.cumsum(bp)
        cp = (m0 + P) & 1
        # the correction of a tie: cp there xor cp at the previous tie (0 before the first)
        tidx = np.nonzero(tie)[0]
        c = np.zeros(len(bp), np.int64)
        if len(tidx):
            prev = np.concatenate([[0], cp[tidx[:-1]]])
            c[tidx] = cp[tidx] ^ prev
        m = m0 + P + np.cumsum(c)
        mprev = np.concatenate([[m0], m[:-1]])
        LIM = 1 << 24
        crossing = (mprev + q.astype(np.int64)) >= LIM          # the exact sum reaches the next binade: that addition is done in f32
        landed = m >= LIM                                       # the rounded sum is 2^(e+1) itself
        stop = np.nonzero(crossing | landed)[0]
        if len(stop) == 0:
            s = np.float32(m[-1] * u); i0 = n
            break
        k = int(stop[0])
        if crossing[k]:
            sk = np.float32(np.float32(mprev[k] * u) + x[i0 + k])
        else:
            sk = np.float32(m[k] * u)
        s = sk; i0 += k + 1
    if stats is not None:
        stats["segments"] = segments
    return s
