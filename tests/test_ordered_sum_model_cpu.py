"""The exact parallel evaluation of a sequential f32 sum of non-negative terms (tests/ordered_sum_model.py: groundwork for
k_describe's photometric normalisation) against the plain sequential loop, bit for bit."""
import numpy as np

from ordered_sum_model import parallel_sum, sequential_sum


def _cases(rs, n_cases):
    for case in range(n_cases):
        n = int(rs.choice([1, 2, 5, 64, 300, 1257, 1257]))
        kind = case % 7
        if kind == 0: x = rs.uniform(0, 255, n)                    # a patch's pixels
        elif kind == 1: x = rs.uniform(0, 255, n) ** 2             # squared deviations
        elif kind == 2: x = np.floor(rs.uniform(0, 256, n))        # integers: ties at every binade above 2^24 ulps ...
        elif kind == 3: x = np.floor(rs.uniform(0, 1024, n)) / 8   # ... and multiples of 1/8
        elif kind == 4: x = rs.choice([0, 0, 1e-3, 0.5, 1, 3, 1000.0, 65536.0], n)   # zeros and jumps over several binades
        elif kind == 5: x = rs.gamma(0.3, 50, n)
        else: x = np.abs(rs.normal(0, 1, n)) * rs.choice([1e-6, 1, 1e6])
        yield x.astype(np.float32)


def test_parallel_ordered_sum_equals_the_sequential_one():
    rs = np.random.RandomState(7)
    segs = []
    for x in _cases(rs, 700):
        st = {}
        a, b = sequential_sum(x), parallel_sum(x, st)
        assert a.tobytes() == b.tobytes(), (x[:8], a, b)
        segs.append(st.get("segments", 0))
    assert max(segs) <= 40          # a few binades per sum: the parallel form is ~log2(n) passes, not n steps


def test_parallel_ordered_sum_ties_and_edges():
    f = np.float32
    for x in ([0, 0, 0], [1], [2 ** 24, 1, 1, 1, 1], [2 ** 24, 1, 2, 1, 3], [1, 2 ** 24 - 1, 1, 1], [0.5] * 100, [1.5, 2 ** 23, 0.5, 0.5, 1.5, 2.5],
              [3] * 3000, [2 ** -120, 2 ** -120, 1], [16777215, 0.5, 0.5, 1, 1, 2]):
        x = np.array(x, f)
        assert sequential_sum(x).tobytes() == parallel_sum(x).tobytes(), x
