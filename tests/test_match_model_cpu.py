"""The decision logic of the round-5 device matcher (tests/match_model.py) against the oracle's MatchFlannFGINN restatement
(matching/matching.cpp:357-461) on the inputs of the GPU parity tests: exact ties, duplicated trains, clustered
near-duplicates, several splits, tiny problems.  CPU only."""
import numpy as np
import pytest

from tests import match_model as M


def _cmp(oracle, d1, d2, pos2, ratio, cd, nn=50, K=4, S=None):
    ref = oracle.match_fginn(d1, d2, pos2, ratio, cd, nn)
    got = M.rows_to_tentatives(M.match_rows(d1, d2, pos2, ratio, cd, nn, K=K, S=S), nn)
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert g[:4] == (r["q"], r["t0"], r["tj"], r["t1"])
        assert g[4] == r["d1"] and g[5] == r["d2"] and g[6] == r["d2by2ndcl"]
        assert g[7] == r["ratio"] or (np.isnan(g[7]) and np.isnan(r["ratio"]))


def test_model_ties_duplicates_ragged(oracle):
    rs = np.random.RandomState(9)
    for n1, n2 in ((1, 50), (33, 95), (70, 257), (5, 64)):
        d1 = rs.randint(0, 3, (n1, 128)).astype(np.float32) * 40
        d2 = rs.randint(0, 3, (n2, 128)).astype(np.float32) * 40
        d2[n2 // 2:] = d2[: n2 - n2 // 2]
        d1[0] = d2[3]
        pos2 = rs.uniform(0, 60, (n2, 2))
        for ratio, cd in ((0.8, 30.0), (0.95, 80.0), (0.8, 5.0)):
            for K in (4, 6):
                _cmp(oracle, d1, d2, pos2, ratio, cd, K=K)


def test_model_odd_parity_ties(oracle):
    """values +-1 so that both parity classes tie against each other by one unit and inside themselves exactly"""
    rs = np.random.RandomState(3)
    n1, n2 = 40, 700
    base = rs.randint(0, 4, (1, 128))
    d2 = np.clip(base + (rs.rand(n2, 128) < 0.01), 0, 255).astype(np.float32)
    d1 = np.clip(base + (rs.rand(n1, 128) < 0.01), 0, 255).astype(np.float32)
    pos2 = rs.uniform(0, 40, (n2, 2))
    for ratio, cd, nn in ((0.8, 30.0, 50), (0.99, 100.0, 20), (0.9, 10.0, 256)):
        _cmp(oracle, d1, d2, pos2, ratio, cd, nn, S=3)


def test_model_clustered_runs_and_splits(oracle):
    rs = np.random.RandomState(31)
    for n1, n2, kmax in ((90, 2100, 40), (40, 5000, 25)):
        d2 = rs.randint(0, 90, (n2, 128)).astype(np.float32)
        d1 = rs.randint(0, 90, (n1, 128)).astype(np.float32)
        pos2 = rs.uniform(0, 2000, (n2, 2))
        for q in range(0, n1, 3):
            k = int(rs.randint(2, kmax))
            start = int(rs.randint(0, n2 - k))
            d2[start:start + k] = np.clip(d1[q][None, :] + rs.randint(-2, 3, (k, 128)), 0, 255)
            pos2[start:start + k] = pos2[start] + rs.uniform(-3, 3, (k, 2))
            if q % 6 == 0:
                d2[start + k - 1] = d2[start]
        for ratio, cd, nn in ((0.8, 30.0, 50), (0.9, 30.0, 20), (0.8, 2.0, 50)):
            _cmp(oracle, d1, d2, pos2, ratio, cd, nn)


def test_model_pdf_mode(oracle):
    """ratio >= 1 (matching.cpp:397-428): a record per query, closed by its first contradictive neighbour or by neighbour nn - 1"""
    rs = np.random.RandomState(5)
    n1, n2 = 60, 1500
    d2 = rs.randint(0, 90, (n2, 128)).astype(np.float32)
    d1 = rs.randint(0, 90, (n1, 128)).astype(np.float32)
    pos2 = rs.uniform(0, 300, (n2, 2))
    for q in range(0, n1, 2):          # runs of near-duplicates at one place: walks that go deep
        k = int(rs.randint(2, 30))
        start = int(rs.randint(0, n2 - k))
        d2[start:start + k] = np.clip(d1[q][None, :] + rs.randint(-2, 3, (k, 128)), 0, 255)
        pos2[start:start + k] = pos2[start] + rs.uniform(-3, 3, (k, 2))
    for ratio, cd, nn in ((1.0, 30.0, 50), (1.5, 10.0, 8), (1.0, 500.0, 12)):
        _cmp(oracle, d1, d2, pos2, ratio, cd, nn)
