"""The committed oracle fixtures (tests/golden/oracle/, tests/helpers.py) answer for the oracle THAT IS IN THE TREE: their keys hold
a hash of the oracle's sources, INDEX.txt names it, and every day two of them are recomputed by the live oracle here (CPU) and
compared -- a fixture that no longer is what the oracle says fails the CPU suite, not silently the GPU suite's meaning."""
import datetime
import hashlib
import os

import numpy as np

import helpers
import oracle_ffi as O


def _index():
    rows = []
    with open(os.path.join(helpers.ORACLE_CACHE_DIR, "INDEX.txt")) as fh:
        for line in fh:
            parts = line.split()
            if parts:
                rows.append(parts)
    return rows


def test_fixtures_were_written_by_the_oracle_in_the_tree():
    """every fixture is listed with the hash of the oracle sources that wrote it, and that hash is the tree's: after a change to
    oracle/*.c, jtk_oracle.h, the Makefile's flags, include/jtk_math.h or jtk_eigen.h, regenerate (tests/golden/make_oracle_cache.py)"""
    rows = _index()
    assert len(rows) >= 20
    sha = "oracle=" + helpers.oracle_sources_sha()[:16]
    stale = [r for r in rows if len(r) < 3 or r[2] != sha]
    assert not stale, "fixtures of another oracle: run tests/golden/make_oracle_cache.py (%s)" % stale[:3]
    files = {f[:-4] for f in os.listdir(helpers.ORACLE_CACHE_DIR) if f.endswith(".npz")}
    assert files == {r[0] for r in rows}


def test_two_fixtures_a_day_against_the_live_oracle(oracle):
    """two of the eight plain batches of the random shape sweep, picked by the date: the live oracle's answer (the function behind
    the fixture wrapper) must be the stored one, field for field, bit for bit"""
    plain = [it for it in range(10) if it % 5 != 4]            # (the recursive-split batches take a minute each)
    day = datetime.date.today().isoformat().encode()
    h = int.from_bytes(hashlib.sha256(day).digest()[:8], "little")
    first = h % len(plain)
    second = (first + 1 + (h >> 8) % (len(plain) - 1)) % len(plain)                # (never the same as the first)
    pick = {plain[first], plain[second]}
    assert len(pick) == 2
    live_fn = getattr(O.cluster_chunks, "_orig", None)
    assert live_fn is not None, "conftest did not install the fixture wrapper"
    for it, b, p in helpers.shape_sweep_inputs(only=pick):
        po = helpers.oracle_params(p)
        key = helpers._cache_key([b"cluster_chunks", b.chunks, b.tmpl_bases, b.read_bases, b.read_off, b.ops, b.ops_off,
                                  b.strand, bytes(po), bytes([0])])
        fix = helpers.oracle_cache_get(key)
        assert fix is not None, "no fixture for sweep batch %d (key %s)" % (it, key)
        live = live_fn(po, b, skip_polish=False)
        assert int(live["rc"]) == int(fix["rc"])
        n, m = int(live["cons_off"][-1]), int(live["ops_out_off"][-1])
        assert np.array_equal(live["label"], fix["label"])
        assert live["result"].tobytes() == fix["result"].tobytes()
        assert np.array_equal(helpers.bits(live["log_post"]), helpers.bits(fix["log_post"]))
        assert np.array_equal(live["cons_off"], fix["cons_off"]) and bytes(live["cons"][:n]) == bytes(fix["cons"])
        assert np.array_equal(live["ops_out_off"], fix["ops_out_off"]) and np.array_equal(live["ops_out"][:m], fix["ops_out"])
