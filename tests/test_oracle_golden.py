"""The oracle reproduces its committed hashes (tests/golden/oracle_hashes.json) on the seeded cases, up to the full frame size."""
import json
import os

import golden_cases as G


def test_oracle_reproduces_committed_hashes():
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_hashes.json")))["cases"]
    got = G.hashes(G.OracleBackend())
    assert sorted(got) == sorted(want)
    for size in want:
        assert got[size] == want[size], size
