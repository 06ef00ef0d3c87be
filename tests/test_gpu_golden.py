"""libsvx reproduces the committed oracle hashes (tests/golden/oracle_hashes.json) on the seeded cases, from 32x64 up to
the full 368x1232 frame at 128 disparities, for every routine on the path; the oracle is not run here."""
import json
import os

import pytest

import golden_cases as G

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_libsvx_reproduces_committed_hashes():
    import stereovision_amd as S
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_hashes.json")))["cases"]
    ctx = S.Context(0)
    try:
        got = G.hashes(G.GpuBackend(ctx))
    finally:
        ctx.close()
    assert sorted(got) == sorted(want)
    for size in want:
        for k in want[size]:
            assert got[size][k] == want[size][k], (size, k)
