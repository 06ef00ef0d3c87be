"""The allocation-free entry point svx_dense_census_wta_u8_noalloc against svx_dense_census_wta_u8, through the C ABI:
bit-equal disparity maps on two calls in a row, the second on a smaller input (the fixed scratch keeps the first call's
contents), then a batch at the size limit and the refusal beyond it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    import stereovision_amd as S
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    c = S.Context(0)
    yield c
    c.close()


def _pair(rows, cols, D, frames, seed):
    from stereovision_amd import synth
    ls, rs = synth.stereo_batch(rows, cols, D, frames, seed0=seed)
    return torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()


def test_noalloc_equals_the_allocating_entry_point_twice_in_a_row(ctx):
    for rows, cols, D, frames, seed in [(368, 1232, 128, 3, 5), (120, 200, 48, 2, 9), (37, 133, 200, 1, 2)]:   # large, then smaller
        left, right = _pair(rows, cols, D, frames, seed)
        want = ctx.dense_census_wta(left, right, D)
        got = torch.full_like(want, -5.0)
        before = torch.cuda.mem_get_info()[0]          # free device memory as the driver sees it: a hipMalloc in the library would lower it
        ctx.dense_census_wta_noalloc(left, right, D, got)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == before
        assert torch.equal(got, want), (rows, cols, (got != want).nonzero()[:5].tolist())


def test_noalloc_matches_the_oracle(ctx):
    import oracle_lib as O
    rows, cols, D = 48, 96, 24
    left, right = _pair(rows, cols, D, 1, 40)
    got = ctx.dense_census_wta_noalloc(left, right, D, torch.empty((1, rows, cols), dtype=torch.float32, device="cuda"))[0].cpu().numpy()
    l, r = left[0].cpu().numpy(), right[0].cpu().numpy()
    want = O.dense_wta5x5(O.census5x5(O.u8_to_s8(l)), O.census5x5(O.u8_to_s8(r)), D)
    assert np.array_equal(got, want)


def test_noalloc_size_limit(ctx):
    import stereovision_amd as S
    limit = 4194304                                    # SVX_NOALLOC_MAX_PIXELS
    rows, cols = 1024, 1024
    left = torch.randint(0, 256, (4, rows, cols), dtype=torch.uint8, device="cuda")
    right = torch.roll(left, -7, dims=2)
    assert left.numel() == limit
    out = torch.empty(left.shape, dtype=torch.float32, device="cuda")
    ctx.dense_census_wta_noalloc(left, right, 32, out)
    assert torch.equal(out, ctx.dense_census_wta(left, right, 32))
    big = torch.zeros((5, rows, cols), dtype=torch.uint8, device="cuda")
    with pytest.raises(S.SvxError) as e:
        ctx.dense_census_wta_noalloc(big, big, 32, torch.empty(big.shape, dtype=torch.float32, device="cuda"))
    assert e.value.code == -6 and "fixed scratch" in str(e.value)
