"""csrc/lines.hip: the stream kernels that move the stage-1 generator's lines (row stacking for the output-row phases of a
ConvTranspose2d, its transpose, the phase interleave) against the torch slicing they replaced -- pure data movement, so the
forward kernels are compared bitwise; the fold sums at most four terms in a fixed order."""
import numpy as np
import pytest

from conftest import stable_seed

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

PHASES = {"s2": [[0, -1], [1, 0]], "s1": [[1, 0, -1]]}


def _ref_stack(x, phases):
    B, H, C, W = x.shape
    xp = torch.nn.functional.pad(x, (0, 0, 0, 0, 1, 1))
    return torch.stack([torch.cat([xp[:, 1 + dy:1 + dy + H] for dy in taps], dim=2).reshape(B * H, len(taps) * C, W)
                        for taps in phases])


@pytest.mark.parametrize("geom", ["s2", "s1"])
@pytest.mark.parametrize("shape", [(2, 5, 24, 8), (1, 4, 64, 4), (3, 1, 8, 12), (2, 128, 6, 64)], ids=lambda s: "x".join(map(str, s)))
def test_stack_and_fold(geom, shape):
    from featuresynth._ops import prims as P
    phases = PHASES[geom]
    rng = np.random.default_rng(stable_seed("lines%s%s" % (geom, shape)))
    x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda()
    d = P.lines_desc(x.shape, phases)
    out = P.lines_stack(x, d)
    xr = x.clone().requires_grad_(True)
    ref = _ref_stack(xr, phases)
    assert tuple(out.shape) == tuple(ref.shape)
    assert torch.equal(out, ref.detach())
    g = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32)).cuda()
    gx = P.lines_fold(g, d)
    ref.backward(g)
    assert tuple(gx.shape) == tuple(x.shape)
    err = float((gx - xr.grad).norm() / xr.grad.norm())
    assert err < 1e-6, err
    # the fold is the exact transpose of the stack: <stack(x), g> == <x, fold(g)>
    lhs, rhs = float((out.double() * g.double()).sum()), float((x.double() * gx.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))


@pytest.mark.parametrize("rows,phases,n", [(10, 2, 24), (512, 2, 128 * 64), (7, 1, 8)])
def test_interleave_round_trip(rows, phases, n):
    from featuresynth._ops import prims as P
    src = torch.arange(phases * rows * n, dtype=torch.float32, device="cuda").reshape(phases, rows, n)
    il = P.lines_interleave(src, rows, phases, n).reshape(rows, phases, n)
    assert torch.equal(il, src.permute(1, 0, 2).contiguous())
    back = P.lines_interleave(il, rows, phases, n, inverse=True).reshape(phases, rows, n)
    assert torch.equal(back, src)


def test_bad_geometry_is_refused():
    from featuresynth._ops import prims as P
    x = torch.zeros((1, 2, 3, 5), device="cuda")          # C * W = 15: not a multiple of 4
    with pytest.raises(RuntimeError):
        P.lines_stack(x, P.lines_desc(x.shape, PHASES["s2"]))
    with pytest.raises(RuntimeError):
        P.lines_desc((1, 2, 4, 4), [[0, 1, 2, 3]])
