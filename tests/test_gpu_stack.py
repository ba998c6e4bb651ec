"""Fused ResidualStack forward, inference (csrc/stack_fused.hip: the three atoms of a generator stack, dilations 1 / 3 / 9,
in one launch) -- reference util/modules.py:391-405 -- against float64 torch, against the three fused-atom launches it
replaces and against the CPU oracle: one-tile rows, rows that end inside a tile, rows shorter than the halo, several batch
rows, the generator's own row lengths at B = 32, input scales from 1e-12 to 1e+6 (block scaling), and through
gen_forward(save=False), which must take the stack kernel and agree with the training-mode forward."""
import numpy as np
import pytest

from conftest import rel_l2, stable_seed

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

DILS = (1, 3, 9)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _stack_inputs(name, B, C, Lg, n=3, scale=1.0):
    rng = np.random.default_rng(stable_seed(name))
    x = (rng.standard_normal((B, C, Lg)) * scale).astype(np.float32)
    sc = 1.0 / np.sqrt(3 * C)
    ws = [((rng.standard_normal((C, C, 3)) * sc).astype(np.float32), (rng.standard_normal((C,)) * 0.1 * scale).astype(np.float32),
           (rng.standard_normal((C, C, 3)) * sc).astype(np.float32), (rng.standard_normal((C,)) * 0.1 * scale).astype(np.float32))
          for _ in range(n)]
    return x, ws


def _f64_stack(x, ws, dils):
    import torch.nn.functional as F
    h = x.double()
    for (w0, b0, w1, b1), d in zip(ws, dils):
        t = F.leaky_relu(F.conv1d(h, w0.double(), b0.double(), padding=d, dilation=d), 0.2)
        h = h + F.leaky_relu(F.conv1d(t, w1.double(), b1.double(), padding=1), 0.2)
    return h


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def _run(x, ws, dils):
    from featuresynth._ops import graph as G
    from featuresynth._ops import prims as P
    xt = dev(x)
    wt = [tuple(dev(a) for a in w) for w in ws]
    imgs = [P.atom_image(x.shape[1], xt.device) for _ in wt]
    P.atom_pack([(w[0], w[2], im) for w, im in zip(wt, imgs)])
    assert P.stack_supported(xt, dils), "the stack kernel must take the generator's geometries"
    y = P.stack_fwd(xt, imgs, [w[1] for w in wt], [w[3] for w in wt], dils)
    h = xt
    for w, im, d in zip(wt, imgs, dils):
        h, _ = G.atom_forward(h, w[0], w[1], w[2], w[3], d, False, image=im)
    return xt, wt, y, h


# name, B, C, L      (a tile stores 96 columns of a 128-column frame)
STACK_CASES = [("c32_one_tile", 1, 32, 96), ("c32_ragged", 2, 32, 1000), ("c32_tiny", 3, 32, 8), ("c32_halo_sized", 1, 32, 20),
               ("c32_rows", 7, 32, 292), ("c64_two_tiles", 1, 64, 192), ("c64_tail4", 2, 64, 196), ("c64_rows", 5, 64, 516),
               ("c64_one", 1, 64, 96), ("c64_ragged", 2, 64, 300)]


@pytest.mark.parametrize("case", STACK_CASES, ids=[c[0] for c in STACK_CASES])
def test_fused_stack_vs_float64_atoms_and_oracle(case):
    from oracle import oracle as O
    name, B, C, Lg = case
    x, ws = _stack_inputs(name, B, C, Lg)
    xt, wt, y, y3 = _run(x, ws, DILS)
    yr = _f64_stack(xt, wt, DILS)
    e, e3 = _rel(y, yr), _rel(y3, yr)
    print("%s: stack vs float64 %.2e, three atoms vs float64 %.2e, stack vs atoms %.2e" % (name, e, e3, _rel(y, y3)))
    assert e < 1e-6 and e < 3 * e3 + 1e-7
    assert _rel(y, y3) < 2e-6
    h = x
    for (w0, b0, w1, b1), d in zip(ws, DILS):
        t = O.conv1d_fwd(h, w0, b0, 1, d, d, 1, O.PAD_ZERO, 1)
        h = h + O.conv1d_fwd(t, w1, b1, 1, 1, 1, 1, O.PAD_ZERO, 1)
    assert rel_l2(y.cpu().numpy(), h) < 1e-5


@pytest.mark.parametrize("C,Lg", [(32, 8192), (64, 4096)])
def test_fused_stack_at_bench_shapes(C, Lg):
    """BASELINE config 3's shapes (B = 32): float64 accuracy, agreement with the three atom launches, determinism."""
    x, ws = _stack_inputs("bench_stack_%d" % C, 32, C, Lg)
    xt, wt, y, y3 = _run(x, ws, DILS)
    yr = _f64_stack(xt, wt, DILS)
    e, e3 = _rel(y, yr), _rel(y3, yr)
    print("C=%d: stack vs float64 %.2e, three atoms %.2e" % (C, e, e3))
    assert e < 1e-6 and e < 3 * e3 + 1e-7 and _rel(y, y3) < 2e-6
    _, _, y2, _ = _run(x, ws, DILS)
    assert torch.equal(y, y2), "launch-to-launch determinism"


@pytest.mark.parametrize("scale", [1e-12, 1e-6, 1e6])
def test_fused_stack_block_scaling(scale):
    """Per-tile, per-operand power-of-two scales: the relative error does not depend on the magnitude of the data."""
    x, ws = _stack_inputs("stack_scale", 2, 64, 1000, scale=scale)
    xt, wt, y, y3 = _run(x, ws, DILS)
    yr = _f64_stack(xt, wt, DILS)
    assert _rel(y, yr) < 1e-6, (scale, _rel(y, yr))


def test_fused_stack_partial_and_switch(monkeypatch):
    """One and two atoms per launch; dilation sets whose halo does not fit are refused; MSYNTH_STACK=0 switches it off."""
    from featuresynth._ops import prims as P
    x, ws = _stack_inputs("stack_partial", 2, 32, 500, n=2)
    xt, wt, y, y3 = _run(x, ws, (3, 1))
    assert _rel(y, _f64_stack(xt, wt, (3, 1))) < 1e-6 and _rel(y, y3) < 2e-6
    xt, wt, y, y3 = _run(x, ws[:1], (2,))
    assert _rel(y, _f64_stack(xt, wt, (2,))) < 1e-6
    assert not P.stack_supported(xt, (9, 3, 1))         # first window would need 9 columns beyond the frame
    assert not P.stack_supported(xt, (3, 9, 9))         # halo 4 + 10 + 10 > 16
    assert not P.stack_supported(dev(np.zeros((1, 48, 64))), DILS)
    assert not P.stack_supported(dev(np.zeros((1, 128, 64))), DILS)     # 128 / 256 channels: per atom (measured slower fused)
    monkeypatch.setenv("MSYNTH_STACK", "0")
    assert not P.stack_supported(xt, DILS)


def test_generator_inference_takes_the_stack_kernel(monkeypatch):
    """gen_forward(save=False) -- the D-step's generator pass, BASELINE config 2 -- runs each ResidualStack as ONE launch
    where the kernel takes the width, and gives what the training-mode forward (one launch per atom) gives."""
    import featuresynth as fs
    from featuresynth._ops import graph as G
    from featuresynth._ops import lib as L
    from featuresynth._synthetic import module_param_shapes, synthetic_features, synthetic_state_dict
    g = fs.MelGanGenerator(32, 80)
    sd = synthetic_state_dict(module_param_shapes(g), seed=7)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.cuda()
    params = list(g.parameters())
    for B in (1, 4):
        f = torch.from_numpy(synthetic_features(B, 80, 32, rank=3)).cuda()
        L.profile_begin()
        y, _ = G.gen_forward(f, params, save=False)
        rec = L.profile_end()
        names = [c.get("kernel") or n for n, c, _ in rec]
        assert sum(n.startswith("k_stack_fwd") for n in names) == 2, names           # 64 and 32 channels
        assert sum(n.startswith("k_atom_fwd") for n in names) == 6, names            # the 256- / 128-channel stacks: per atom
        yt, _ = G.gen_forward(f, params, save=True)
        assert _rel(y, yt) < 2e-6
        monkeypatch.setenv("MSYNTH_STACK", "0")
        y0, _ = G.gen_forward(f, params, save=False)
        monkeypatch.delenv("MSYNTH_STACK")
        assert torch.equal(y0, yt), "per-atom inference == training forward (bitwise)"
