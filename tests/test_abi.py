"""The C-ABI library loads and exports every symbol include/msynth.h declares (no compute
calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "msynth.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import torch  # noqa: F401  (the library shares torch's HIP runtime)
    from featuresynth._ops import lib as L
    if not os.path.exists(L.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "music-synthesis_amd", "csrc")])
    return L


def test_every_declared_symbol_is_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    raw = ctypes.CDLL(lib.LIB_PATH)
    for name in syms:
        assert hasattr(raw, name), "libmsynth_hip.so does not export %s" % name
        assert name in lib.SIGNATURES, "ctypes binding missing for %s" % name
    assert set(lib.SIGNATURES) == set(syms)


def test_integration_md_lists_every_entry_point():
    """INTEGRATION.md's table (entry point -> the reference interface it replaces) covers every symbol the header declares,
    spelled out, through the table's `ms_stem_a / _b` and `(+ _suffix)` shorthands, or through a `ms_*_suffix` row."""
    lines = open(os.path.join(ROOT, "INTEGRATION.md")).read().splitlines()

    def covered(sym):
        parts = sym.split("_")
        for ln in lines:
            if sym in ln:
                return True
            for i in range(2, len(parts)):
                stem, rest = "_".join(parts[:i]), "_" + "_".join(parts[i:])
                if ("ms_*" + rest) in ln:
                    return True
                if re.search("`" + re.escape(stem) + r"[a-z0-9_]*\b", ln) and (rest in ln or (stem + "_*") in ln):
                    return True
        return False

    missing = [s for s in declared_symbols() if not covered(s)]
    assert not missing, "INTEGRATION.md does not mention %s" % missing


def test_host_side_queries(lib):
    L = lib.load()
    assert L.ms_version() == 200
    assert b"ok" == L.ms_status_string(0)
    assert b"unsupported" in L.ms_status_string(-2) or b"not supported" in L.ms_status_string(-2)
    # geometry of the hot-path layers (no kernel is launched)
    d = lib.ConvDesc(32, 16, 8192, 64, 41, 4, 20, 1, 4, 0, 1, 0.2, 0)
    assert L.ms_conv1d_out_len(d) == 2048
    d = lib.ConvDesc(1, 1, 8192, 16, 15, 1, 7, 1, 1, 0, 1, 0.2, 0)
    assert L.ms_conv1d_out_len(d) == 8192
    bad = lib.ConvDesc(1, 6, 10, 4, 3, 1, 1, 1, 4, 0, 0, 0.2, 0)      # 6 channels, 4 groups
    assert L.ms_conv1d_out_len(bad) < 0
    t = lib.ConvTDesc(32, 512, 32, 256, 16, 8, 4, 1, 0.2, 0)
    assert L.ms_convt1d_out_len(t) == 256
    t = lib.ConvTDesc(1, 64, 4096, 32, 4, 2, 1, 1, 0.2, 0)
    assert L.ms_convt1d_out_len(t) == 8192
    assert L.ms_audio2mel_frames(22050, 1024, 256) == 84
    assert L.ms_audio2mel_frames(100, 1024, 256) == 0
    # r04 predicates (host arithmetic only): whole stacks in one launch at 32 / 64 channels, sign words where forward, backward
    # data and the batched weight gradients all take them
    def stack(B, C, Lg, dils=(1, 3, 9)):
        d = lib.StackDesc()
        d.B, d.C, d.L, d.count, d.slope = B, C, Lg, len(dils), 0.2
        for i, v in enumerate(dils):
            d.dil[i] = v
        return d
    assert L.ms_residual_stack_supported(stack(32, 64, 4096)) == 1 and L.ms_residual_stack_supported(stack(1, 32, 8192)) == 1
    assert L.ms_residual_stack_supported(stack(32, 128, 2048)) == 0 and L.ms_residual_stack_supported(stack(2, 64, 64, (9, 3, 1))) == 0
    assert L.ms_residual_atom_sign_words(lib.AtomDesc(32, 64, 4096, 3, 0.2)) == 32 * 2 * 2 * 4096
    for B, C, Lg in ((32, 32, 8192), (32, 64, 4096), (32, 128, 2048), (32, 256, 256)):
        assert L.ms_residual_stack_signs_supported(stack(B, C, Lg)) == 1, (B, C, Lg)
    assert L.ms_residual_stack_signs_supported(stack(2, 64, 300)) == 0          # rows the batched weight gradient does not take
    assert L.ms_reduce_workspace_bytes(10) >= 4
    # the dispatch names a kernel for every hot-path geometry
    for args, which in (((32, 128, 2048, 128, 3, 1, 3, 3, 1, 0, 1, 0.2, 0), 0),
                        ((32, 1024, 32, 1024, 5, 1, 2, 1, 1, 0, 1, 0.2, 0), 1),
                        ((32, 256, 512, 1024, 41, 4, 20, 1, 64, 0, 1, 0.2, 0), 2)):
        assert L.ms_conv1d_kernel_name(lib.ConvDesc(*args), which)
    # null pointers are rejected before anything is launched
    assert L.ms_conv1d_fwd(d, None, None, None, None, None, None, None, 0, None) == -1
    assert L.ms_adam_step(None, None, None, None, 4, 1e-4, 0.5, 0.9, 1e-8, 1.0, None, None) == -1


def test_discriminator_layers_are_one_launch_per_pass(lib):
    """r05: the reference applies ONE FullDiscriminator to x, pool(x), pool(pool(x))
    (discriminator/melgan.py:13-27) and the trainers run it on [fake; real]: a layer's three scales are one piece of work.
    ms_conv1d_parts_launches (host arithmetic; the placeholder addresses are never dereferenced) must plan ONE launch per
    layer and pass -- forward, backward data, weight gradient -- at the bench geometry: the 8192-sample window at B = 64
    ([fake; real] of the D-step / G-step forward) and B = 32 (the G-step's backward over the fake half)."""
    L = lib.load()

    def out_len(Lin, K, s, p):
        return (Lin + 2 * p - K) // s + 1

    # (Cin, Cout, K, stride, pad, groups) of FullDiscriminator.main and the judge conv (discriminator/full.py:13-22)
    layers = [(1, 16, 15, 1, 7, 1), (16, 64, 41, 4, 20, 4), (64, 256, 41, 4, 20, 16), (256, 1024, 41, 4, 20, 64),
              (1024, 1024, 41, 4, 20, 256), (1024, 1024, 5, 1, 2, 1), (1024, 1, 3, 1, 1, 1)]
    for B in (64, 32):
        Ls = [8192, 4097, 2049]                      # AvgPool1d(4, 2, padding=2) twice
        for li, (ci, co, K, s, p, g) in enumerate(layers):
            d = lib.ConvDesc(B, ci, Ls[0], co, K, s, p, 1, g, 0, 1 if li < 6 else 0, 0.2, 0)
            parts = lib.ConvParts()
            parts.count = 3
            for i in range(3):
                parts.B[i], parts.Lin[i] = B, Ls[i]
                for f, base in (("x", 0x10000000), ("y", 0x20000000), ("gy", 0x30000000), ("y_act", 0x40000000), ("gx", 0x50000000)):
                    getattr(parts, f)[i] = base + 0x1000000 * i
            for which in (0, 1, 2):
                n = L.ms_conv1d_parts_launches(d, parts, which, 1 if li == 5 else 0)
                assert n == 1, "layer %d, pass %d at B = %d: %d launches" % (li, which, B, n)
            Ls = [out_len(v, K, s, p) for v in Ls]
        assert Ls == [32, 17, 9]
    # a single tensor is not a parts call, and rows the parts kernels do not take fall back part by part (never an error)
    d = lib.ConvDesc(2, 16, 8000, 64, 41, 4, 20, 1, 4, 0, 1, 0.2, 0)
    parts = lib.ConvParts()
    parts.count = 3
    for i, v in enumerate((8000, 4001, 2001)):
        parts.B[i], parts.Lin[i] = 2, v
        parts.x[i] = parts.y[i] = 0x10000000 + 0x1000000 * i
    assert L.ms_conv1d_parts_launches(d, parts, 0, 0) in (1, 3)
    parts.count = 0
    assert L.ms_conv1d_parts_launches(d, parts, 0, 0) < 0


def test_struct_layout(lib):
    assert ctypes.sizeof(lib.ConvDesc) == 13 * 4
    assert ctypes.sizeof(lib.ConvTDesc) == 10 * 4
    assert ctypes.sizeof(lib.L1MultiDesc) == 8 + 24 * (8 + 8 + 8 + 8 + 4)     # ms_l1_multi_desc
    assert ctypes.sizeof(lib.WnMultiDesc) == 8 + 64 * (5 * 8 + 2 * 4)         # ms_wn_multi_desc
    assert ctypes.sizeof(lib.WgradMultiDesc) == 8 + 8 * (13 * 4 + 5 * 8 + 4) + 8 * (3 * 8)    # ms_wgrad_multi_desc (+ xmax, gmax, y_signs)
    assert ctypes.sizeof(lib.JudgeMultiDesc) == 8 + 8 * (4 * 8 + 8)               # ms_judge_multi_desc
    assert ctypes.sizeof(lib.AtomDesc) == 5 * 4                                   # ms_atom_desc
    assert ctypes.sizeof(lib.AtomPackDesc) == 8 + 16 * (4 + 3 * 8 + 4)               # ms_atom_pack_desc
    assert ctypes.sizeof(lib.StackDesc) == 8 * 4                                  # ms_stack_desc
