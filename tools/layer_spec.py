#!/usr/bin/env python3
"""Prints the algorithmic work of the native D-step / G-step (per 8192-sample batch element
and per call at B=32) from the layer spec in featuresynth/_workload.py."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location(
    "w", os.path.join(ROOT, "music-synthesis_amd", "featuresynth", "_workload.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)

HBM, F32PEAK = 8.0e12, 157.3e12
for B in (1, 32):
    d, g = W.d_step_launches(B), W.g_step_launches(B)
    td, tg = W.totals(d), W.totals(g)
    print("B=%d  D-step: %.2f GFLOP %.1f MB (%d launches) | G-step: %.2f GFLOP %.1f MB (%d launches)" % (
        B, td["flops"] / 1e9, td["bytes"] / 1e6, len(d), tg["flops"] / 1e9, tg["bytes"] / 1e6, len(g)))
    rd, rg = W.roofline_seconds(d, HBM, F32PEAK), W.roofline_seconds(g, HBM, F32PEAK)
    print("      per-layer roofline time: D %.3f ms, G %.3f ms -> %.3e samples/s at 100%%" % (
        rd * 1e3, rg * 1e3, 2 * B * 8192 / (rd + rg)))
if "-v" in sys.argv:
    agg = {}
    for name, c in W.d_step_launches(32) + W.g_step_launches(32):
        a = agg.setdefault(name, [0, 0, 0])
        a[0] += 1; a[1] += c["flops"]; a[2] += c["bytes"]
    for name, (n, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-28s x%-3d %9.2f GFLOP %9.1f MB  %6.1f FLOP/B" % (name, n, fl / 1e9, by / 1e6, fl / max(by, 1)))
