"""What does it cost to READ a layer's weights once per B = 1 generator forward (BASELINE config 2)?  DESIGN section 8: four
arithmetic forms of the 512 -> 256 and 256 -> 128 transposed convs take the same 20 / 10 us.  Here the weights of each
upsampling layer are read by a plain coalesced reduction (torch .sum(): every byte once, no arithmetic to speak of)
  (a) right after a full forward -- the cache state the layer's own kernel finds them in --, and
  (b) a second time straight away (as warm as they get),
with HIP events around the single launch (the event pair itself is measured on an empty region and subtracted)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import numpy as np, torch
import featuresynth as fs
from featuresynth._synthetic import module_param_shapes, synthetic_state_dict
g = fs.MelGanGenerator(32, 80)
g.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(module_param_shapes(g), seed=7).items()})
g.cuda()
x = torch.from_numpy(np.random.default_rng(1).standard_normal((1, 80, 32)).astype(np.float32)).cuda()
ws = [(n, p) for n, p in g.named_parameters() if p.dim() == 3 and p.numel() * 4 >= 30000 and "weight" in n]
ws = [(n, p) for n, p in ws if p.shape[2] in (16, 4) or p.shape[2] == 7][:6]
def ev():
    return torch.cuda.Event(enable_timing=True)
def timed(fn):
    e0, e1 = ev(), ev()
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
with torch.no_grad():
    for _ in range(3): g(x)
    empty = sorted(timed(lambda: None) for _ in range(20))[10]
    print("empty event pair: %.1f us" % empty)
    for name, p in ws:
        cold, warm = [], []
        for _ in range(12):
            g(x); torch.cuda.synchronize()
            cold.append(timed(lambda: p.sum()) - empty)
            warm.append(timed(lambda: p.sum()) - empty)
        cold.sort(); warm.sort()
        mb = p.numel() * 4 / 1e6
        print("%-22s %-18s %6.2f MB  after a forward %6.1f us (%.2f TB/s)   again %6.1f us (%.2f TB/s)"
              % (name, tuple(p.shape), mb, cold[6], mb / cold[6], warm[6], mb / warm[6]))
