"""Run the hand-scheduled D-step and G-step twice from identical state (eager) and report which gradient tensors differ bitwise.
usage: [MSYNTH_STREAMS=0] python tools/scratch/probe_determinism.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
os.environ["MSYNTH_GRAPH"] = "0"
import numpy as np, torch
import featuresynth as fs
from featuresynth import loss as LS
from featuresynth._synthetic import module_param_shapes, synthetic_features, synthetic_samples, synthetic_state_dict
from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = 32
res = []
for rep in range(3):
    g, d = fs.MelGanGenerator(T, 80), fs.MelGanDiscriminator()
    g.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(module_param_shapes(g), seed=7).items()})
    d.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(module_param_shapes(d), seed=7).items()})
    g.cuda(); d.cuda()
    go = fs.FlatAdam(g.parameters(), lr=1e-4, betas=(0.5, 0.9)); do = fs.FlatAdam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
    dt = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss); gt = GeneratorTrainer(g, go, d, do, LS.mel_gan_gen_loss)
    s = torch.from_numpy(synthetic_samples(B, T * 256)).cuda(); f = torch.from_numpy(synthetic_features(B, 80, T)).cuda()
    out = {}
    r1 = dt.train(s, f); torch.cuda.synchronize()
    out["d_loss"] = r1["d_loss"]
    for k, p in d.named_parameters(): out["dg/" + k] = p.grad.detach().cpu().numpy().copy()
    r2 = gt.train(s, f); torch.cuda.synchronize()
    out["g_loss"] = r2["g_loss"]; out["fake"] = r2["fake"].copy()
    for k, p in g.named_parameters(): out["gg/" + k] = p.grad.detach().cpu().numpy().copy()
    res.append(out)
for rep in (1, 2):
    bad = [k for k in res[0] if not np.array_equal(np.asarray(res[0][k]), np.asarray(res[rep][k]))]
    print("run %d vs run 0: %d of %d differ:" % (rep, len(bad), len(res[0])), bad[:12])
    for k in bad[:3]:
        a, b = np.asarray(res[0][k]), np.asarray(res[rep][k])
        if a.ndim == 3:
            idx = np.argwhere(a != b)
            print("   ", k, a.shape, "differing elements", len(idx), "max |d|", float(np.abs(a - b).max()), "rel", float(np.abs(a - b).max() / np.abs(a).max()))
            print("    co:", sorted(set(idx[:, 0].tolist()))[:24], "ci:", sorted(set(idx[:, 1].tolist())), "k:", sorted(set(idx[:, 2].tolist())))
print("d_loss", [r["d_loss"] for r in res], "g_loss", [r["g_loss"] for r in res])
