import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd"))
import torch, numpy as np
import featuresynth as fs
from featuresynth import loss as LS
from featuresynth._synthetic import *
from featuresynth.train import DiscriminatorTrainer, GeneratorTrainer
which = sys.argv[1]
g = fs.MelGanGenerator(32, 80).cuda(); d = fs.MelGanDiscriminator().cuda()
go = fs.FlatAdam(g.parameters(), lr=1e-4, betas=(0.5, 0.9)); do = fs.FlatAdam(d.parameters(), lr=1e-4, betas=(0.5, 0.9))
dt = DiscriminatorTrainer(g, go, d, do, LS.mel_gan_disc_loss); gt = GeneratorTrainer(g, go, d, do, LS.mel_gan_gen_loss)
B, T = 2, 4
s = torch.from_numpy(synthetic_samples(B, T*256)).cuda(); f = torch.from_numpy(synthetic_features(B, 80, T)).cuda()
tr = dt if which == "d" else gt
for i in range(4):
    print(which, "call", i, flush=True)
    r = tr.train(s, f)
    print("  ->", {k: (v if not hasattr(v, 'shape') else v.shape) for k, v in r.items()}, flush=True)
print("done", which, flush=True)
