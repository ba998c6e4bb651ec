"""Loss trajectories: two gloo ranks on half batches vs one process on the concatenated batch (the comparison
tests/test_gpu_dp.py::test_two_ranks_match_global_batch makes), printed for the kernel switches in the environment."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "music-synthesis_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.multiprocessing as mp
import test_gpu_dp as T

if __name__ == "__main__":
    from featuresynth._synthetic import synthetic_features, synthetic_samples
    tmp = tempfile.mkdtemp()
    mp.spawn(T._worker, args=(2, T._free_port(), tmp), nprocs=2, join=True)
    r0, r1 = np.load(os.path.join(tmp, "rank0.npz")), np.load(os.path.join(tmp, "rank1.npz"))
    samples = np.concatenate([synthetic_samples(2, 4 * 256, rank=r) for r in range(2)])
    feats = np.concatenate([synthetic_features(2, 80, 4, rank=r) for r in range(2)])
    losses, sd, _ = T._run_steps(samples, feats, 6)
    mean = 0.5 * (r0["losses"] + r1["losses"])
    print("switches", {k: v for k, v in os.environ.items() if k.startswith("MSYNTH_")})
    print("dp mean ", ["%.6f" % v for v in mean])
    print("single  ", ["%.6f" % float(v) for v in losses])
    print("abs diff", ["%.2e" % abs(a - float(b)) for a, b in zip(mean, losses)])
    print("max |dW|", max(float(np.abs(r0[k] - v).max()) for k, v in sd.items()))
