"""Mirror of the reference's featuresynth/util/device.py:3 -- except that this build only runs on
a HIP device (one process per GPU: LOCAL_RANK selects it)."""
import os

import torch

if torch.cuda.is_available():
    device = torch.device("cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")))
else:  # the modules can be constructed / checkpointed on the host, but never run there
    device = torch.device("cpu")
