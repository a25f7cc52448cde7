import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from threadpoolctl import threadpool_limits
from oracle import zvx_oracle as O
from zerovox_amd import config as zcfg, weights as zw
h = zcfg.hifigan_config("v1"); hsd = zw.hifigan_state_dict(h, 0)
mel = np.random.default_rng(0).standard_normal((80, 200)).astype(np.float32)
for n in (8, 16, 32, 64, 128):
    with threadpool_limits(limits=n):
        O.hifigan_generator(mel[:, :8], hsd, h)
        t = time.time(); O.hifigan_generator(mel, hsd, h); print(n, "threads:", round(time.time() - t, 2), "s", flush=True)
