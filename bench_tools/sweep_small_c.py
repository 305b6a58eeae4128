"""commit latency on small setups against the window target of the SRS window tables (KEAKI_MSM_C_SHARED; 0 = the automatic choice)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from keaki_amd import keaki as K
rng = K.Rng(1)
out = []
for size in (129, 1024, 16384, 65536):
    s = K.KZGSetup.setup(rng.fr_rand(), size)
    p = np.stack([rng.fr_rand() for _ in range(min(size, 4096))]) if size <= 4096 else np.tile(np.stack([rng.fr_rand() for _ in range(4096)]), (size // 4096, 1))
    K.commit(s, p); K.commit(s, p)
    t0 = time.perf_counter()
    for _ in range(8): K.commit(s, p)
    out.append("setup %d: commit(full) %.3f ms" % (size, (time.perf_counter() - t0) / 8 * 1e3))
print("c_shared=%s  " % os.environ.get("KEAKI_MSM_C_SHARED", "auto") + " | ".join(out), flush=True)
