"""Capture the reference's so3/torus score-norm lookup tables as a data fixture.

TEST INFRASTRUCTURE (oracle/): run ONCE in the build container, never on the GPU box.

Imports the reference's own `utils/so3.py` and `utils/torus.py` (read-only, by path) and
stores the two 1-D tables the score model reads at run time:
  * so3._exp_score_norms[1000]   (reference utils/so3.py:41-60, lookup :85-89)
  * torus.score_norm_[5001]      (reference utils/torus.py:71-75, lookup :78-82)
torus.score_norm_ is a Monte-Carlo estimate drawn from the *global* numpy RNG at import time
(utils/torus.py:65-75), so it differs ~1 % between processes; we pin it with np.random.seed(0)
set immediately before the import.  Both modules write large .npy caches into the cwd, so the
script chdirs to a scratch directory first.

Usage:  python oracle/make_score_norm_tables.py [out.npz]
"""
import os
import sys
import tempfile

import numpy as np

REF = os.environ.get("DDP_REFERENCE", "/root/reference")


def main():
    out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else
                          os.path.join(os.path.dirname(__file__), "..", "diffdock_pocket_amd", "assets",
                                       "score_norm_tables.npz"))
    scratch = tempfile.mkdtemp(prefix="ddp_tables_")
    os.chdir(scratch)
    sys.path.insert(0, REF)
    from utils import so3  # noqa: E402  (~4 min, pure numpy)
    np.random.seed(0)
    from utils import torus  # noqa: E402  (~4 min, draws 5e7 normals from the global RNG)
    np.savez_compressed(
        out,
        so3_exp_score_norms=np.asarray(so3._exp_score_norms, dtype=np.float64),
        so3_min_eps=np.float64(so3.MIN_EPS), so3_max_eps=np.float64(so3.MAX_EPS),
        so3_n_eps=np.int64(so3.N_EPS),
        torus_score_norm=np.asarray(torus.score_norm_, dtype=np.float64),
        torus_sigma_min=np.float64(torus.SIGMA_MIN), torus_sigma_max=np.float64(torus.SIGMA_MAX),
        torus_sigma_n=np.int64(torus.SIGMA_N),
        # a few reference-evaluated probes so the restated lookups can be pinned
        probe_so3_eps=np.array([0.03, 0.5, 1.55]),
        probe_so3_val=so3.score_norm(__import__("torch").tensor([0.03, 0.5, 1.55], dtype=__import__("torch").float64)).numpy(),
        probe_torus_sigma=np.array([0.0301, 0.1, 1.0, 3.0, 3.14]),
        probe_torus_val=torus.score_norm(np.array([0.0301, 0.1, 1.0, 3.0, 3.14])),
    )
    print("wrote", out)


if __name__ == "__main__":
    main()
