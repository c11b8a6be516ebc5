"""CPU restatement of the reference's score-norm lookups (TEST INFRASTRUCTURE, oracle/).

  so3.score_norm(eps)    reference utils/so3.py:85-89  (table _exp_score_norms[1000], utils/so3.py:41-60)
  torus.score_norm(sig)  reference utils/torus.py:78-82 (table score_norm_[5001], utils/torus.py:71-75)

The tables themselves are DATA captured once from the reference's own modules by
oracle/make_score_norm_tables.py (np.random.seed(0) before importing utils.torus, whose table is a
Monte-Carlo estimate); the .npz also holds probe values evaluated by the reference's own lookup
functions, against which this restatement is pinned (tests/test_oracle_golden.py).
"""
import os

import numpy as np

ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffdock_pocket_amd", "assets",
                     "score_norm_tables.npz")


class ScoreNormTables:
    def __init__(self, z):
        self.so3_table = np.asarray(z["so3_exp_score_norms"], np.float64)
        self.so3_min, self.so3_max, self.so3_n = float(z["so3_min_eps"]), float(z["so3_max_eps"]), int(z["so3_n_eps"])
        self.torus_table = np.asarray(z["torus_score_norm"], np.float64)
        self.t_min, self.t_max, self.t_n = float(z["torus_sigma_min"]), float(z["torus_sigma_max"]), int(z["torus_sigma_n"])
        self.probes = {k: np.asarray(z[k]) for k in z.files if k.startswith("probe_")}

    @classmethod
    def load(cls, path=ASSET):
        with np.load(path) as z:
            return cls(z)

    def so3_score_norm(self, eps):
        eps = np.asarray(eps)
        idx = (np.log10(eps) - np.log10(self.so3_min)) / (np.log10(self.so3_max) - np.log10(self.so3_min)) * self.so3_n
        idx = np.clip(np.around(idx).astype(int), a_min=0, a_max=self.so3_n - 1)
        return self.so3_table[idx]

    def torus_score_norm(self, sigma):
        s = np.log(np.asarray(sigma) / np.pi)
        s = (s - np.log(self.t_min)) / (np.log(self.t_max) - np.log(self.t_min)) * self.t_n
        s = np.round(np.clip(s, 0, self.t_n)).astype(int)
        return self.torus_table[s]
