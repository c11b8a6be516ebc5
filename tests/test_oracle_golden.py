"""CPU: the oracle restatement (oracle/ref_model.py) reproduces the outputs of the reference's OWN model files
(captured by oracle/make_golden.py under the dependency shim) on every golden case."""
import numpy as np
import pytest
import torch

from oracle.cases import CASES
from oracle.ref_model import OracleScoreModel
from oracle.score_norm import ScoreNormTables

from helpers import case_inputs, load_golden, rel_err

TOL = 2e-5  # fp32 restatement vs fp32 reference: only summation-order noise is allowed


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_outputs(name):
    case, gold, batch, sd = case_inputs(name)
    model = OracleScoreModel(case.oracle_config(), sd)
    res = model(batch)
    g = gold["outputs"]
    named = {"confidence": res} if case.confidence_mode else dict(zip(("tr", "rot", "tor", "sc_tor"), res))
    assert set(named) == set(g)
    for key, got in named.items():
        assert got.shape == g[key].shape, key
        assert rel_err(got, g[key]) < TOL, (key, rel_err(got, g[key]))
    assert int(batch["atom", "atom"].edge_index.shape[1]) == gold["edge_counts"]["aa"]


def test_score_norm_lookups_match_reference_probes():
    """The table lookups restated in oracle/score_norm.py reproduce values evaluated by the reference's own
    utils/so3.py:85-89 and utils/torus.py:78-82 (stored beside the tables when they were captured)."""
    t = ScoreNormTables.load()
    p = t.probes
    np.testing.assert_allclose(t.so3_score_norm(p["probe_so3_eps"]), p["probe_so3_val"], rtol=1e-6)
    np.testing.assert_allclose(t.torus_score_norm(p["probe_torus_sigma"]), p["probe_torus_val"], rtol=1e-12)
    assert t.so3_table.shape == (1000,) and t.torus_table.shape == (5001,)


def test_oracle_fp64_close_to_fp32():
    case, gold, batch, sd = case_inputs("cfg1_full")
    m32 = OracleScoreModel(case.oracle_config(), sd)
    m64 = OracleScoreModel(case.oracle_config(), sd, dtype=torch.float64)
    a, b = m32(case.make_batch()), m64(case.make_batch())
    for x, y in zip(a, b):
        assert rel_err(x, y) < 1e-4
