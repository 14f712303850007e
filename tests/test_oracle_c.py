"""oracle/langevin_oracle.c (the C restatement build() compiles) against the reference
golden fixtures and the numpy oracle."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import lanpaint_oracle as orc
from tests import golden_cases as gc
from tests.helpers import assert_close, assert_matches_golden, load_golden, xi_list
from tests.stubs import MODELS

B1_CASES = [n for n, c in sorted(gc.CASES.items())
            if c["shape"][0] == 1 and "audio" not in c and "model_options" not in c and not c.get("zero_noise")]


@pytest.mark.parametrize("name", B1_CASES)
def test_c_oracle_matches_reference_golden(name):
    case = gc.build_case(name)
    g = load_golden(name)
    it = iter(xi_list(g))
    model = MODELS[case["model"]](flow=case["flow"] or case["flux"])
    h = case["hyper"]
    eng = c_oracle.COracleLanPaint(model, h["NSteps"], h["Lambda"], h["Beta"], h["StepSize"],
                                   is_flow=case["flow"] or case["flux"], min_step_frac=h["MinStepFrac"],
                                   randn=lambda like: next(it))
    x = case["x"].copy()
    out = eng(x, case["y"], case["noise"], case["sigma"], case["mask"], case["times"], n_steps=case["n_steps"])
    assert sum(1 for _ in it) == 0
    assert_matches_golden(x, out, g, name, rel=3e-6)


def test_c_nearest_exact_index_matches_numpy():
    lib = c_oracle.load()
    for n_in, n_out in [(8, 2), (16, 4), (124, 37), (864, 30), (480, 54), (3, 1), (1, 9), (100, 40)]:
        want = orc.nearest_exact_src_index(n_out, n_in)
        got = [lib.orc_nearest_exact_index(i, n_in, n_out) for i in range(n_out)]
        assert list(want) == got


def test_c_reshape_mask_plane_matches_numpy():
    rng = np.random.default_rng(2)
    m = (rng.random((31, 20, 12)) > 0.8).astype(np.float32)
    want = orc.reshape_mask(m, (1, 1, 9, 6, 5), video_inpainting=True)[0, 0]
    got = c_oracle.reshape_mask_plane(m, (9, 6, 5), 5)
    assert np.array_equal(want, got)
