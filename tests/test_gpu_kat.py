"""The hand-derived known-answer cases, run on the GPU library through its C ABI."""
import pytest

import kat_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", kat_cases.ALL, ids=[c.__name__ for c in kat_cases.ALL])
def test_kat_gpu(case, gctx):
    case(gctx)
