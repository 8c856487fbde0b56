"""Hand-derived known-answer tests against both CPU restatements (no GPU)."""
import pytest

import oracle
from oracle import np_ref

import kat_cases

IMPLS = {"c": oracle.c, "numpy": np_ref}


@pytest.mark.parametrize("impl", list(IMPLS), ids=list(IMPLS))
@pytest.mark.parametrize("case", kat_cases.ALL, ids=[c.__name__ for c in kat_cases.ALL])
def test_kat(case, impl):
    case(IMPLS[impl])
