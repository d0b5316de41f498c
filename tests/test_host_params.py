"""CPU: host-side parameter logic (lumenos_amd/params.py) against the oracle and the reference's
known answers."""
import numpy as np
import pytest

from helpers import T_REF
from lumenos_amd import params as lp


@pytest.mark.parametrize("cols,log_n", [(16, 10), (1024, 12), (4096, 14)])
def test_bgv_params_match_oracle(oracle, cols, log_n):
    from oracle.loader import Params
    P = lp.generate_bgv_params_for_ntt(cols, log_n)
    O = Params.for_ntt(oracle, cols, log_n, T_REF)
    assert P.q + P.p == O.moduli and P.psi == O.psi
    assert [m.bit_length() for m in P.q] == [59] + [57] * (len(P.q) - 1) or all(
        abs(m - (1 << b)) < (1 << 30) for m, b in zip(P.q, [58] + [56] * (len(P.q) - 1)))
    for m, r in zip(P.q + P.p, P.psi):
        assert m % (2 << log_n) == 1 and pow(r, 1 << log_n, m) == m - 1


def test_param_errors_mirror_reference():
    """fhe/bfv.go:126-140 error behaviour"""
    with pytest.raises(ValueError, match="nttSize"):
        lp.bgv_param_bits(1, 12, T_REF)
    with pytest.raises(ValueError, match="logN"):
        lp.bgv_param_bits(16, 0, T_REF)
    with pytest.raises(ValueError, match="does not satisfy T = 1"):
        lp.bgv_param_bits(16, 12, 65537 * 3)


def test_field_roots_match_oracle(oracle):
    for n in (16, 2048, 8192):
        assert lp.field_roots_forward(T_REF, n) == [int(x) for x in oracle.field_roots(T_REF, n)]


def test_queries():
    assert lp.calculate_queries(128, 2) == 309
    assert lp.calculate_queries(128, 1) == 0  # 1 - log2(2) <= 0 (ligero.go:67-69)
