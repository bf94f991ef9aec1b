"""CPU-side checks of the C ABI: the library loads, exports every symbol the header declares,
and its host-only entry points (RegTerm, RegTermDeriv, dense operators) agree with the oracle."""
import os
import re

import numpy as np
import pytest

import filters as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "wavenet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from wavenet_amd import capi
    lib = capi.load_library()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in wavenet_hip.h but not exported"
    assert sorted(capi.SIGNATURES) == syms


def test_status_strings():
    from wavenet_amd import capi
    lib = capi.load_library()
    assert lib.wn_status_string(0) == b"ok"
    assert b"radix" in lib.wn_status_string(capi.WN_ERR_BAD_SHAPE)


@pytest.mark.parametrize("N", [2, 4, 6, 8, 16, 32])
@pytest.mark.parametrize("wavelet", [True, False])
def test_reg_term_matches_oracle_bitwise(oracle, N, wavelet):
    from wavenet_amd import capi
    for seed in range(3):
        a = F.random_filter(N, seed) * 1.3
        assert capi.reg_term(a, wavelet) == oracle.reg_term(a, wavelet)
        assert np.array_equal(capi.reg_term_deriv(a, wavelet), oracle.reg_term_deriv(a, wavelet))


def test_reg_term_rejects_odd_filter():
    from wavenet_amd import capi
    with pytest.raises(capi.WavenetError) as e:
        capi.reg_term(np.ones(3))
    assert e.value.status == capi.WN_ERR_BAD_FILTER


@pytest.mark.parametrize("N", [2, 4, 6, 16, 32])
@pytest.mark.parametrize("size", [0, 1, 3, 5])
def test_dense_operators_match_oracle(oracle, N, size):
    from wavenet_amd import capi
    a = F.random_filter(N, size)
    # taps that fold onto one column are added in a different order: allow one rounding
    assert np.allclose(capi.dense_operator(a, size), oracle.lowpass_operator(a, size), rtol=0, atol=1e-15)
    assert np.allclose(capi.dense_operator(a, size, highpass=True), oracle.highpass_operator(a, size),
                       rtol=0, atol=1e-15)
