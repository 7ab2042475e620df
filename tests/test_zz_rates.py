"""The throughput floors the parity tests noted on their way (conftest.RateFloors): collected last, so that a busy box fails
HERE and nowhere earlier."""
import pytest


@pytest.mark.gpu
@pytest.mark.rate
def test_rate_floors_noted_by_the_parity_tests(rate_floors):
    assert not rate_floors.missed, "; ".join(rate_floors.missed)
