"""CPU oracle vs the hand-derived known-answer scenarios (both resource algebras)."""
import pytest

from oracle import pyoracle
from tests import kat


@pytest.mark.parametrize("algebra", [pyoracle.MASK, pyoracle.LITERAL])
@pytest.mark.parametrize("scn", kat.scenarios(), ids=lambda s: s[0])
def test_kat(scn, algebra):
    name, c, j, cfg, expect = scn
    r = pyoracle.select(c, j, kat.NOW, algebra=algebra, **cfg)
    kat.check(name, c, j, r.placements, expect, costs=r.costs(), timeline=r.timeline)
