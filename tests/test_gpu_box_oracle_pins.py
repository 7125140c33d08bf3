"""The oracle's own pins, run AGAIN on the GPU box (`-m gpu`): every parity claim of tests/test_gpu_parity.py is "the HIP path equals
the oracle", so the record of a GPU run should also show that the oracle was pinned THERE — same compiler output, that box's libm —
to the reference's artefacts: the known answers of tests/test_oracle_kat.py (closed forms derived from the cited formulas) and the block
/ region statistics of the reference's two committed renders (tests/test_oracle_png_pins.py against tests/golden/reference_png_stats.json:
committed DATA; nothing here reads /root/reference).  The CPU suite (`-m "not gpu"`) runs the same functions from their own modules.
"""
import pytest

import test_oracle_kat as _kat
import test_oracle_png_pins as _pins

pytestmark = pytest.mark.gpu

sc = _kat.sc  # the fixture of the known-answer tests

_n = 0
for _mod in (_kat, _pins):
    for _name in sorted(vars(_mod)):
        if _name.startswith("test_") and callable(getattr(_mod, _name)):
            globals()[_name] = getattr(_mod, _name)
            _n += 1
assert _n >= 29, _n
