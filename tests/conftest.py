import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_sessionfinish(session, exitstatus):
    from tests import margins
    margins.dump(ROOT)


def pytest_terminal_summary(terminalreporter):
    from tests import margins
    if not margins.RECORDS:
        return
    terminalreporter.write_sep("-", "parity margins (measured / tolerance)")
    for r in margins.RECORDS:
        terminalreporter.write_line(f"{r['measured']:.3e} / {r['tolerance']:.1e} ({100 * r['used']:.0f} %)  {r['test']} {r['what']}")
