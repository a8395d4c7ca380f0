import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_terminal_summary(terminalreporter):
    """Print what tests.helpers.assert_close / record collected: one line per compared tensor."""
    import json
    from tests import helpers
    if not helpers.PARITY_REPORT:
        return
    terminalreporter.write_sep("-", "parity report (max_err | rel_l2 | outlier share)")
    for r in helpers.PARITY_REPORT:
        if "tol" in r:
            terminalreporter.write_line(f"{r['what']}: tol {r['tol']:g}  max {r['max_err']:.2e}  l2 {r['rel_l2']:.2e} "
                                        f"(<= {r['l2_limit']:.1e})  outliers {r['outliers']:.2e}")
        else:
            terminalreporter.write_line(json.dumps(r))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_report.json"), "w") as fh:
            json.dump(helpers.PARITY_REPORT, fh, indent=1)
