import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-splatting-own_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle32():
    import numpy as np
    import oracle
    oracle.build()
    return oracle.Oracle(np.float32)


@pytest.fixture(scope="session")
def oracle64():
    import numpy as np
    import oracle
    oracle.build()
    return oracle.Oracle(np.float64)


@pytest.fixture(scope="session")
def oracle32k():
    """The fp32 oracle with the Gaussian's exponent in the HIP kernels' operation order."""
    import numpy as np
    import oracle
    oracle.build()
    return oracle.Oracle(np.float32, kernel_order=True)
