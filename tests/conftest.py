import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle.Oracle()


@pytest.fixture(scope="session")
def ref(request):
    import oracle
    if not oracle.have_ref():
        # a GPU run is expected to carry the prebuilt reference (it travels with the tree): losing it must not pass silently
        if "gpu" in (request.config.getoption("-m") or "") and "not gpu" not in (request.config.getoption("-m") or "") \
                and os.environ.get("GBNNS_ALLOW_NO_REF") != "1":
            pytest.fail("oracle/_ref/libgbnns_ref.so is missing on a GPU run (GBNNS_ALLOW_NO_REF=1 to skip knowingly)")
        pytest.skip("compiled reference (oracle/_ref) not present")
    return oracle.Ref()
