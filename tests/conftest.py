import os
import sys

import pytest

# a glibc abort message (heap corruption, ...) goes to stderr instead of /dev/tty: together with --capture=sys (pytest.ini:
# native stderr is not swallowed) a crash inside the HIP runtime or the library leaves its last words in the log
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # an abort / fault inside native code leaves the C-level stack of the raising thread in the log (tests/c_client/abort_bt.c)
    import faulthandler
    if not faulthandler.is_enabled():
        faulthandler.enable()
    from _util import install_abort_bt
    install_abort_bt()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; see oracle/cgv_oracle.cpp)."""
    from oracle import oracle as o
    o.build()
    return o
