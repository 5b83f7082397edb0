import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pkg():
    return importlib.import_module("codegraph-rust_amd")


def install_abort_bt():
    """Best effort: build tests/c_client/abort_bt.c and hook it into THIS process, so that an abort / fault inside native
    code (HIP runtime, RCCL, the library) leaves the C-level stack of the raising thread in the log. Returns True when
    installed. Test infrastructure only."""
    import ctypes
    import subprocess
    src = os.path.join(ROOT, "tests", "c_client", "abort_bt.c")
    so = os.path.join(ROOT, "tests", "c_client", "libabort_bt.so")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", "-rdynamic", src, "-o", so])
        lib = ctypes.CDLL(so, mode=ctypes.RTLD_GLOBAL)
        lib.abort_bt_install()
        return True
    except Exception:   # no gcc / read-only tree: the suite runs without it
        return False
