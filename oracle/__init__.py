"""oracle/ -- CPU restatement of the hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / reported baseline.  The
product package never imports it (the import direction is oracle -> product).

PARITY UNPINNED at op level: see the header of ``tpgref.c``.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtpgref.so")


def build(force=False):
    """Compile tpgref.c -> libtpgref.so with gcc (no-op when up to date)."""
    src = os.path.join(_HERE, "tpgref.c")
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= os.path.getmtime(src)):
        return LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libtpgref.so"],
                          stdout=subprocess.DEVNULL)
    return LIB_PATH
