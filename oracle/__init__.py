"""CPU restatement of the reference's ORB front-end -- TEST INFRASTRUCTURE ONLY.

Only tests, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``visual_slam_amd``) never
does: it fails loudly when its HIP library is missing.

PARITY UNPINNED: the reference holds no golden vectors for this path and cannot
be built or run here; see ``orb_oracle.c`` for what the restatement follows.

``c`` is the ctypes binding of ``liborb_oracle.so`` (plain C, built by
``oracle/Makefile``); ``np_ref`` is an independent NumPy restatement used to
cross-check it.
"""
from . import c_oracle as c  # noqa: F401
