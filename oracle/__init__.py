"""CPU oracle for the Pioneer-arm step path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; ``pioneer_amd`` never does.  PARITY UNPINNED: see
``pnr_oracle.h``.

Two restatements live here:

* ``COracle``  — ctypes binding of ``libpnr_oracle.so`` (``pnr_oracle.c``), the
  checker used at batch sizes;
* ``numpy_twin`` — an independent pure-NumPy transliteration of the same
  reference lines, used to cross-check the C code at small sizes.
"""
from .binding import COracle, DynOracle, OrcParams, OrcState, build_oracle, oracle_lib_path  # noqa: F401
from . import numpy_twin  # noqa: F401
