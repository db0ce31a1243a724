"""CPU oracle for the recursive-LU hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package
(see the header of ``rflu_oracle.c``).  The product package never does.
"""
from .oracle import (  # noqa: F401
    build,
    butterfly_materialize_uv,
    butterfly_mul,
    butterfly_mul_level,
    fill_uniform,
    generic_lufact,
    lib,
    lu,
    np_uniform,
    nsplit,
    perm_from_ipiv,
    residual,
    set_threads,
    unpack_lu,
    use_native,
)
