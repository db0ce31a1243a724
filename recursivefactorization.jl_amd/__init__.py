"""recursivefactorization.jl_amd -- MI355X-native recursive LU behind RecursiveFactorization.jl's ``lu`` / ``lu!`` API.

The product is ``librflu.so`` (hand-written HIP for gfx950, C ABI in ``include/rflu.h``); this package is the thin
host-side mirror of the reference's interface.  No CPU fallback exists: without the built library and a gfx950 device
every call raises.
"""
from ._ffi import Handle, RfluError, default_handle  # noqa: F401
from .lu import (  # noqa: F401
    LU,
    NOPIVOT_NEGATIVE_INFO,
    Adjoint,
    NoPivot,
    NotIPIV,
    RowMaximum,
    SingularException,
    Transpose,
    Val,
    last_path,
    ldiv_,
    lu,
    lu_,
    normalize_pivot,
)

from .butterfly import (  # noqa: F401,E402
    ButterflyWorkspace,
    butterfly_mul_,
    butterfly_solve_,
    butterfly_workspace,
)

from . import linsolve  # noqa: F401,E402  (LinearSolve.jl's RFLUFactorization cache protocol)

__all__ = [
    "linsolve",
    "ButterflyWorkspace", "butterfly_workspace", "butterfly_solve_", "butterfly_mul_",
    "lu", "lu_", "ldiv_", "LU", "NotIPIV", "RowMaximum", "NoPivot", "Val", "Adjoint", "Transpose", "SingularException",
    "normalize_pivot", "last_path", "Handle", "RfluError", "default_handle", "NOPIVOT_NEGATIVE_INFO",
]
