"""CPU oracle for the topolow relaxation path -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this
package, and only as the checker.  See oracle/topolow_oracle.cpp for what is restated
(reference file:line) and for the parity status ("parity unpinned" numerically: the
reference ships no golden vectors and cannot be built here).
"""
from .topolow_oracle import (  # noqa: F401
    OracleError,
    build,
    controller_script,
    edge_error,
    optimize_layout_exact,
    post_metrics,
)
