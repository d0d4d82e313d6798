"""topolow_amd -- MI355X-native relaxation path of topolow's `euclidean_embedding()`.

Public surface mirrors the reference's exports for this path (NAMESPACE:26,28 and the S3
methods NAMESPACE:9,12 of the reference): `euclidean_embedding`, `create_topolow_map`,
`print_topolow`, `summary_topolow`; the native step runs in libtopolow_relax.so (HIP,
gfx950) and there is no CPU fallback.
"""
from .core import (  # noqa: F401
    RMatrix,
    Topolow,
    create_topolow_map,
    euclidean_embedding,
    prepare_layout_call,
    print_topolow,
    summary_topolow,
)
from ._native import NativeError, options, set_seed  # noqa: F401

__all__ = ["euclidean_embedding", "create_topolow_map", "print_topolow", "summary_topolow",
           "Topolow", "RMatrix", "options", "set_seed", "NativeError", "prepare_layout_call"]
