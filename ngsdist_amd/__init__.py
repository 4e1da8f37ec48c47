"""ngsdist_amd -- MI355X-native engine for the gen_dist() hot path of ngsDist.

Only what the path needs lives here: csrc/ (HIP kernels, the C ABI, the C++
host driver) and this thin ctypes door.  Importing the package does not load
the library; the first Engine()/finish()/Taus() does, and fails loudly if the
HIP engine was not built.
"""
from .engine import (DEFAULT_SCORE, KERNELS, Engine, NgdError, Taus, device_count, finish, format_matrix, n_pairs,
                     score_congruence, score_matrix)

__all__ = ["Engine", "NgdError", "Taus", "finish", "format_matrix", "device_count", "n_pairs", "score_matrix",
           "score_congruence", "DEFAULT_SCORE", "KERNELS"]
