"""innr_amd -- MI355X-native (gfx950) batch brute-force k-NN scan behind innr's own API.

One hot path of arclabs561/innr, rebuilt from scratch for CDNA4: batch::VerticalBatch + batch_dot /
batch_l2_squared / batch_cosine / batch_knn_* (src/batch.rs). Host side = thin mirror of the reference's
function surface over a C ABI (include/innr_hip.h); device side = hand-written HIP kernels
(innr_amd/csrc). No CPU fallback: importing works without a GPU, computing does not.
"""
from . import _lib
from ._lib import (GEN_EXAMPLE_LCG, GEN_UNIFORM, KNN_AUTO, KNN_EXACT, KNN_MFMA, KNN_MFMA_BF16, KNN_MFMA_I8, METRIC_COSINE, METRIC_DOT, METRIC_L2SQ, Context, InnrError,
                   InnrPanic, KnnStats, default_context)

__all__ = ["_lib", "Context", "InnrError", "InnrPanic", "KnnStats", "default_context", "KNN_AUTO", "KNN_EXACT", "KNN_MFMA_I8",
           "KNN_MFMA", "KNN_MFMA_BF16", "METRIC_DOT", "METRIC_L2SQ", "METRIC_COSINE", "GEN_EXAMPLE_LCG", "GEN_UNIFORM"]
