"""causalgpslc.jl_amd — MI355X (gfx950) implementation of the CausalGPSLC.jl GP-kernel +
posterior-prediction hot path behind the reference's function names.

The directory name carries a dot, so it is imported through the repo-root shim
``causalgpslc_jl_amd`` (``import causalgpslc_jl_amd as gp``).
"""
from ._lib import GPSLCError, GPSLCLibraryError, PosDefException, load as load_library  # noqa: F401
from .api import (  # noqa: F401
    Context, GPSLCObject, HyperParameters, PREDICTION_COVARIANCE_NOISE,
    rbfKernelLog, rbfKernelLogScalar, logit, expit, processCov, likelihoodDistribution, extractParameters, conditionalITE, ITEDistributions, ITEsamples, conditionalSATE,
    SATEDistributions, SATEsamples, sampleITE, sampleSATE, predictCounterfactualEffects,
    summarizeEstimates, yLogpdf, gpLogpdf, nodesLogpdf, nodesDraw, mvnLogpdf, mvnDraw, predict, doTRange, getN, getNX, getNU, getNumPosteriorSamples,
)
from . import synth  # noqa: F401
from .pack import saveGPSLCObject, loadGPSLCObject, readPackHeader  # noqa: F401
from .inference import (  # noqa: F401
    gpslc, Posterior, prepareData, generateSigmaU, removeAdjacent, getPriorParameters, getHyperParameters,
    toMatrixModel,
)
from .sharded import predict_sharded, predict_sharded_full, predict_sharded_pack, shard_range, ShardedResult  # noqa: F401
