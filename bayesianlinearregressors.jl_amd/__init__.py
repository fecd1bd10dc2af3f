"""MI355X-native posterior / logpdf / marginals / rand path of BayesianLinearRegressors.jl.

Import as ``import blr_amd`` (the directory name follows the reference repository and is not a valid
Python identifier; ``blr_amd.py`` at the repository root loads it under that name).

Exports mirror reference src/BayesianLinearRegressors.jl:11-12.
"""
from . import _abi
from ._abi import BLRError, PosDefException
from .regressor import (
    BasisFunctionRegressor,
    BayesianLinearRegressor,
    BLRFunctionSample,
    ColVecs,
    Diagonal,
    FiniteGP,
    Normal,
    PDMat,
    RandomFourierFeatures,
    ResidentPosterior,
    RowVecs,
    Symmetric,
    cov,
    evaluate,
    logpdf,
    logpdf_and_gradient,
    logpdf_columns,
    logpdf_map,
    marginals,
    mean,
    mean_and_cov,
    mean_and_var,
    posterior,
    posterior_map,
    rand,
    rand_and_pullback,
    rand_b,
    std,
    var,
)

__all__ = [
    "logpdf", "rand", "mean", "std", "cov", "var", "BayesianLinearRegressor", "marginals", "posterior",
    "BasisFunctionRegressor", "ColVecs", "RowVecs", "Diagonal", "Symmetric", "PDMat", "Normal", "FiniteGP",
    "BLRFunctionSample", "RandomFourierFeatures", "mean_and_var", "mean_and_cov", "rand_b", "rand_and_pullback", "evaluate", "logpdf_columns", "logpdf_and_gradient", "logpdf_map", "posterior_map", "BLRError", "PosDefException", "ResidentPosterior",
]
