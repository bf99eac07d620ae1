"""
ibo_amd -- MI355X-native GP-posterior + acquisition hot path of
misterwindupbird/IBO, behind the reference's own Python API.

    from ibo_amd.gaussianprocess import GaussianProcess, PrefGaussianProcess
    from ibo_amd.gaussianprocess.kernel import GaussianKernel_ard, MaternKernel5, ...
    from ibo_amd.acquisition import maximizeEI, maximizePI, maximizeUCB, EI, PI, UCB, sweep
    from ibo_amd.acquisition.gallery import fastUCBGallery
    from ibo_amd.utils.optimize import direct, cdirect
    from ibo_amd.utils.latinhypercube import lhcSample
    from ibo_amd.utils.testfunctions import Hartman6, Shekel5, Synthetic, learnHyper
    from ibo_amd.gaussianprocess.prior import RBFNMeanPrior        # .train(X, Y, bounds, k, seed)

`install_as_ego()` registers the same modules under the reference's package
name so existing `from ego.acquisition import maximizeEI` code runs unchanged.
All numerics go through libibo_hip.so (hand-written HIP for gfx950); importing
this package without the built library raises.
"""
import sys

from . import _lib                                      # noqa: F401  (raises if the .so is missing)
from ._lib import DeviceArray, IBOError, NotPositiveDefinite, device_count, trim      # noqa: F401

__version__ = "0.1.0"


def install_as_ego():
    """alias ibo_amd.* as ego.* (drop-in for the reference's import paths)"""
    import importlib
    names = ["", ".gaussianprocess", ".gaussianprocess.kernel", ".gaussianprocess.prior",
             ".gaussianprocess.trainhyper", ".acquisition", ".acquisition.gallery", ".utils",
             ".utils.optimize", ".utils.latinhypercube", ".utils.testfunctions"]
    for n in names:
        sys.modules["ego" + n] = importlib.import_module("ibo_amd" + n)
