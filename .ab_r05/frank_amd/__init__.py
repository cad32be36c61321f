"""frank_amd: MI355X-native drop-in for the visibility-fitting hot path of discsim/frank.

    from frank_amd import FrankFitter, FixedGeometry            # instead of frank.radial_fitters / frank.geometry
    from frank_amd.geometry import FitGeometryFourierBessel     # geometry fits on the resident table
    from frank_amd import utilities, io, debris_fitters         # as frank.utilities / frank.io / frank.debris_fitters

The arithmetic lives in frank_amd/libfrank_hip.so (hand-written HIP kernels for gfx950 + rocBLAS /
rocSOLVER, C ABI in include/frank_hip.h).  Importing the fitter classes loads that library and fails
if it has not been built; there is no CPU fallback.
"""
__version__ = "0.1.0"

from frank_amd.constants import rad_to_arcsec, deg_to_rad  # noqa: F401


def __getattr__(name):
    # lazy: `import frank_amd.mock` / `frank_amd.constants` must work on a box without the built library
    if name in ("FrankFitter", "FourierBesselFitter", "FrankRadialFit", "FrankGaussianFit", "FrankLogNormalFit"):
        from frank_amd import radial_fitters
        return getattr(radial_fitters, name)
    if name in ("FixedGeometry", "SourceGeometry", "FitGeometryGaussian", "FitGeometryFourierBessel"):
        from frank_amd import geometry
        return getattr(geometry, name)
    if name in ("FrankDebrisFitter", "FourierBesselDebrisFitter"):
        from frank_amd import debris_fitters
        return getattr(debris_fitters, name)
    if name == "DiscreteHankelTransform":
        from frank_amd.hankel import DiscreteHankelTransform
        return DiscreteHankelTransform
    if name in ("VisibilityMapping", "GaussianModel", "LogNormalMAPModel"):
        from frank_amd import statistical_models
        return getattr(statistical_models, name)
    if name == "CriticalFilter":
        from frank_amd.filter import CriticalFilter
        return CriticalFilter
    raise AttributeError(name)
