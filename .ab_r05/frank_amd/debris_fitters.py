"""Fitters for vertically thick, optically thin (debris) discs: frank/debris_fitters.py:26-157.

Both are the radial fitters with the visibility model pinned to 'debris' -- every design-matrix entry carries
exp(-kz^2 H2[k]), H2 from the scale height at the collocation radii (statistical_models.py:494-496), which the binning
kernels apply per (row, column) -- and `scale_height` made a required argument.
"""
from frank_amd.radial_fitters import FourierBesselFitter, FrankFitter

__all__ = ["FourierBesselDebrisFitter", "FrankDebrisFitter"]


class FourierBesselDebrisFitter(FourierBesselFitter):
    """FourierBesselFitter for a disc of scale height H(R) [arcsec]: `scale_height` is a function R -> H taking the
    collocation radii in arcsec (debris_fitters.py:26-65)."""

    def __init__(self, Rmax, N, geometry, scale_height, nu=0, block_data=True, block_size=10 ** 5, verbose=True, **native):
        # **native: the arguments this package adds to the base class (device=, arithmetic=)
        FourierBesselFitter.__init__(self, Rmax, N, geometry, nu=nu, block_data=block_data, assume_optically_thick=False,
                                     scale_height=scale_height, block_size=block_size, verbose=verbose, **native)


class FrankDebrisFitter(FrankFitter):
    """FrankFitter for a disc of scale height H(R) [arcsec] (debris_fitters.py:68-157); every other argument as in
    FrankFitter, with the same defaults."""

    def __init__(self, Rmax, N, geometry, scale_height, nu=0, block_data=True, block_size=10 ** 5, alpha=1.05, p_0=None,
                 weights_smooth=1e-4, tol=1e-3, method='Normal', I_scale=1e5, max_iter=2000, check_qbounds=True,
                 store_iteration_diagnostics=False, verbose=True, convergence_failure='raise', **native):
        FrankFitter.__init__(self, Rmax, N, geometry, nu=nu, block_data=block_data, block_size=block_size, alpha=alpha,
                             p_0=p_0, weights_smooth=weights_smooth, tol=tol, method=method, I_scale=I_scale,
                             max_iter=max_iter, check_qbounds=check_qbounds,
                             store_iteration_diagnostics=store_iteration_diagnostics, assume_optically_thick=False,
                             scale_height=scale_height, verbose=verbose, convergence_failure=convergence_failure, **native)
