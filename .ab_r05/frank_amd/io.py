"""UVTables and fit results on disk: the file formats either side of the path (frank/io.py:29-218).

A UVTable is five columns -- u, v [lambda], Re V, Im V [Jy], weights [Jy^-2] -- as text (`.txt` / `.dat`, optionally
`.gz` / `.bz2`) or the arrays u, v, V (complex), weights in an `.npz`.  `save_fit` writes what the reference writes
(pickled solution, profile with its statistical uncertainty, visibility fit on the collocation frequencies, fit and
residual UVTables); the model visibilities of the two tables come from one device pass (FrankRadialFit.predict).
"""
import logging
import os
import pickle

import numpy as np

_TEXT, _PACKED = ('.txt', '.dat'), ('.gz', '.bz2')
_COLUMNS = 'u [lambda]\tv [lambda]\tRe(V)  [Jy]\tIm(V) [Jy]\tWeight [Jy^-2]'


def load_uvtable(data_file):
    """(u, v, vis, weights) from a UVTable file (io.py:29-85)."""
    logging.info('  Loading UVTable')
    stem, ext = os.path.splitext(data_file)
    if ext in _PACKED:
        ext = os.path.splitext(stem)[1]
        if ext not in _TEXT:
            raise ValueError("Compressed UV tables (`.gz` or `.bz2`) must be in one of the formats `.txt` or `.dat`.")
    if ext in _TEXT:
        u, v, re, im, weights = np.genfromtxt(data_file).T  # (genfromtxt opens .gz / .bz2 by itself)
        return u, v, re + 1j * im, weights
    if ext == '.npz':
        table = np.load(data_file)
        u, v, vis, weights = (table[key] for key in ('u', 'v', 'V', 'weights'))
        if not np.iscomplexobj(vis):
            raise ValueError("You provided a UVTable with the extension {}. This extension requires the UVTable's variable "
                             "'V' to be complex (of the form Re(V) + Im(V) * 1j).".format(ext))
        return u, v, vis, weights
    raise ValueError("You provided a UVTable with the extension {}. Please provide it as a `.txt`, `.dat`, or `.npz`. Formats "
                     ".txt and .dat may optionally be compressed (`.gz`, `.bz2`).".format(ext))


def save_uvtable(filename, u, v, vis, weights):
    """Write a UVTable as text (five columns with a header line) or `.npz` (io.py:88-124)."""
    ext = os.path.splitext(filename)[1]
    if ext in _TEXT:
        np.savetxt(filename, np.stack([u, v, vis.real, vis.imag, weights], axis=-1), header=_COLUMNS)
    elif ext == '.npz':
        np.savez(filename, u=u, v=v, V=vis, weights=weights,
                 units={'u': 'lambda', 'v': 'lambda', 'V': 'Jy', 'weights': "Jy^-2"})
    else:
        raise ValueError("file extension must be 'npz', 'txt', or 'dat'.")


def load_sol(sol_file):
    """The pickled solution object of a fit (io.py:127-144).  Result objects hold no device handles: they unpickle on a
    machine without a GPU, and re-create their device context on first use where there is one."""
    with open(sol_file, 'rb') as f:
        return pickle.load(f)


def save_fit(u, v, vis, weights, sol, prefix, save_solution=True, save_profile_fit=True, save_vis_fit=True,
             save_uvtables=True, save_iteration_diag=False, iteration_diag=None, format='npz'):
    """Write the results of a fit under `prefix` (io.py:147-218): `_frank_sol.obj`, `_frank_iteration_diagnostics.obj`,
    `_frank_profile_fit.txt` (r, I, statistical uncertainty), `_frank_vis_fit.<format>` (q, Re V on the collocation
    frequencies), `_frank_uv_fit.<format>` and `_frank_uv_resid.<format>` (UVTables of the model and of data - model)."""
    from frank_amd.utilities import get_fit_stat_uncer
    logging.info('  Saving fit results to {}*'.format(prefix))
    if format not in {'txt', 'dat', 'npz'}:
        raise ValueError("'format' must be 'npz', 'txt', or 'dat'.")
    if save_solution:
        with open(prefix + '_frank_sol.obj', 'wb') as f:
            pickle.dump(sol, f)
    if save_iteration_diag:
        with open(prefix + '_frank_iteration_diagnostics.obj', 'wb') as f:
            pickle.dump(iteration_diag, f)
    if save_profile_fit:
        np.savetxt(prefix + '_frank_profile_fit.txt', np.array([sol.r, sol.I, get_fit_stat_uncer(sol)]).T,
                   header='r [arcsec]\tI [Jy/sr]\tI_uncer [Jy/sr]')
    if save_vis_fit:
        np.savetxt(prefix + '_frank_vis_fit.' + format, np.array([sol.q, sol.predict_deprojected(sol.q).real]).T,
                   header='Baseline [lambda]\tProjected Re(V) [Jy]')
    if save_uvtables:
        logging.info('    Saving fit and residual UVTables. N.B.: These will be of comparable size to your input UVTable')
        model = sol.predict(u, v)
        save_uvtable(prefix + '_frank_uv_fit.' + format, u, v, model, weights)
        save_uvtable(prefix + '_frank_uv_resid.' + format, u, v, vis - model, weights)
