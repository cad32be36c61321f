"""method='LogNormal' for 320 < N <= 640, many basis sizes: the persistent kernel's WIDE form on a cluster of eight against one
workgroup (same bits by construction) and against the host-driven route of round 4 (FRANK_AMD_LN_WIDE=host: the same profile to 1e-5
of the brightest point after six passes -- two Newton schemes on a problem the reference itself reproduces to ~1e-5).   python3 tools/size_sweep_ln_wide.py [N,N,...]"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from frank_amd import FixedGeometry, FrankFitter
    from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
    u, v, V, w = mock_disc_visibilities(200000, seed=31, noise_seed=32)
    out = {}
    for N in [int(x) for x in sys.argv[2].split(",")]:
        FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), alpha=1.3, weights_smooth=1e-2, method="LogNormal", max_iter=6,
                         convergence_failure="ignore", verbose=False, check_qbounds=False)
        sol = FF.fit(u, v, V, w)
        out[str(N)] = np.concatenate([sol.I, sol.power_spectrum])
    np.savez(sys.argv[3], **out)
else:
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [
        321, 322, 335, 336, 337, 351, 352, 353, 368, 383, 384, 385, 400, 415, 416, 431, 448, 449, 463, 480, 495, 511, 512, 513, 528,
        543, 544, 559, 575, 576, 577, 592, 607, 608, 623, 624, 625, 638, 639, 640]
    res = {}
    for tag, env in (("cluster8", {}), ("single", {"FRANK_AMD_LN_CLUSTER": "1"}), ("host", {"FRANK_AMD_LN_WIDE": "host"})):
        path = "/tmp/ln_wide_sweep_%s.npz" % tag
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", ",".join(map(str, sizes)), path], env=dict(os.environ, **env), check=True)
        res[tag] = np.load(path)
    bad, far = [], []
    for N in sizes:
        a, b, h = res["cluster8"][str(N)], res["single"][str(N)], res["host"][str(N)]
        same = bool(np.array_equal(a, b) and np.all(np.isfinite(a)))
        dI = np.abs(a[:N] - h[:N]).max() / np.abs(h[:N]).max()
        if not same:
            bad.append(N)
        if not dI < 1e-5:
            far.append(N)
        print("N=%3d  cluster of 8 == one workgroup: %s;  against the host-driven route: max |dI| / max I = %.1e" % (N, same, dI), flush=True)
    print("sizes checked: %d, cluster != single at: %s, further than 1e-5 from the host route at: %s" % (len(sizes), bad, far))
