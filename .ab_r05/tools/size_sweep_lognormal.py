"""LogNormal MAP solves over many basis sizes: the cluster of workgroups against one workgroup (bit-identical by construction),
and against the pivoted-LU route (FRANK_AMD_LN_PIVOTED: agreement to 1e-6 in s).   python3 tools/size_sweep_lognormal.py"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from frank_amd import FixedGeometry, FrankFitter
    from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
    u, v, V, w = mock_disc_visibilities(60000, seed=31, noise_seed=32)
    out = {}
    for N in [int(x) for x in sys.argv[2].split(",")]:
        FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), alpha=1.3, weights_smooth=1e-2, method="LogNormal", max_iter=12,
                         convergence_failure="ignore", verbose=False, check_qbounds=False)
        sol = FF.fit(u, v, V, w)
        out[str(N)] = np.concatenate([sol.I, sol.power_spectrum])
    np.savez(sys.argv[3], **out)
else:
    sizes = [113, 120, 127, 128, 129, 144, 159, 160, 161, 176, 191, 192, 193, 208, 224, 239, 240, 241, 255, 256, 257, 272, 288, 300, 303, 304, 305, 319, 320]
    res = {}
    for tag, env in (("cluster8", {}), ("single", {"FRANK_AMD_LN_CLUSTER": "1"}), ("cluster3", {"FRANK_AMD_LN_CLUSTER": "3"})):
        path = "/tmp/ln_sweep_%s.npz" % tag
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", ",".join(map(str, sizes)), path], env=dict(os.environ, **env), check=True)
        res[tag] = np.load(path)
    bad = []
    for N in sizes:
        a, b, c = res["cluster8"][str(N)], res["single"][str(N)], res["cluster3"][str(N)]
        same = np.array_equal(a, b) and np.array_equal(a, c) and np.all(np.isfinite(a))
        if not same:
            bad.append(N)
        print("N=%3d  cluster 8 == single == cluster 3: %s  (max rel diff %.1e)" % (N, same, np.abs(a - b).max() / np.abs(b).max()), flush=True)
    print("sizes checked: %d, mismatches: %s" % (len(sizes), bad))
