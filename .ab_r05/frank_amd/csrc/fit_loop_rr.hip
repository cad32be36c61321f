// The fit loop with the matrix resident in registers: fit_loop.hip compiled for EIGHT waves per workgroup (two per SIMD, 256 vector
// registers each -- 24 tiles of the matrix and the working set of a step), only its instantiation <0, 6> (solve_posterior_rr).
// Reference: the loop of radial_fitters.py:769-785 around GaussianModel._fit (statistical_models.py:742), as fit_loop.hip.
#define FIT_LOOP_THREADS 512
#define FIT_LOOP_RR 1
#include "fit_loop.hip"
