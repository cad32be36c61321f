// Host-side DiscreteHankelTransform set-up (hankel.py:55-93): O(N^2), once per fitter.
#pragma once
#include <vector>

struct fh_dht {
    int N = 0;
    int nu = 0;
    double Rmax = 0, Qmax = 0, j_nN = 0;
    std::vector<double> zeros;         // j_{0,1..N+1}
    std::vector<double> r, q;          // collocation points
    std::vector<double> scale_factor;  // 1 / J1(j_k)^2
    std::vector<double> Ykm;           // N*N row-major, hankel.py:84-87
};

// Returns 0 or a negative FH_ERR_* code.
int fh_dht_build(double Rmax_rad, int N, fh_dht *out);
// Y = 0.5 * j_nN * norm * Ykm (hankel.py:197-199), norm = 1/(pi Qmax^2)
void fh_dht_self_coefficients(const fh_dht &d, double *Y);
