// lpp_host.h -- host-only pieces of the engine (no HIP): error plumbing, the symmetric
// tridiagonal eigen-solver used by the convergence test / Ritz reconstruction, the 1-D row
// partition and the local/remote CSR split.  Compiled into liblpp_engine.so; also exercised
// by the CPU test-suite (no GPU needed).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/lpp_engine.h"

namespace lpp {

void set_error(const std::string& msg);
lpp_status fail(lpp_status code, const std::string& msg);

// number of eigenvalues of tridiag(d,e) strictly below x (Sturm count)
int sturm_count(int n, const double* d, const double* e, double x);
// k-th (0-based) eigenvalue by bisection
double tridiag_kth(int n, const double* d, const double* e, int k);
// full decomposition by the implicit symmetric QR step with Wilkinson shift (Givens bulge
// chasing, deflation from the bottom); eigenvalues ascending in w, eigenvectors in the columns
// of z (row-major n x n).  Returns false when it does not converge.
bool tridiag_qr(int n, const double* d, const double* e, double* w, double* z);

} // namespace lpp
