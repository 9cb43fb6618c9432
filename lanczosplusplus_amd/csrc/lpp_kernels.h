// lpp_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the Lanczos inner engine.
//
// Every kernel here is HBM-bandwidth bound (SpMV: ~0.17 flop/byte); there is deliberately no MFMA.
// Conventions: vectors are arrays of doubles padded to an even count so BLAS-1 kernels move
// 16 B per lane (double2); complex values are interleaved (re,im) == one double2.
// Reductions are two-stage (per-block partial -> k_reduce_final) so results are bitwise
// reproducible run to run (no float atomics).
//
// Reference semantics restated (all under /root/reference/src):
//   x += H y                         Engine/InternalProductStored.h:77,121-124
//   a = Re<y|x>; x -= a y; b = |x|;  (y,x) <- (x/b, -b y)     LanczosSolver [PsimagLite], SURVEY 3.1
#pragma once
#include "lpp_common.h"
#include "lpp_spmv_kernels.h"
#include "lpp_layout_kernels.h"
#include "lpp_blas_kernels.h"
