// agsf_geom: the output block and launch geometry of the augmented Gaussian-sum scan (agsf_scan.hpp), shared with the build
// from the caller's source (user_model.hip).
#pragma once
#include "scan_common.hpp"

namespace bf {

struct AgsfOut {
  SView w, m, P;
  int* anc;  // [B][T][N0] index of the leaf each carried component was drawn from (NULL = not emitted)
};

#ifndef BF_JIT
// dynamic LDS of a launch: leaf records, carried records, cumulative weights, carried weights, cross-wave scratch, resampling tables
static inline size_t agsf_lds_bytes(int n, int nw, int n0) {
  const int rec = n + n * n, nt = nw == 1 ? 256 : 64 * nw;
  const int carry_records = nw == 1 ? 256 : ((n0 + 3) & ~3);
  return sizeof(float) * ((size_t)nt * rec + (size_t)carry_records * rec + nt + carry_records + 64 + (nw > 1 ? 4 * nt : 0));
}
#endif

}  // namespace bf
