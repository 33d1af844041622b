// Dense-row LIF kernel instantiations with INMODE 1 (see lif_dense.h).
#include "lif_dense.h"

namespace lsm_lif {
dense_fn_t pick_dense_1(int sl, int wpc, bool refm) { return pick_dense_sl<1>(sl, wpc, refm); }
}  // namespace lsm_lif
