// Dense-row LIF kernel instantiations with INMODE 0 (see lif_dense.h).
#include "lif_dense.h"

namespace lsm_lif {
dense_fn_t pick_dense_0(int sl, int wpc, bool refm) { return pick_dense_sl<0>(sl, wpc, refm); }
}  // namespace lsm_lif
