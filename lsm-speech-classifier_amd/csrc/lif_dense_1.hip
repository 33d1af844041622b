// Dense-row LIF kernel instantiations with INREG=1 (see lif_dense.h).
#include "lif_dense.h"

namespace lsm_lif {
dense_fn_t pick_dense_1(int sl, int wpc) { return pick_dense_sl<true>(sl, wpc); }
}  // namespace lsm_lif
