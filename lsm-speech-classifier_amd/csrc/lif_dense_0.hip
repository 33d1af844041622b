// Dense-row LIF kernel instantiations with INREG=0 (see lif_dense.h).
#include "lif_dense.h"

namespace lsm_lif {
dense_fn_t pick_dense_0(int sl, int wpc) { return pick_dense_sl<false>(sl, wpc); }
}  // namespace lsm_lif
