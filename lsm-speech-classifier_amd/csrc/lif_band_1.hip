// Dense-row LIF kernel on band rows + rewired-synapse lists, INMODE 1 (see lif_dense.h).
#include "lif_dense.h"

namespace lsm_lif {
dense_fn_t pick_band_1(int sl, int wpc) { return pick_dense_sl<1, true>(sl, wpc); }
}  // namespace lsm_lif
