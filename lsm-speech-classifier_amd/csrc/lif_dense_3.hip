// Dense-row LIF kernel instantiations with INMODE 3: input-channel masks in registers, bit positions coloured so that
// one popcount counts a neuron's active inputs (see lif_dense.h).
#include "lif_dense.h"

namespace lsm_lif {
dense_fn_t pick_dense_3(int sl, int wpc, bool refm) { return pick_dense_sl<3>(sl, wpc, refm); }
#if LSM_STAMP
LSM_DEFINE_STAMP_READER(read_lif_stamps_d3)
#endif
}  // namespace lsm_lif
