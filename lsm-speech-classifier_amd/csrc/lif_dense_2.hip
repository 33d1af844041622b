// Dense-row LIF kernel instantiations with INMODE 2: input-channel masks in registers (see lif_dense.h).
#include "lif_dense.h"

namespace lsm_lif {
dense_fn_t pick_dense_2(int sl, int wpc, bool refm) { return pick_dense_sl<2>(sl, wpc, refm); }
#if LSM_STAMP
LSM_DEFINE_STAMP_READER(read_lif_stamps_d2)
#endif
}  // namespace lsm_lif
