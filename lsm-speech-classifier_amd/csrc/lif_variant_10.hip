// LIF kernel instantiations with INREG=1, SEGLDS=0 (see lif_kernel.h); one translation unit
// per combination so that the four build in parallel.
#include "lif_kernel.h"

namespace lsm_lif {
lif_fn_t pick_lif_10(int sl, int wpc) { return pick_sl<true, false>(sl, wpc); }
#if LSM_STAMP
LSM_DEFINE_STAMP_READER(read_lif_stamps_10)
#endif
}  // namespace lsm_lif
