// LIF kernel instantiations with INREG=0, SEGLDS=0 (see lif_kernel.h); one translation unit
// per combination so that the four build in parallel.
#include "lif_kernel.h"

namespace lsm_lif {
lif_fn_t pick_lif_00(int sl, int wpc) { return pick_sl<false, false>(sl, wpc); }
#if LSM_STAMP
LSM_DEFINE_STAMP_READER(read_lif_stamps_00)
#endif
}  // namespace lsm_lif
