// Ring-row LIF kernel instantiations with 2 quad(s) (= 8 neurons per lane) per wave (see lif_ring.h).
#include "lif_ring.h"

namespace lsm_lif {
ring_fn_t pick_ring_2(int wpc, bool inreg, bool strided) { return pick_ring<2>(wpc, inreg, strided); }
ring_fn_t pick_ring_mask_2(int wpc, int inmask) { return pick_ring_mask<2>(wpc, inmask); }
}  // namespace lsm_lif
