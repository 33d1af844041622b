// Ring-row LIF kernel instantiations with 1 quad(s) (= 4 neurons per lane) per wave (see lif_ring.h).
#include "lif_ring.h"

namespace lsm_lif {
ring_fn_t pick_ring_1(int wpc, bool inreg, bool strided) { return pick_ring<1>(wpc, inreg, strided); }
ring_fn_t pick_ring_mask_1(int wpc, int inmask) { return pick_ring_mask<1>(wpc, inmask); }
}  // namespace lsm_lif
