// Ring-row LIF kernel instantiations with 4 quad(s) (= 16 neurons per lane) per wave (see lif_ring.h).
#include "lif_ring.h"

namespace lsm_lif {
ring_fn_t pick_ring_4(int wpc, bool inreg, bool strided) { return pick_ring<4>(wpc, inreg, strided); }
}  // namespace lsm_lif
