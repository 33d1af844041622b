// Ring-row LIF kernel instantiations with 3 quad(s) (= 12 neurons per lane) per wave (see lif_ring.h): the strided layouts of
// reservoirs whose quad count is a multiple of 3 (N = 3072: 12 quads = 4 waves x 3).
#include "lif_ring.h"

namespace lsm_lif {
ring_fn_t pick_ring_3(int wpc, bool inreg, bool strided) { return pick_ring<3>(wpc, inreg, strided); }
}  // namespace lsm_lif
