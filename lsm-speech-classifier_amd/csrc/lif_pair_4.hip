// Pair-block ring LIF kernel instantiations with 4 block(s) (= 8 neurons per lane) per wave (see lif_pair.h).
#include "lif_pair.h"

namespace lsm_lif {
pair_fn_t pick_pair_4(int wpc, int inmask, bool leakv) { return pick_pair<4>(wpc, inmask, leakv); }
}  // namespace lsm_lif
