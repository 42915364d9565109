// CPU shim for dump1090_amd/csrc/modes_order.h (the host half of the record hand-off).
#include "../../dump1090_amd/csrc/modes_order.h"
extern "C" unsigned long long shim_order_records(const modes_record *slots, unsigned long long nslots, unsigned first_block,
                                                 modes_record *out, int threads) {
    static modes_order_scratch sc;
    return modes_order_records(slots, (size_t)nslots, 0xFFFFFFFFu, first_block, out, sc, threads);
}
