// conv_h2c32.hip -- the 32-channel instantiation of the row-pipelined fused BasicBlock kernel (conv_h2c.h), two workgroups per CU
// (conv_h2b.hip's launch_bblock32 hands over to it when the ops carry per-wave weight packs).
#include "conv_h2c.h"

namespace romp {

int launch_bblock32r(const romp_op& op1, const romp_op& op, const float* x, float* y, int B, int* queue, hipStream_t st) {
    return launch_bblockr<32>(op1, op, x, y, B, queue, st);
}

}  // namespace romp
