// conv_h2c.hip -- the 64-channel instantiation of the row-pipelined fused BasicBlock kernel (conv_h2c.h; the 32-channel one is
// conv_h2c32.hip: seven builds of a 270-MFMA hand-scheduled kernel per channel count are the longest compile of the library, the two
// translation units halve the build's critical path).
#include "conv_h2c.h"

namespace romp {

int launch_bblock64(const romp_op& op1, const romp_op& op, const float* x, float* y, int B, int* queue, hipStream_t st) {
    return launch_bblockr<64>(op1, op, x, y, B, queue, st);
}

}  // namespace romp
