// seg_reduce_bf16.hip -- the bfloat16 instantiations of seg_reduce.hip's kernels (run_segment_op<T> and everything it launches) as their
// own object, so that the storage types compile side by side.  See "Build" at the top of seg_reduce.hip.
#define GEOT_SEG_PART 4
#include "seg_reduce.hip"
