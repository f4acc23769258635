// seg_reduce_f16.hip -- the half instantiations of seg_reduce.hip's kernels (run_segment_op<T> and everything it launches) as their
// own object, so that the storage types compile side by side.  See "Build" at the top of seg_reduce.hip.
#define GEOT_SEG_PART 3
#include "seg_reduce.hip"
