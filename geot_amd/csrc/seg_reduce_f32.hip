// seg_reduce_f32.hip -- the float instantiations of seg_reduce.hip's kernels (run_segment_op<T> and everything it launches) as their
// own object, so that the storage types compile side by side.  See "Build" at the top of seg_reduce.hip.
#define GEOT_SEG_PART 1
#include "seg_reduce.hip"
