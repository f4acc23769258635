// seg_reduce_f64.hip -- the double instantiations of seg_reduce.hip's kernels (run_segment_op<T> and everything it launches) as their
// own object, so that the storage types compile side by side.  See "Build" at the top of seg_reduce.hip.
#define GEOT_SEG_PART 2
#include "seg_reduce.hip"
