// internal.h -- shared by the translation units of libgeot_hip.so (not part of the ABI)
#ifndef GEOT_INTERNAL_H
#define GEOT_INTERNAL_H
// records `msg` as the calling thread's geot_last_error() and returns `code`
extern "C" int geot_internal_fail(int code, const char *msg);
// experiment knobs of seg_slab.hip, forwarded by geot_set_option
extern "C" int geot_internal_slab_option(const char *name, int value);   // 1 = known name
extern "C" const char *geot_last_kernel(void);
// records the name of the dominant kernel the calling thread's last operator call launched (geot_last_kernel)
extern "C" void geot_internal_note_kernel(const char *name);
// fills `bytes` (a multiple of 4, 4-byte aligned) with the 32-bit `word` by a KERNEL on `stream` (hipStream_t).  Used instead of
// hipMemsetAsync wherever a call may be captured into a graph: a memset node of a graph captured through PyTorch was seen to write a
// repeating 16-byte pattern (another kernel's argument block) instead of its value from the SECOND replay on (round 6; a bare HIP
// program does not show it: tools/kexp6.hip).  A kernel node replays as captured.
extern "C" int geot_internal_fill(void *p, unsigned long bytes, unsigned int word, void *stream);
#endif
