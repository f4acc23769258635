/*
 * geot_hip_dev.h -- measurement hooks and experiment switches of libgeot_hip.so.
 *
 * NOT part of the stable ABI of include/geot_hip.h: nothing a caller of the operators needs, and the option NAMES and
 * their values change from build to build (they exist so that bench.py, tools/ and the tests can time kernels, label a
 * roofline with the kernel the launcher picked, and force code paths).  A maintainer binding the reference to the
 * library binds geot_hip.h only and checks geot_abi_version() against GEOT_ABI_VERSION.
 */
#ifndef GEOT_HIP_DEV_H
#define GEOT_HIP_DEV_H

#include "geot_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement hooks (used by bench.py / tools; not needed by a caller) ---------------
 * With profiling on, every segment-reduction call records hipEvents around its kernels on
 * the call's stream; geot_profile_read waits for them and returns the accumulated device
 * time per kernel class since the last reset. */
void geot_profile_enable(int on);
void geot_profile_reset(void);
/* main = tile kernel, fixup = carry/gap kernel, aux = memsets; *_calls = launches counted */
int geot_profile_read(double *main_ms, double *fixup_ms, double *aux_ms, int64_t *calls);

/* Name of the dominant kernel the calling thread's LAST operator call launched, spelled as rocprofv3 prints it
 * (e.g. "seg_tile_kernel<float, 4, false, 0, false, 3, 3, 16>"): bench.py labels its roofline with what the launcher
 * picked instead of a literal.  Valid until the thread's next call; "" before the first. */
const char *geot_last_kernel(void);

/* What this box can do right now (bench.py reports it next to the roofline: devices of the pool differ by a few
 * per cent): best-of-`iters` bandwidth of a pure non-temporal 16-B-per-lane read of `buf` (device memory, `bytes`
 * long), and the shader clock a busy wave sees (MHz; s_memtime ticks per 100 MHz s_memrealtime tick).
 * Synchronous; allocates a few bytes; not capturable. */
int geot_profile_box(const void *buf, size_t bytes, int iters, double *read_gbps, double *sclk_mhz, void *stream);

/* The same box's RANDOM-ROW rate: uniform-random rows of `row_bytes` (16 * 2^k <= 1024) out of the caller's own table of `rows`
 * rows, 16 row reads in flight per lane, nothing else - with default-policy loads (*row_gbps) and non-temporal ones
 * (*row_gbps_nt).  The yardstick of the per-edge gather kernels on tables far beyond the caches (BASELINE.json configs[4]:
 * 57 GB): bench.py quotes the kernel's own row-read rate as a fraction of it.  mix_row_gbps != NULL: a third form that also WRITES
 * one row (the sum) per `run` rows read into `out` (a scratch buffer of `out_rows` rows of the same width; it is overwritten) -
 * the operator's read / write mix with trivial segment handling: on this part a few per cent of writes among random reads cost
 * more than their bytes (tools/kexp4.hip), and that, not the kernel, is most of the distance to the pure-read rate.  Rates are
 * bytes of ROWS READ per second in all three forms.  ~16 GB of reads per pass; synchronous. */
int geot_profile_box_rows(const void *table, int64_t rows, int64_t row_bytes, int iters, double *row_gbps, double *row_gbps_nt, void *out,
                          int64_t out_rows, int run, double *mix_row_gbps, void *stream);

/* Tuning knobs for experiments: edges per lane-group sub-chunk (0 = auto), forced vector
 * width in elements (0 = auto), non-temporal policy (-1 = auto; 0 = default cache policy, anything else = nt row loads AND nt
 * dst stores on streamed rows - gathered rows always use the default policy; the half-and-half forms of round 1 are no longer
 * instantiated), lanes per row log2 (-1 = auto). */
void geot_tune(int edges_per_group, int vec, int nontemporal, int lpr_log2);
/* Named switches.  Returns GEOT_OK, or GEOT_EINVAL for a name this build does not know (never silently ignored).
 * In every build:
 * "handoff" = 1 | 0: the tile kernel of a sorted geot_index_scatter* call finishes the runs that straddle
 * tiles itself (write-through carry rows + per-tile flags; the second launch then only tidies up) | classic second pass;
 * "unroll" = 0 | 8 | 16 row loads in flight per lane (fp32 index_scatter, 0 = rule);
 * "narrow" = 1 | 0 lane-per-edge kernel for fp32 rows of <= 7 elements; "xcd" = 1 | 0 XCD-contiguous tile
 * ranges in the gather modes; "nt_keys" = 0 | 1 non-temporal key loads; "hub" = -1 | 0 | 1 per-window carry
 * sums for chains of tiles under one key (-1: when nnz / out_rows >= 4096, the few-key regime; 1: always);
 * "slab_blocks" = 1..4 workgroups per CU of the source-blocked kernel's persistent grid (plans built afterwards),
 * "slab_window" = -2 | -1 | n: how many slabs a wave may run ahead of the slowest wave of its XCD (-2 rule, -1 free);
 * "slab_far" = n: a slowest wave more than n steps behind is not waited for (12); "slab_turn" = 1 | 0: the persistent
 * source-blocked grids of this process take turns per device (a launch waits on its stream for the event of the previous one;
 * skipped on a capturing stream);
 * "slab_sddmm_mfma" = 1 | 0 and "slab_spmm_mfma" = 1 | 0: 16-bit SDDMM / SpMM over plans cut into waves of 512- / 256-byte rows on the
 * matrix cores | the row-per-wave kernels;
 * "handoff_tries" = polls of a predecessor's flag before a run is left to the second launch (0: sample once);
 * "lds_floor" = -1 | bytes: dynamic LDS a tile-kernel launch asks for at least - the cap on workgroups per CU
 * (-1: the rule; 33000 -> 4, 41000 -> 3, 54000 -> 2 per CU); "gather_grid" = tiles a gathered fp32 call is cut into
 * at least (4096; 0 off); "sddmm_shift" = -1 | n: lanes per row of the per-edge SDDMM = natural >> n (-1: the rule).
 *
 * In the DEVELOPMENT build only (geot_amd/libgeot_hip_dev.so: the same sources with -DGEOT_DEV_EXPERIMENTS; what tools/ and the
 * A/B tests load - GEOT_HIP_LIB=dev - and what no product process ever does): the switches of variants that were measured and
 * rejected, whose kernels the product does not instantiate -
 * "slab_nt" = 0 | 1: non-temporal loads of the plan's streams; "slab_unroll" = 8 | 16 row loads in flight per lane of the
 * row-per-wave kernel; "slab_tight" = 1 | 0 and "slab_stage" = 1 | 0 | 2: the window / pre-pass rules for per-call weights;
 * "slab_wrow_all" = 0 | 1: every plan of 512 / 256-byte rows cut into waves; "slab_pair" = 0 | 1: multi-head plans of 512-byte rows
 * read two edges' rows per instruction (seg_slab_wpair_kernel; measured slower);
 * "slab_probe" = 0 | 1: TIMING experiment - the gathered table's buffer descriptor gets zero records, every row read of the
 * wave-row kernels is dropped by the range check (results are WRONG by design; what a kernel costs without its gathers). */
int geot_set_option(const char *name, int value);

#ifdef __cplusplus
}
#endif
#endif /* GEOT_HIP_DEV_H */
