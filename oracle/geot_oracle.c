/*
 * geot_oracle.c -- CPU restatement of the GeoT segment-reduction hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (geot_amd/) never
 * links, imports or calls anything in oracle/ and fails loudly without its HIP library.
 *
 * Parity status: PINNED for index_scatter (checked bit-for-bit against the reference's
 * own CPU implementation compiled into oracle/_ref, through the operand identity
 * O(index, src) == R(index, src[index]); see tests/test_oracle_vs_ref.py and the
 * captured outputs under tests/golden/).  The gather ops have no CPU implementation in
 * the reference; they follow the reference's sequential checkers and are pinned against
 * the torch comparators the reference's own tests use (tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference tree).  Plain C99, int64 bookkeeping, compiled with -ffp-contract=off so
 * that every product and every sum is rounded separately (no FMA contraction).
 *
 * Build:  gcc -O2 -fopenmp -ffp-contract=off -fPIC -shared geot_oracle.c -o libgeot_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_OK 0
#define ORACLE_EINVAL (-1)
#define ORACLE_ERANGE (-2)
#define ORACLE_ENOMEM (-3)

/* reduce codes: order of csrc/reducetype.h:3 (MAX, MEAN, MIN, SUM, PROD) */
enum { RED_MAX = 0, RED_MEAN = 1, RED_MIN = 2, RED_SUM = 3, RED_PROD = 4 };

int geot_oracle_abi_version(void) { return 1; }

int geot_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Output-row rule shared by every COO op: rows = index[-1] + 1.
 * csrc/index_scatter.cpp:15-20,30-35; csrc/gather_scatter.cpp:27-30;
 * csrc/gather_weight_scatter.cpp:26-29; csrc/mh_spmm.cpp:13-16. */
int64_t geot_oracle_out_rows(const int64_t *index, int64_t nnz) {
  if (nnz <= 0) return ORACLE_EINVAL; /* the reference fails on index[-1] of an empty index */
  return index[nnz - 1] + 1;
}

static int check_keys(const int64_t *index, int64_t nnz, int64_t K) {
  for (int64_t i = 0; i < nnz; i++)
    if (index[i] < 0 || index[i] >= K) return ORACLE_ERANGE;
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------
 * index_scatter, intended semantics dst[index[i]] += src[i]
 * README.md:24-28; csrc/util/check.cuh:78-87 (segment_coo_sequencial);
 * test/test_index_scatter.py:17-23 (scatter_add_ / index_add_ comparators).
 * Accumulation order: strictly sequential in edge order, the order the reference CPU
 * kernel uses inside one segment (csrc/cpu/index_scatter_cpu.cpp:106-113).
 * Works for sorted and unsorted index alike (what the GPU atomics compute, minus ordering).
 * ------------------------------------------------------------------------------------------ */
#define DEF_INDEX_SCATTER(NAME, T, ACC)                                                       \
  int NAME(const int64_t *index, const T *src, T *out, int64_t nnz, int64_t F, int64_t K) {   \
    if (nnz < 0 || F < 0 || K < 0) return ORACLE_EINVAL;                                      \
    int rc = check_keys(index, nnz, K);                                                       \
    if (rc) return rc;                                                                        \
    ACC *acc = (ACC *)calloc((size_t)(K * F > 0 ? K * F : 1), sizeof(ACC));                   \
    if (!acc) return ORACLE_ENOMEM;                                                           \
    for (int64_t i = 0; i < nnz; i++) {                                                       \
      ACC *d = acc + index[i] * F;                                                            \
      const T *s = src + i * F;                                                               \
      for (int64_t j = 0; j < F; j++) d[j] = (ACC)(d[j] + (ACC)s[j]);                         \
    }                                                                                         \
    for (int64_t i = 0; i < K * F; i++) out[i] = (T)acc[i];                                   \
    free(acc);                                                                                \
    return ORACLE_OK;                                                                         \
  }

DEF_INDEX_SCATTER(geot_oracle_index_scatter_f32, float, float)
DEF_INDEX_SCATTER(geot_oracle_index_scatter_f32_acc64, float, double) /* error-bound variant */
DEF_INDEX_SCATTER(geot_oracle_index_scatter_f64, double, double)

/* The reference CPU kernel AS SHIPPED accumulates src[index[n]] instead of src[n]
 * (csrc/cpu/index_scatter_cpu.cpp:110-112, `col = index_data[n]; update(buf, src + col*K)`).
 * This variant restates that quirk so the oracle can be compared with oracle/_ref directly. */
int geot_oracle_index_scatter_refquirk_f32(const int64_t *index, const float *src, float *out,
                                           int64_t nnz, int64_t F, int64_t K) {
  if (nnz < 0 || F < 0 || K < 0) return ORACLE_EINVAL;
  int rc = check_keys(index, nnz, K);
  if (rc) return rc;
  for (int64_t i = 0; i < nnz; i++)
    if (index[i] >= nnz) return ORACLE_ERANGE; /* src[index[n]] must exist */
  memset(out, 0, (size_t)(K * F) * sizeof(float));
  for (int64_t i = 0; i < nnz; i++) {
    float *d = out + index[i] * F;
    const float *s = src + index[i] * F;
    for (int64_t j = 0; j < F; j++) d[j] = d[j] + s[j];
  }
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------
 * The reference CPU algorithm itself: 3-pass segment discovery + per-segment reduce.
 * csrc/cpu/index_scatter_cpu.cpp:25-122:
 *   pass 1 (:38-49)  count key changes per thread, prefix-sum
 *   pass 2 (:51-75)  record (row_index, row_index_offset) per non-empty segment
 *   pass 3 (:89-121) parallel over segments: init -> update per edge -> write (mean divides)
 * Intended operand (src[n]); reductions sum/mean/min/max/prod with ATen's init values
 * (ATen/native/cpu/ReduceUtils.h init_value / update / write); rows without edges stay 0
 * because the shim zero-fills the output first (csrc/index_scatter.cpp:21,35).
 * Used (a) as the CPU baseline that bench.py times, (b) to pin the segment bookkeeping.
 * `sorted` index required (csrc/cpu/index_scatter_cpu.cpp:148-152).
 * ------------------------------------------------------------------------------------------ */
static inline float red_init_f32(int red) {
  switch (red) {
  case RED_PROD: return 1.0f;
  case RED_MAX: return -INFINITY;
  case RED_MIN: return INFINITY;
  default: return 0.0f;
  }
}

/* ATen's _max/_min propagate NaN (ReduceUtils.h:108-135): isnan(y) ? y : std::max(x, y), and
 * std::max(x, y) = (x < y) ? y : x keeps an accumulator that is already NaN. */
static inline float red_max_f32(float x, float y) { return isnan(y) ? y : (x < y ? y : x); }
static inline float red_min_f32(float x, float y) { return isnan(y) ? y : (y < x ? y : x); }

int geot_oracle_index_scatter_3pass_f32(const int64_t *index, const float *src, float *out,
                                        int64_t nnz, int64_t F, int64_t K, int red,
                                        int nthreads) {
  if (nnz <= 0 || F < 0 || K < 0 || red < 0 || red > 4) return ORACLE_EINVAL;
  for (int64_t i = 1; i < nnz; i++)
    if (index[i] < index[i - 1]) return ORACLE_EINVAL; /* sorted only */
  if (index[0] < 0 || index[nnz - 1] >= K) return ORACLE_ERANGE;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  /* pass 1+2: segment table */
  int64_t nseg = 1;
#pragma omp parallel for reduction(+ : nseg) schedule(static)
  for (int64_t i = 1; i < nnz; i++) nseg += (index[i] != index[i - 1]);
  int64_t *row = (int64_t *)malloc((size_t)nseg * sizeof(int64_t));
  int64_t *off = (int64_t *)malloc((size_t)(nseg + 1) * sizeof(int64_t));
  if (!row || !off) { free(row); free(off); return ORACLE_ENOMEM; }
  {
    int64_t m = 0;
    row[0] = index[0];
    off[0] = 0;
    for (int64_t i = 1; i < nnz; i++)
      if (index[i] != index[i - 1]) { m++; row[m] = index[i]; off[m] = i; }
    off[nseg] = nnz;
  }
  memset(out, 0, (size_t)(K * F) * sizeof(float));
  /* pass 3 */
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t m = 0; m < nseg; m++) {
    float *d = out + row[m] * F;
    const float init = red_init_f32(red);
    for (int64_t j = 0; j < F; j++) d[j] = init;
    for (int64_t n = off[m]; n < off[m + 1]; n++) {
      const float *s = src + n * F;
      switch (red) {
      case RED_SUM:
      case RED_MEAN: for (int64_t j = 0; j < F; j++) d[j] = d[j] + s[j]; break;
      case RED_PROD: for (int64_t j = 0; j < F; j++) d[j] = d[j] * s[j]; break;
      case RED_MAX: for (int64_t j = 0; j < F; j++) d[j] = red_max_f32(d[j], s[j]); break;
      case RED_MIN: for (int64_t j = 0; j < F; j++) d[j] = red_min_f32(d[j], s[j]); break;
      }
    }
    if (red == RED_MEAN) {
      const float cnt = (float)(off[m + 1] - off[m]);
      for (int64_t j = 0; j < F; j++) d[j] = d[j] / cnt;
    }
  }
  free(row);
  free(off);
  return ORACLE_OK;
}

/* Segment table alone (int64 bookkeeping): non-empty keys and their first-edge offsets.
 * csrc/cpu/index_scatter_cpu.cpp:38-75.  rows/offs must hold nnz / nnz+1 entries. */
int64_t geot_oracle_segment_table(const int64_t *index, int64_t nnz, int64_t *rows,
                                  int64_t *offs) {
  if (nnz <= 0) return ORACLE_EINVAL;
  int64_t m = 0;
  rows[0] = index[0];
  offs[0] = 0;
  for (int64_t i = 1; i < nnz; i++)
    if (index[i] != index[i - 1]) { m++; rows[m] = index[i]; offs[m] = i; }
  offs[m + 1] = nnz;
  return m + 1;
}

/* ------------------------------------------------------------------------------------------
 * gather_scatter: dst[dst_index[e]] += src[src_index[e]]           (unweighted SpMM)
 * csrc/cuda/gather_scatter_kernel.cuh:118-186; comparator test/test_gather_scatter.py:4-12.
 * gather_weight_scatter: dst[d[e]] += src[s[e]] * w[e]             (weighted SpMM)
 * csrc/util/check.cuh:102-111 (gws_sequencial); csrc/cuda/gather_weight_scatter_kernel.cuh:118-185;
 * comparator test/test_gather_weight_scatter.py:4-11.
 * weight == NULL means unweighted.  Product rounded, then added (no FMA).
 * ------------------------------------------------------------------------------------------ */
#define DEF_GWS(NAME, T, ACC)                                                                 \
  int NAME(const int64_t *src_index, const int64_t *dst_index, const T *weight, const T *src, \
           T *out, int64_t nnz, int64_t F, int64_t src_rows, int64_t K) {                     \
    if (nnz < 0 || F < 0 || K < 0 || src_rows < 0) return ORACLE_EINVAL;                      \
    int rc = check_keys(dst_index, nnz, K);                                                   \
    if (rc) return rc;                                                                        \
    rc = check_keys(src_index, nnz, src_rows);                                                \
    if (rc) return rc;                                                                        \
    ACC *acc = (ACC *)calloc((size_t)(K * F > 0 ? K * F : 1), sizeof(ACC));                   \
    if (!acc) return ORACLE_ENOMEM;                                                           \
    for (int64_t e = 0; e < nnz; e++) {                                                       \
      ACC *d = acc + dst_index[e] * F;                                                        \
      const T *s = src + src_index[e] * F;                                                    \
      if (weight) {                                                                           \
        const T w = weight[e];                                                                \
        for (int64_t j = 0; j < F; j++) {                                                     \
          const T p = (T)(s[j] * w);                                                          \
          d[j] = (ACC)(d[j] + (ACC)p);                                                        \
        }                                                                                     \
      } else {                                                                                \
        for (int64_t j = 0; j < F; j++) d[j] = (ACC)(d[j] + (ACC)s[j]);                       \
      }                                                                                       \
    }                                                                                         \
    for (int64_t i = 0; i < K * F; i++) out[i] = (T)acc[i];                                   \
    free(acc);                                                                                \
    return ORACLE_OK;                                                                         \
  }

DEF_GWS(geot_oracle_gather_weight_scatter_f32, float, float)
DEF_GWS(geot_oracle_gather_weight_scatter_f32_acc64, float, double)
DEF_GWS(geot_oracle_gather_weight_scatter_f64, double, double)

/* ------------------------------------------------------------------------------------------
 * mh_spmm: dst[d[e], h, :] += w[e, h] * src[s[e], h, :]     weight [nnz, H]
 *          (transposed != 0: weight is [H, nnz], w = weight[h * nnz + e])
 * csrc/cuda/mh_spmm_kernel.cuh:28-111 and :130-213; layout choice csrc/cuda/wrapper/mh_spmm_base.h:38-49;
 * comparator test/test_mh_spmm.py:4-10.
 * ------------------------------------------------------------------------------------------ */
#define DEF_MH(NAME, T, ACC)                                                                  \
  int NAME(const int64_t *src_index, const int64_t *dst_index, const T *weight, const T *src, \
           T *out, int64_t nnz, int64_t H, int64_t F, int64_t src_rows, int64_t K,            \
           int transposed) {                                                                  \
    if (nnz < 0 || F < 0 || H < 0 || K < 0 || src_rows < 0) return ORACLE_EINVAL;             \
    int rc = check_keys(dst_index, nnz, K);                                                   \
    if (rc) return rc;                                                                        \
    rc = check_keys(src_index, nnz, src_rows);                                                \
    if (rc) return rc;                                                                        \
    const int64_t N = H * F;                                                                  \
    ACC *acc = (ACC *)calloc((size_t)(K * N > 0 ? K * N : 1), sizeof(ACC));                   \
    if (!acc) return ORACLE_ENOMEM;                                                           \
    for (int64_t e = 0; e < nnz; e++) {                                                       \
      ACC *d = acc + dst_index[e] * N;                                                        \
      const T *s = src + src_index[e] * N;                                                    \
      for (int64_t h = 0; h < H; h++) {                                                       \
        const T w = transposed ? weight[h * nnz + e] : weight[e * H + h];                     \
        for (int64_t j = 0; j < F; j++) {                                                     \
          const T p = (T)(s[h * F + j] * w);                                                  \
          d[h * F + j] = (ACC)(d[h * F + j] + (ACC)p);                                        \
        }                                                                                     \
      }                                                                                       \
    }                                                                                         \
    for (int64_t i = 0; i < K * N; i++) out[i] = (T)acc[i];                                   \
    free(acc);                                                                                \
    return ORACLE_OK;                                                                         \
  }

DEF_MH(geot_oracle_mh_spmm_f32, float, float)
DEF_MH(geot_oracle_mh_spmm_f32_acc64, float, double)
DEF_MH(geot_oracle_mh_spmm_f64, double, double)

/* ------------------------------------------------------------------------------------------
 * sddmm_coo: out[e] = < mat_1[dst_index[e], :], mat_2[src_index[e], :] >
 * csrc/cuda/gather_weight_scatter_cuda.cu:41-62 (row_indices = dst_index, col_indices = src_index,
 * X1 = mat_1, X2 = mat_2); kernels csrc/cuda/sddmm_coo_kernel.cuh:3-210.
 * Sequential left-to-right dot product; acc64 variant bounds the error.
 * ------------------------------------------------------------------------------------------ */
#define DEF_SDDMM(NAME, T, ACC)                                                               \
  int NAME(const int64_t *src_index, const int64_t *dst_index, const T *mat1, const T *mat2,  \
           T *out, int64_t nnz, int64_t F, int64_t rows1, int64_t rows2) {                    \
    if (nnz < 0 || F < 0) return ORACLE_EINVAL;                                               \
    int rc = check_keys(dst_index, nnz, rows1);                                               \
    if (rc) return rc;                                                                        \
    rc = check_keys(src_index, nnz, rows2);                                                   \
    if (rc) return rc;                                                                        \
    for (int64_t e = 0; e < nnz; e++) {                                                       \
      const T *a = mat1 + dst_index[e] * F;                                                   \
      const T *b = mat2 + src_index[e] * F;                                                   \
      ACC s = 0;                                                                              \
      for (int64_t j = 0; j < F; j++) {                                                       \
        const ACC p = (ACC)a[j] * (ACC)b[j];                                                  \
        s = s + p;                                                                            \
      }                                                                                       \
      out[e] = (T)s;                                                                          \
    }                                                                                         \
    return ORACLE_OK;                                                                         \
  }

DEF_SDDMM(geot_oracle_sddmm_coo_f32, float, float)
DEF_SDDMM(geot_oracle_sddmm_coo_f32_acc64, float, double)

/* ------------------------------------------------------------------------------------------
 * gather rows (backward of index_scatter): dst[e] = src[index[e]]
 * csrc/util/check.cuh:90-99 (gather_sequencial); kernel csrc/cuda/index_scatter_kernel.cuh:266-315.
 * ------------------------------------------------------------------------------------------ */
int geot_oracle_gather_rows_f32(const int64_t *index, const float *src, float *out, int64_t nnz,
                                int64_t F, int64_t src_rows) {
  int rc = check_keys(index, nnz, src_rows);
  if (rc) return rc;
  for (int64_t e = 0; e < nnz; e++) memcpy(out + e * F, src + index[e] * F, (size_t)F * sizeof(float));
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------------------------
 * csr_gws: out[r] = sum_{e in [indptr[r], indptr[r+1])} src[indices[e]] * weight[e]
 * csrc/csr_gws.cpp:24-35 (output has indptr.size(0) = nrow+1 rows, the extra row stays 0),
 * kernel csrc/cuda/csr_gws_kernel.cuh:12-186; comparator test/test_csr_gws.py:16-25.
 * weight == NULL -> ones.  Sequential in nonzero order; acc64 variant for the error bound.
 * ------------------------------------------------------------------------------------------ */
#define DEF_CSR(NAME, ACC)                                                                     \
  int NAME(const int64_t *indptr, const int64_t *indices, const float *weight, const float *src, \
           float *out, int64_t nrow, int64_t F, int64_t src_rows, int64_t out_rows) {         \
    if (nrow < 0 || F < 0 || out_rows < nrow) return ORACLE_EINVAL;                           \
    for (int64_t r = 0; r < nrow; r++)                                                        \
      if (indptr[r] > indptr[r + 1] || indptr[r] < 0) return ORACLE_EINVAL;                   \
    for (int64_t i = 0; i < out_rows * F; i++) out[i] = 0.0f;                                 \
    for (int64_t r = 0; r < nrow; r++) {                                                      \
      for (int64_t j = 0; j < F; j++) {                                                       \
        ACC s = 0;                                                                            \
        for (int64_t e = indptr[r]; e < indptr[r + 1]; e++) {                                 \
          if (indices[e] < 0 || indices[e] >= src_rows) return ORACLE_ERANGE;                 \
          const float p = weight ? src[indices[e] * F + j] * weight[e] : src[indices[e] * F + j]; \
          s = (ACC)(s + (ACC)p);                                                              \
        }                                                                                     \
        out[r * F + j] = (float)s;                                                            \
      }                                                                                       \
    }                                                                                         \
    return ORACLE_OK;                                                                         \
  }

DEF_CSR(geot_oracle_csr_gws_f32, float)
DEF_CSR(geot_oracle_csr_gws_f32_acc64, double)

/* coo_to_csr: int32 row pointers of length nrow+1 = [0, cumsum(histogram(coo_row))]
 * geot/match_replace/format_transform.py:5-18 (hist by geot/triton/coo_to_csr.py:14-26). */
int geot_oracle_coo_to_csr(const int64_t *coo_row, int64_t nnz, int64_t nrow, int32_t *rowptr) {
  if (nnz < 0 || nrow < 0) return ORACLE_EINVAL;
  for (int64_t r = 0; r <= nrow; r++) rowptr[r] = 0;
  for (int64_t e = 0; e < nnz; e++) {
    if (coo_row[e] < 0 || coo_row[e] >= nrow) return ORACLE_ERANGE;
    rowptr[coo_row[e] + 1]++;
  }
  for (int64_t r = 0; r < nrow; r++) rowptr[r + 1] += rowptr[r];
  return ORACLE_OK;
}
