/*
 * sa_diag.h -- entry points of libsuffix_array_amd_diag.so ONLY (built with -DSA_AMD_DIAG from the same sources as the
 * product library).  The diagnostic library additionally honours SA_AMD_SORT_VARIANT values that select timing
 * ablations (kernels that skip ranking or stores and therefore produce WRONG orders) and
 * SA_AMD_TIMING_ONLY_INITIAL_SORT (stops after the initial sort).  None of that exists in libsuffix_array_amd.so.
 * Users: tools/ (profiles/*ablation*, phase stamps) and the primitive tests of tests/test_gpu_parity.py.
 */
#ifndef SUFFIX_ARRAY_AMD_DIAG_H
#define SUFFIX_ARRAY_AMD_DIAG_H
#include "../../include/suffix_array_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* cycles per phase of the stamped kernels, summed over tiles and workgroups; reading zeroes the counters */
int32_t sa_amd_debug_phase_cycles(uint64_t *out, int32_t count);
/* k_group_sort: switch its per-phase stamps on / off (entries 8..13 of the same array) */
int32_t sa_amd_debug_group_sort_stamps(int32_t on);
int32_t sa_amd_debug_sort_variant_count(void);
const char *sa_amd_debug_sort_variant_name(int32_t index);
/* stable LSD radix sort of (u64 key, u32 value) pairs on bits [begin_bit, end_bit); host buffers */
int32_t sa_amd_test_sort_pairs(uint64_t *keys, uint32_t *vals, int64_t count, int32_t begin_bit, int32_t end_bit);
/* the sample sort of the 64-bit stage (kernels/sample_sort.hpp; count >= 262 144): keys in order out, vals[i] = the index key i came
 * from; *done = 0: a bucket that is no equality bucket did not fit a workgroup -- the pipeline then takes the LSD sort */
int32_t sa_amd_test_sample_sort64(uint64_t *keys, uint32_t *vals, int64_t count, int32_t key_bits, int32_t *done);
/* the 32-bit-key form of the same sort (first stage of the two-stage initial sort) */
int32_t sa_amd_test_sort_pairs32(uint32_t *keys, uint32_t *vals, int64_t count, int32_t begin_bit, int32_t end_bit);
/* the same order through two global passes over the top 16 (or 18: nine-bit digits) key bits + the in-LDS bucket sort of the
 * rest (kernels/bucket_sort.hpp); *largest = the largest bucket; returns 1 (buffers untouched) when no workgroup shape holds it */
int32_t sa_amd_test_bucket_sort32(uint32_t *keys, uint32_t *vals, int64_t count, int32_t top_bits, uint32_t *largest);
/* initial packed keys of a text (host buffers; keys has n entries); returns bits in *bits, symbols in *k */
int32_t sa_amd_test_build_keys(const uint8_t *T, int32_t n, uint64_t *keys, int32_t *bits, int32_t *k);
/* micro-prototype for DESIGN.md section 2 / VERDICT r1 row (g): one exact level-0 L-type induce sweep of SA-IS executed by a
 * single wavefront with the bucket heads in LDS; SA (n + 1 slots) holds the LMS suffixes at their places and 0xffffffff
 * elsewhere and is completed in place; typeL: bit j = suffix j is L-type; head: 256 bucket starts; counters[0] = induced
 * suffixes, counters[1] = 64-slot blocks that had to be re-read */
int32_t sa_amd_proto_induce_l(const uint8_t *T, const uint8_t *typeL, uint32_t *SA, int32_t n, const uint32_t *head,
                              double *ms, uint64_t *counters);
#ifdef __cplusplus
}
#endif
#endif
