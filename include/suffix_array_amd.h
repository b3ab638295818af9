/*
 * suffix_array_amd.h -- C ABI of the MI355X-native suffix-array construction engine.
 *
 * Drop-in boundary for ONE path of hucsmn/suffix_array: `SuffixArray::new(&[u8])`
 * -> `saca()` -> `cdivsufsort::sort_in_place`.  Citations are file:line in the reference
 * crate (v0.5.0).  Plain pointers and sizes only; no C++ or torch types cross this ABI.
 *
 * Output contract (reference src/saca.rs:9-15, src/sa.rs:72-84): the n non-empty suffixes of
 * T in ascending unsigned-lexicographic order (a proper prefix sorts first), as start offsets.
 * All suffixes are distinct, so the array is unique: the result is bit-identical to the
 * crate's divsufsort path on the same bytes.
 *
 * Every entry point is synchronous, thread-safe and re-entrant; the library keeps no pointer
 * to caller memory after returning and never writes past the stated output length.  There is
 * NO CPU fallback: without a usable HIP device the calls return SA_AMD_ENODEVICE.
 */
#ifndef SUFFIX_ARRAY_AMD_H
#define SUFFIX_ARRAY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes; 0 / -1 / -2 keep the meaning of libdivsufsort's return value that the
 * reference's dependency asserts on (call site reference src/saca.rs:14) */
#define SA_AMD_OK          0
#define SA_AMD_EINVAL     (-1)   /* null pointer with n > 0, n < 0, bad device ordinal */
#define SA_AMD_ENOMEM     (-2)   /* host or device allocation failed */
#define SA_AMD_EHIP       (-3)   /* HIP runtime error (launch, copy, sync) */
#define SA_AMD_ENODEVICE  (-4)   /* no HIP device visible */
#define SA_AMD_EINTERNAL  (-5)   /* refinement did not converge (cannot happen for valid input) */
#define SA_AMD_ERANGE     (-6)   /* check_integrity: an entry exceeds n (the reference panics, src/sa.rs:77-78) */

/* reference src/saca.rs:6  `pub const MAX_LENGTH: usize = std::i32::MAX as usize;` */
#define SA_AMD_MAX_LENGTH 2147483647

/* reference src/saca.rs:6 as a callable, for FFI users that cannot read macros */
int32_t sa_amd_max_length(void);

/*
 * Replaces the C engine the reference binds: `int divsufsort(const unsigned char *T, int *SA,
 * int n)` behind `cdivsufsort::sort_in_place(&[u8], &mut [i32])` (reference src/saca.rs:3,14).
 * T: n bytes (host), any byte values.  SA: n entries (host), contents on entry arbitrary
 * (reference src/sa.rs:31 re-uses an old buffer).  n == 0 is a successful no-op.
 */
int32_t sa_amd_divsufsort(const uint8_t *T, int32_t *SA, int32_t n);

/*
 * Replaces `pub fn saca(s: &[u8], sa: &mut [u32])` itself (reference src/saca.rs:9-15):
 * SA has n + 1 entries, SA[0] = n (src/saca.rs:13), SA[1..=n] as above.  The two asserts of
 * src/saca.rs:10-11 become the caller's job (lengths are implied by n).
 */
int32_t sa_amd_saca_u8(const uint8_t *T, uint32_t *SA, int32_t n);

/*
 * Batch of independent texts, text i built on HIP device `device[i]` (NULL: i mod device
 * count), one host thread per device, no inter-device traffic.
 * status[i] receives the per-text code; the return value is the first non-zero status or 0.
 * Each SA[i] has n[i] + 1 entries (layout of sa_amd_saca_u8).
 * The texts of up to SA_AMD_SMALL_MAX (8192) bytes of a device are built together: one launch per chunk
 * of up to 96 MiB of them, one workgroup per text (a caller that indexes thousands of short strings, the
 * reference's own test domain src/tests.rs:13-17, pays ~0.04-1.6 us per text instead of 20-100 us per call).
 */
int32_t sa_amd_saca_batch(const uint8_t *const *T, uint32_t *const *SA, const int32_t *n,
                          const int32_t *device, int32_t count, int32_t *status);

/* ---- device-resident entry points (text already in HBM; used by bench.py) ---- */

typedef struct sa_amd_stats {
    int32_t sigma;            /* distinct byte values in T */
    int32_t bits_per_symbol;  /* packed code width when sigma is a power of two, 0 = base-sigma packing */
    int32_t symbols_per_key;  /* symbols in the initial 64-bit key */
    int32_t rounds;           /* prefix-doubling refinement rounds after the initial sort */
    int32_t sort_passes;      /* 8-bit radix passes executed in total */
    int32_t sparse_mode;      /* 1: few tied suffixes, ranks looked up in the sorted keys instead of a full ISA */
    int64_t sorted_elements;  /* sum over sort passes of elements moved */
    int64_t unresolved_after_initial; /* suffixes still in groups > 1 after the initial sort */
    int32_t text_rounds;      /* of `rounds`: text-keyed rounds (secondary key read from the text, no rank array) */
    int32_t top32_first;      /* 1: the entropy probe chose to sort on the top 32 key bits first and finish the ties locally */
    int64_t locally_sorted;   /* tied suffixes refined by the in-LDS group sort instead of the global radix sort */
    int32_t readbacks;        /* blocking device -> host read-backs of counters during the build (each one drains the stream) */
    int32_t reserved;
} sa_amd_stats;

/* bytes of device scratch sa_amd_saca_device needs for a text of n bytes */
int64_t sa_amd_workspace_bytes(int32_t n);

/*
 * dT: n bytes in device memory; dSA: n + 1 uint32 in device memory (layout of sa_amd_saca_u8);
 * dWork: sa_amd_workspace_bytes(n) bytes of device scratch, 256-byte aligned (else SA_AMD_EINVAL); stream: a
 * hipStream_t (NULL = default stream) on the current device.  Blocks until the array is
 * complete (the refinement loop reads a 4-byte counter back per round).  stats may be NULL.
 * dT may sit at any byte address; when it is not 4-byte aligned the kernels that read the text in words round the address
 * down and may READ (never write) up to three bytes in front of it -- inside the same device allocation, whose start is
 * aligned.
 */
int32_t sa_amd_saca_device(const uint8_t *dT, uint32_t *dSA, int32_t n, void *dWork,
                           int64_t work_bytes, void *stream, sa_amd_stats *stats);

/* The host-pointer entry points take their device memory (text + SA + workspace of one build = one block), streams and
 * pinned staging buffers from a process-wide pool, so repeated calls do not pay hipMalloc / hipFree; the pool retains at
 * most SA_AMD_CACHE_MAX_BYTES of device memory (default 128 GiB of the 288 GB).  This empties the pool now. */
void sa_amd_release_cache(void);

/* wall-clock phases of the calling thread's most recent host-pointer build, milliseconds:
 * [0] acquire device block + stream, [1] text upload, [2] build on the device, [3] suffix array download,
 * [4] release, [5] total, [6] helper threads of the staged download (0 = one plain hipMemcpy), [7] the fraction of the
 * array that was copied to the host before the build was done (early download: large arrays whose last refinement rounds
 * touch few slots start travelling while those rounds run; [3] is then the time behind the build only), [8] bytes of the
 * workspace that lived in pinned host memory (reduced-memory route: the device could not give the whole workspace).  Returns 9. */
int32_t sa_amd_last_host_timing(double *ms, int32_t capacity);

/* statistics of the most recent build issued by the calling thread (any entry point) */
void sa_amd_last_stats(sa_amd_stats *out);

/* number of visible HIP devices (0 when none; never initialises a context by itself) */
int32_t sa_amd_device_count(void);
/* PCI address ("0000:c1:00.0") of HIP device `device` into buf (capacity >= 16): which physical GPU an ordinal is -- bench.py
 * prints it per rank so that an N-GPU run can be seen to have used N distinct GPUs */
int32_t sa_amd_device_pci_bus_id(int32_t device, char *buf, int32_t capacity);

const char *sa_amd_strerror(int32_t code);
const char *sa_amd_version(void);

/* ---- next rows, on either side of the construction path ----
 *
 * Bucket table of `enable_buckets` (reference src/sa.rs:89-119): 256 * 257 + 1 = 65 793 entries in the
 * layout of src/sa.rs:94; bkt[i] = exclusive right edge of bucket i inside the SA (what the reference
 * gets by counting bigrams, src/sa.rs:100-108, and prefix-summing, src/sa.rs:112-116).  Built the same
 * way here -- from the TEXT alone: a bigram histogram kernel (two half tables of 32-bit LDS counters per
 * pair of workgroups) and one scan; only the text is uploaded (n bytes), 257 KiB come back.
 */
#define SA_AMD_BUCKET_TABLE_LEN 65793
/* T: n bytes, bkt: 65 793 entries; host buffers.  SA is not read (the reference's enable_buckets never touches the
 * array either) and may be NULL; the parameter stays for callers of the earlier form */
int32_t sa_amd_bucket_table(const uint8_t *T, int32_t n, const uint32_t *SA, uint32_t *bkt);
/* SuffixArray::new followed by enable_buckets in one device round trip (the text is uploaded once) */
int32_t sa_amd_saca_u8_buckets(const uint8_t *T, uint32_t *SA, int32_t n, uint32_t *bkt);
/* dSA == NULL: from the text alone (bigram counts; dBkt doubles as the scratch); dSA = a valid device-resident suffix
 * array of dT: one binary search per bucket edge instead (what sa_amd_index_buckets uses) */
int32_t sa_amd_bucket_table_device(const uint8_t *dT, const uint32_t *dSA, int32_t n, uint32_t *dBkt, void *stream);

/*
 * `check_integrity` (reference src/sa.rs:72-84), the validation behind `from_parts` and every
 * `load*` (src/sa.rs:57-64, :293-361), in linear time: returns 1 (true), 0 (false; also when
 * sa_len != n + 1, src/sa.rs:73-75), SA_AMD_ERANGE when an entry exceeds n (the reference panics
 * on the slice index there), or another negative status.  dWork: at least 4 * (n + 1) + 256 bytes
 * (random-store inverse permutation); with sa_amd_check_integrity_work_bytes(n) bytes, 256-byte
 * aligned, and a 16-byte aligned dSA the check runs at streaming cost (binned inverse permutation,
 * one random rank line per slot: 25 -> ~10 ms at 256 MiB).
 */
int64_t sa_amd_check_integrity_work_bytes(int32_t n);
int32_t sa_amd_check_integrity(const uint8_t *T, int32_t n, const uint32_t *SA, int64_t sa_len);
int32_t sa_amd_check_integrity_device(const uint8_t *dT, int32_t n, const uint32_t *dSA, void *dWork,
                                      int64_t work_bytes, void *stream);

/*
 * Device-resident index: the text and its suffix array stay in HBM (SA == NULL: the array is built
 * there, i.e. SuffixArray::new without the 4(n+1)-byte download), then serve the bucket table, the
 * integrity check and BATCHED search -- `contains` / `search_all` / `search_lcp` of reference
 * src/sa.rs:164-253 (no-bucket paths) for `count` patterns per call, one wavefront per pattern.
 * Patterns are concatenated in pat_data; pattern q is pat_data[pat_off[q] .. pat_off[q+1]).
 * Outputs (any may be NULL), per pattern:
 *   contains[q]            1 iff the pattern occurs                            (src/sa.rs:164-170)
 *   range_lo/hi[q]         search_all(pat) == &sa[lo..hi]                      (src/sa.rs:173-204)
 *   lcp_start/len[q]       search_lcp(pat) == start..start+len                 (src/sa.rs:207-253)
 */
typedef struct sa_amd_index sa_amd_index;
int32_t sa_amd_index_create(const uint8_t *T, int32_t n, const uint32_t *SA, sa_amd_index **out);
void sa_amd_index_destroy(sa_amd_index *ix);
int32_t sa_amd_index_sa(const sa_amd_index *ix, uint32_t *SA_out);               /* n + 1 entries */
int32_t sa_amd_index_buckets(sa_amd_index *ix, uint32_t *bkt);                   /* 65 793 entries; the index keeps the table and
                                                                                    later searches start from the pattern's bucket
                                                                                    (get_bucket, reference src/sa.rs:123-144) */
int32_t sa_amd_index_check_integrity(const sa_amd_index *ix);                    /* as sa_amd_check_integrity */
int32_t sa_amd_index_search(const sa_amd_index *ix, const uint8_t *pat_data, const int64_t *pat_off, int32_t count,
                            uint8_t *contains, uint32_t *range_lo, uint32_t *range_hi, uint32_t *lcp_start,
                            uint32_t *lcp_len);

/*
 * Packed format of the `pack` feature (reference src/packed_sa.rs, src/sa.rs:255-361): header u32 magic
 * "SA4x" LE (src/packed_sa.rs:6-7), u32 length, u64 data length (bincode little-endian Vec<u8>), then the
 * suffix array bit-packed at ceil(log2(length)) bits in blocks of 128 (BitPacker4x), the last partial
 * block right-trimmed of zero bytes (src/packed_sa.rs:36-46).  PARITY UNPINNED at byte level: the block
 * layout is the external `bitpacking 0.8` crate's, restated from its published description; the
 * reference's own test pins the round trip only (src/tests.rs:61-76).
 * sa_amd_pack: SA has `length` entries, out has sa_amd_pack_bound(length) bytes; *out_len = bytes written.
 * sa_amd_unpack: *length receives the stored length; SA (capacity entries) receives the array.
 */
int64_t sa_amd_pack_bound(int64_t length);
int32_t sa_amd_pack(const uint32_t *SA, int64_t length, uint8_t *out, int64_t capacity, int64_t *out_len);
int32_t sa_amd_unpack(const uint8_t *bytes, int64_t nbytes, uint32_t *SA, int64_t capacity, int64_t *length);

/* ---- per-kernel timing (HIP events on the launch stream), per calling thread ----
 * begin() zeroes and enables the counters for builds issued by this thread; end() disables them and
 * copies up to `capacity` classes out (ms = summed event time, launches, units = elements or bytes
 * processed); returns the number of kernel classes.  Used by bench.py for the roofline line. */
void sa_amd_profile_begin(void);
/* the same for a subset of the kernel classes only (bit i = class i of sa_amd_profile_kernel_name): bench.py times just
 * the dominant kernel inside its timed region -- ~300 event records per build are ~0.7 ms of host time -- and takes the
 * full per-kernel table from one extra build outside it */
void sa_amd_profile_begin_classes(uint64_t class_mask);
int32_t sa_amd_profile_end(double *ms, int64_t *launches, int64_t *units, int32_t capacity);
const char *sa_amd_profile_kernel_name(int32_t index);

#ifdef __cplusplus
}
#endif
#endif /* SUFFIX_ARRAY_AMD_H */
