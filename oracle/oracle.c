/*
 * oracle.c -- CPU oracle for the suffix-array construction path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.
 *
 * What it restates
 * ----------------
 * The reference's hot path is `saca()` (reference src/saca.rs:9-15), which
 * forwards to the third-party crate `cdivsufsort = "2.0"` (reference
 * Cargo.toml:17; Cargo.lock is git-ignored, reference .gitignore:3, so the
 * exact version is unpinned) wrapping Yuta Mori's libdivsufsort.  That C
 * source is NOT under /root/reference and no copy exists on this image, so it
 * cannot be compiled or imported here.  The output contract, however, is fully
 * pinned by the reference itself:
 *
 *   - buffer contract: sa.len() == n + 1, sa[0] = n, sa[1..] = the sorted
 *     suffix start offsets (reference src/saca.rs:10-14);
 *   - order: strict `<` on Rust `[u8]` slices, i.e. unsigned byte-wise
 *     lexicographic with a proper prefix first (reference src/sa.rs:72-84,
 *     `check_integrity`, enforced by the property test `conversion_correctness`
 *     reference src/tests.rs:13-17);
 *   - all n+1 suffixes (incl. the empty one) are distinct, so exactly one
 *     array satisfies the check: "bit-exact vs divsufsort" == "bit-exact vs
 *     any correct suffix sorter".
 *
 * The oracle therefore restates the CONTRACT with three independent pieces:
 *   oracle_naive_sa        comparison sort of suffixes (obviously correct)
 *   oracle_sais            linear-time induced sorting (published SA-IS
 *                          algorithm, Nong/Zhang/Chan 2009, written from the
 *                          paper; libdivsufsort's own published scheme is the
 *                          related two-stage induced sort: sort the B* suffixes,
 *                          induce B right-to-left, induce A left-to-right)
 *   oracle_check_integrity literal restatement of reference src/sa.rs:72-84
 *   oracle_verify_sa       linear-time equivalent of the same check
 *   oracle_verify_sa_mt    the same on several host threads (full-size configs)
 * Pinning: tests/test_oracle.py checks all of them against the known answers
 * of SURVEY.md section 8a, the reference's only literal vector
 * (`search_all(b"splend") == [0, 9]` on b"splendid splendor", reference
 * src/lib.rs:28-29) and against each other on the reference's own test domain
 * (random bytes, n in [0, 4096), reference src/tests.rs:6-17).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* naive: comparison sort, follows the ordering of reference           */
/* src/sa.rs:76-82 (slice `<` = memcmp on the common prefix, then the  */
/* shorter slice is smaller).                                          */
/* ------------------------------------------------------------------ */
static const uint8_t *g_text;
static int64_t g_n;

static int suffix_cmp(const void *pa, const void *pb)
{
    uint32_t a = *(const uint32_t *)pa, b = *(const uint32_t *)pb;
    int64_t la = g_n - a, lb = g_n - b;
    int64_t l = la < lb ? la : lb;
    int c = l ? memcmp(g_text + a, g_text + b, (size_t)l) : 0;
    if (c) return c;
    return (la > lb) - (la < lb);
}

/* sa has n+1 entries; sa[0] = n (reference src/saca.rs:13). NOT re-entrant. */
ORACLE_API int32_t oracle_naive_sa(const uint8_t *s, uint32_t *sa, int64_t n)
{
    if (n < 0 || (!s && n > 0) || !sa) return -1;
    for (int64_t i = 0; i <= n; ++i) sa[i] = (uint32_t)i;
    g_text = s;
    g_n = n;
    qsort(sa, (size_t)n + 1, sizeof(uint32_t), suffix_cmp);
    return 0;
}

/* ------------------------------------------------------------------ */
/* literal restatement of check_integrity, reference src/sa.rs:72-84   */
/* returns 1 = true, 0 = false, -1 = would panic (index out of range)  */
/* ------------------------------------------------------------------ */
ORACLE_API int32_t oracle_check_integrity(const uint8_t *s, int64_t n, const uint32_t *sa, int64_t sa_len)
{
    if (n + 1 != sa_len) return 0;                      /* src/sa.rs:73-75 */
    for (int64_t i = 1; i < sa_len; ++i) {              /* src/sa.rs:76 */
        if ((int64_t)sa[i - 1] > n || (int64_t)sa[i] > n) return -1; /* slice index panics */
        const uint8_t *x = s + sa[i - 1];               /* src/sa.rs:77 */
        const uint8_t *y = s + sa[i];                   /* src/sa.rs:78 */
        int64_t lx = n - sa[i - 1], ly = n - sa[i];
        int64_t l = lx < ly ? lx : ly;
        int c = l ? memcmp(x, y, (size_t)l) : 0;
        if (c == 0) c = (lx > ly) - (lx < ly);
        if (c >= 0) return 0;                           /* x >= y -> false, src/sa.rs:79-81 */
    }
    return 1;
}

/* ------------------------------------------------------------------ */
/* linear-time equivalent of the same check:        */
/* sa[0]==n, sa[1..] a permutation of 0..n, and for each adjacent pair */
/* T[a]<T[b] or (T[a]==T[b] and rank[a+1]<rank[b+1]).                  */
/* returns 1 ok, 0 not a suffix array, -2 out of memory                */
/* ------------------------------------------------------------------ */
ORACLE_API int32_t oracle_verify_sa(const uint8_t *s, int64_t n, const uint32_t *sa, int64_t sa_len)
{
    if (n + 1 != sa_len) return 0;
    if (sa[0] != (uint32_t)n) return 0;
    uint32_t *rank = (uint32_t *)malloc(((size_t)n + 1) * sizeof(uint32_t));
    if (!rank) return -2;
    memset(rank, 0xff, ((size_t)n + 1) * sizeof(uint32_t));
    int ok = 1;
    for (int64_t i = 0; i <= n && ok; ++i) {
        if ((int64_t)sa[i] > n || rank[sa[i]] != 0xffffffffu) ok = 0;
        else rank[sa[i]] = (uint32_t)i;
    }
    for (int64_t i = 2; i <= n && ok; ++i) {
        uint32_t a = sa[i - 1], b = sa[i];   /* both < n here because sa[0]==n and sa is a permutation */
        if (s[a] < s[b]) continue;
        if (s[a] > s[b]) { ok = 0; break; }
        if (rank[a + 1] >= rank[b + 1]) ok = 0;
    }
    free(rank);
    return ok;
}

/* ------------------------------------------------------------------ */
/* LCP statistics of a text from its suffix array (Kasai et al.): the   */
/* corpus facts bench.py states for its workload (mean / max LCP, how   */
/* many suffixes share at least 2^k symbols with their SA predecessor). */
/* out: [0] sum of LCP, [1] max LCP, [2 + k] count of LCP >= 2^k, k<30  */
/* ------------------------------------------------------------------ */
ORACLE_API int32_t oracle_lcp_stats(const uint8_t *s, int64_t n, const uint32_t *sa, uint64_t *out)
{
    for (int k = 0; k < 32; ++k) out[k] = 0;
    if (n <= 0) return 0;
    uint32_t *rank = (uint32_t *)malloc(((size_t)n + 1) * sizeof(uint32_t));
    if (!rank) return -2;
    for (int64_t i = 0; i <= n; ++i) rank[sa[i]] = (uint32_t)i;
    int64_t h = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = rank[i];          /* >= 1: slot 0 is the empty suffix */
        const int64_t j = sa[r - 1];        /* predecessor (j == n: the empty suffix, LCP 0) */
        while (i + h < n && j + h < n && s[i + h] == s[j + h]) ++h;
        out[0] += (uint64_t)h;
        if ((uint64_t)h > out[1]) out[1] = (uint64_t)h;
        for (int k = 0; k < 30 && ((int64_t)1 << k) <= h; ++k) out[2 + k]++;
        if (h > 0) --h;
    }
    free(rank);
    return 0;
}

/* the LCP array itself (lcp[i] = LCP of the suffixes at SA slots i-1 and i, lcp[0] = lcp[1] = 0): tools/ only */
ORACLE_API int32_t oracle_lcp_array(const uint8_t *s, int64_t n, const uint32_t *sa, uint32_t *lcp)
{
    if (n < 0) return -1;
    uint32_t *rank = (uint32_t *)malloc(((size_t)n + 1) * sizeof(uint32_t));
    if (!rank) return -2;
    for (int64_t i = 0; i <= n; ++i) rank[sa[i]] = (uint32_t)i;
    lcp[0] = 0;
    int64_t h = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = rank[i], j = sa[r - 1];
        while (i + h < n && j + h < n && s[i + h] == s[j + h]) ++h;
        lcp[r] = (uint32_t)h;
        if (h > 0) --h;
    }
    free(rank);
    return 0;
}

/* ------------------------------------------------------------------ */
/* the same linear-time check on `threads` host threads, for the        */
/* full-size configs (1 GiB: four random accesses per entry).  The      */
/* sequential duplicate test is replaced by a read-back pass: after the */
/* scatter rank[sa[i]] == i must hold for every i, which fails for one  */
/* of any two entries that name the same suffix.                        */
/* ------------------------------------------------------------------ */
#include <pthread.h>

typedef struct {
    const uint8_t *s; const uint32_t *sa; uint32_t *rank; int64_t n, lo, hi; int phase; int ok;
} verify_job;

static void *verify_worker(void *arg)
{
    verify_job *j = (verify_job *)arg;
    const uint8_t *s = j->s; const uint32_t *sa = j->sa; uint32_t *rank = j->rank;
    const int64_t n = j->n;
    int ok = 1;
    if (j->phase == 0) {
        for (int64_t i = j->lo; i < j->hi; ++i) {
            if ((int64_t)sa[i] > n) { ok = 0; break; }
            rank[sa[i]] = (uint32_t)i;
        }
    } else if (j->phase == 1) {
        for (int64_t i = j->lo; i < j->hi; ++i) if (rank[sa[i]] != (uint32_t)i) { ok = 0; break; }
    } else {
        for (int64_t i = j->lo < 2 ? 2 : j->lo; i < j->hi; ++i) {
            const uint32_t a = sa[i - 1], b = sa[i];
            if (s[a] < s[b]) continue;
            if (s[a] > s[b] || rank[a + 1] >= rank[b + 1]) { ok = 0; break; }
        }
    }
    j->ok = ok;
    return NULL;
}

ORACLE_API int32_t oracle_verify_sa_mt(const uint8_t *s, int64_t n, const uint32_t *sa, int64_t sa_len, int32_t threads)
{
    if (n + 1 != sa_len) return 0;
    if (sa[0] != (uint32_t)n) return 0;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    uint32_t *rank = (uint32_t *)malloc(((size_t)n + 1) * sizeof(uint32_t));
    if (!rank) return -2;
    int ok = 1;
    for (int phase = 0; phase < 3 && ok; ++phase) {
        pthread_t th[64];
        verify_job jobs[64];
        const int64_t per = (n + 1 + threads - 1) / threads;
        int started = 0;
        for (int t = 0; t < threads; ++t) {
            verify_job *j = &jobs[t];
            j->s = s; j->sa = sa; j->rank = rank; j->n = n; j->phase = phase; j->ok = 1;
            j->lo = (int64_t)t * per; j->hi = j->lo + per > n + 1 ? n + 1 : j->lo + per;
            if (j->lo >= j->hi) break;
            if (pthread_create(&th[t], NULL, verify_worker, j) != 0) { verify_worker(j); th[t] = 0; }
            started = t + 1;
        }
        for (int t = 0; t < started; ++t) { if (th[t]) pthread_join(th[t], NULL); if (!jobs[t].ok) ok = 0; }
    }
    free(rank);
    return ok;
}

/* ------------------------------------------------------------------ */
/* SA-IS, written from the published description.  Generic over the    */
/* symbol type via a macro: u8 at level 0, int32 names in recursion.   */
/* The sentinel is virtual (position n, smaller than every symbol) so  */
/* all 256 byte values are legal input (reference src/tests.rs:8).     */
/* ------------------------------------------------------------------ */
#define EMPTY (-1)

static void sais_i32(const int32_t *s, int32_t *SA, int32_t n, int32_t K);

#define DEFINE_SAIS(NAME, CHAR_T)                                                              \
static void NAME(const CHAR_T *s, int32_t *SA, int32_t n, int32_t K)                           \
{                                                                                              \
    if (n == 0) return;                                                                        \
    if (n == 1) { SA[0] = 0; return; }                                                         \
    uint8_t *t = (uint8_t *)malloc((size_t)n + 1);        /* 1 = S-type, 0 = L-type */         \
    int32_t *bkt = (int32_t *)malloc(((size_t)K + 1) * sizeof(int32_t));                       \
    int32_t *ptr = (int32_t *)malloc(((size_t)K + 1) * sizeof(int32_t));                       \
    t[n] = 1; t[n - 1] = 0;                                                                    \
    for (int32_t i = n - 2; i >= 0; --i)                                                       \
        t[i] = (s[i] < s[i + 1] || (s[i] == s[i + 1] && t[i + 1])) ? 1 : 0;                    \
    memset(bkt, 0, ((size_t)K + 1) * sizeof(int32_t));                                         \
    for (int32_t i = 0; i < n; ++i) bkt[(int32_t)s[i] + 1]++;                                  \
    for (int32_t c = 0; c < K; ++c) bkt[c + 1] += bkt[c];      /* bkt[c] = start of bucket c */\
    /* ---- stage 1: sort the LMS substrings by induced sorting ---- */                        \
    for (int32_t i = 0; i < n; ++i) SA[i] = EMPTY;                                             \
    for (int32_t c = 0; c < K; ++c) ptr[c] = bkt[c + 1];                                       \
    int32_t n1 = 0;                                                                            \
    for (int32_t i = n - 1; i >= 1; --i)                                                       \
        if (t[i] && !t[i - 1]) { SA[--ptr[(int32_t)s[i]]] = i; ++n1; }                         \
    for (int pass = 0; pass < 2; ++pass) {                                                     \
        /* induce L, left to right; the virtual sentinel suffix induces n-1 first */           \
        for (int32_t c = 0; c < K; ++c) ptr[c] = bkt[c];                                       \
        SA[ptr[(int32_t)s[n - 1]]++] = n - 1;                                                  \
        for (int32_t i = 0; i < n; ++i) {                                                      \
            int32_t j = SA[i];                                                                 \
            if (j > 0 && !t[j - 1]) SA[ptr[(int32_t)s[j - 1]]++] = j - 1;                      \
        }                                                                                      \
        /* induce S, right to left */                                                          \
        for (int32_t c = 0; c < K; ++c) ptr[c] = bkt[c + 1];                                   \
        for (int32_t i = n - 1; i >= 0; --i) {                                                 \
            int32_t j = SA[i];                                                                 \
            if (j > 0 && t[j - 1]) SA[--ptr[(int32_t)s[j - 1]]] = j - 1;                       \
        }                                                                                      \
        if (pass == 1 || n1 == 0) break;                                                       \
        /* ---- stage 2: name the sorted LMS substrings, build the reduced string ---- */      \
        int32_t *P = (int32_t *)malloc((size_t)n1 * sizeof(int32_t));   /* LMS positions, text order */ \
        int32_t *s1 = (int32_t *)malloc((size_t)n1 * sizeof(int32_t));                         \
        int32_t *SA1 = (int32_t *)malloc((size_t)n1 * sizeof(int32_t));                        \
        int32_t *nameof = (int32_t *)malloc(((size_t)n / 2 + 1) * sizeof(int32_t));            \
        int32_t k = 0;                                                                         \
        for (int32_t i = 0; i < n; ++i) {                                                      \
            int32_t j = SA[i];                                                                 \
            if (j > 0 && t[j] && !t[j - 1]) SA1[k++] = j;     /* sorted LMS substrings */      \
        }                                                                                      \
        int32_t names = 0, prev = -1;                                                          \
        for (int32_t i = 0; i < n1; ++i) {                                                     \
            int32_t pos = SA1[i], diff = 0;                                                    \
            if (prev < 0) diff = 1;                                                            \
            else for (int32_t d = 0;; ++d) {                                                   \
                if (pos + d >= n || prev + d >= n) { diff = 1; break; }                        \
                if (s[pos + d] != s[prev + d] || t[pos + d] != t[prev + d]) { diff = 1; break; } \
                if (d > 0 && ((t[pos + d] && !t[pos + d - 1]) || (t[prev + d] && !t[prev + d - 1]))) break; \
            }                                                                                  \
            if (diff) { ++names; prev = pos; }                                                 \
            nameof[pos / 2] = names - 1;                                                       \
        }                                                                                      \
        k = 0;                                                                                 \
        for (int32_t i = 1; i < n; ++i)                                                        \
            if (t[i] && !t[i - 1]) { P[k] = i; s1[k] = nameof[i / 2]; ++k; }                   \
        free(nameof);                                                                          \
        /* ---- stage 3: sort the reduced string ---- */                                       \
        if (names < n1) sais_i32(s1, SA1, n1, names);                                          \
        else for (int32_t i = 0; i < n1; ++i) SA1[s1[i]] = i;                                  \
        /* ---- stage 4: seed the sorted LMS suffixes, then induce (pass 1) ---- */            \
        for (int32_t i = 0; i < n; ++i) SA[i] = EMPTY;                                         \
        for (int32_t c = 0; c < K; ++c) ptr[c] = bkt[c + 1];                                   \
        for (int32_t i = n1 - 1; i >= 0; --i) {                                                \
            int32_t j = P[SA1[i]];                                                             \
            SA[--ptr[(int32_t)s[j]]] = j;                                                      \
        }                                                                                      \
        free(P); free(s1); free(SA1);                                                          \
    }                                                                                          \
    free(t); free(bkt); free(ptr);                                                             \
}

DEFINE_SAIS(sais_i32, int32_t)
DEFINE_SAIS(sais_u8, uint8_t)

/* sa has n+1 entries; sa[0] = n, sa[1..] sorted suffix offsets        */
/* (buffer contract of reference src/saca.rs:9-15). n <= INT32_MAX-1.  */
ORACLE_API int32_t oracle_sais(const uint8_t *s, uint32_t *sa, int64_t n)
{
    if (n < 0 || n > 2147483646LL || (!s && n > 0) || !sa) return -1;
    sa[0] = (uint32_t)n;
    sais_u8(s, (int32_t *)(sa + 1), (int32_t)n, 256);
    return 0;
}

/* mirrors the C engine's signature the reference binds at src/saca.rs:14: */
/* int divsufsort(const unsigned char *T, int *SA, int n), SA has n entries */
ORACLE_API int32_t oracle_divsufsort(const uint8_t *T, int32_t *SA, int32_t n)
{
    if (n < 0 || (!T && n > 0) || (!SA && n > 0)) return -1;
    sais_u8(T, SA, n, 256);
    return 0;
}

/* ------------------------------------------------------------------ */
/* restatement of enable_buckets, reference src/sa.rs:89-119           */
/* (next-row f1).  bkt has 256*257+1 entries.                          */
/* ------------------------------------------------------------------ */
ORACLE_API void oracle_bucket_table(const uint8_t *s, int64_t n, uint32_t *bkt)
{
    const int64_t len = 256 * 257 + 1;
    memset(bkt, 0, (size_t)len * sizeof(uint32_t));
    bkt[0] = 1;                                                   /* src/sa.rs:98 */
    if (n > 0) {
        for (int64_t i = 0; i + 1 < n; ++i)                       /* src/sa.rs:100-105 */
            bkt[(int64_t)s[i] * 257 + ((int64_t)s[i + 1] + 1) + 1]++;
        bkt[(int64_t)s[n - 1] * 257 + 1]++;                       /* src/sa.rs:106-108 */
    }
    uint32_t sum = 0;                                             /* src/sa.rs:112-116 */
    for (int64_t i = 0; i < len; ++i) { sum += bkt[i]; bkt[i] = sum; }
}

/* ------------------------------------------------------------------ */
/* LCP array by Kasai et al. (analysis helper for DESIGN.md: how deep  */
/* must suffixes be compared before they are unique).  lcp[i] = LCP of */
/* the suffixes at sa[i-1] and sa[i], lcp[0] = 0; sa has n+1 entries.  */
/* ------------------------------------------------------------------ */
ORACLE_API int32_t oracle_lcp_kasai(const uint8_t *s, int64_t n, const uint32_t *sa, uint32_t *lcp)
{
    uint32_t *rank = (uint32_t *)malloc(((size_t)n + 1) * sizeof(uint32_t));
    if (!rank) return -2;
    for (int64_t i = 0; i <= n; ++i) rank[sa[i]] = (uint32_t)i;
    int64_t h = 0;
    lcp[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t r = rank[i];          /* r >= 1 because the empty suffix has rank 0 */
        int64_t j = sa[r - 1];
        while (i + h < n && j + h < n && s[i + h] == s[j + h]) ++h;
        lcp[r] = (uint32_t)h;
        if (h > 0) --h;
    }
    free(rank);
    return 0;
}
