/*
 * selftest.c -- the oracle and the corpus generators under AddressSanitizer + UndefinedBehaviorSanitizer
 * (`make -C oracle sanitize`, run by tests/test_oracle.py::test_sanitized_selftest).  GPU sanitizers are not
 * available on this pool, so the CPU side of the test infrastructure is what gets sanitized.
 * Checks: known answers of SURVEY.md section 8a (the last one is the reference's doc-test vector,
 * src/lib.rs:28-29), naive sort == SA-IS == both integrity checks on random and adversarial inputs,
 * the multi-threaded verifier, LCP statistics, every generator at ragged sizes.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int32_t oracle_naive_sa(const uint8_t *s, uint32_t *sa, int64_t n);
int32_t oracle_sais(const uint8_t *s, uint32_t *sa, int64_t n);
int32_t oracle_check_integrity(const uint8_t *s, int64_t n, const uint32_t *sa, int64_t sa_len);
int32_t oracle_verify_sa(const uint8_t *s, int64_t n, const uint32_t *sa, int64_t sa_len);
int32_t oracle_verify_sa_mt(const uint8_t *s, int64_t n, const uint32_t *sa, int64_t sa_len, int32_t threads);
int32_t oracle_lcp_stats(const uint8_t *s, int64_t n, const uint32_t *sa, uint64_t *out);
void oracle_bucket_table(const uint8_t *s, int64_t n, uint32_t *bkt);
void sa_gen_uniform(uint8_t *out, int64_t n, uint64_t seed);
void sa_gen_sigma(uint8_t *out, int64_t n, uint64_t seed, int32_t sigma, int32_t base);
void sa_gen_dna(uint8_t *out, int64_t n, uint64_t seed);
void sa_gen_dna_repeats(uint8_t *out, int64_t n, uint64_t seed, double repeat_fraction);
int32_t sa_gen_english(uint8_t *out, int64_t n, uint64_t seed, int32_t vocab);
int32_t sa_gen_english_corpus(uint8_t *out, int64_t n, uint64_t seed, int32_t vocab, double dup_fraction);

static int failures = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "selftest: %s failed at line %d\n", #c, __LINE__); ++failures; } } while (0)

static void check_text(const uint8_t *t, int64_t n)
{
    uint32_t *a = malloc(((size_t)n + 1) * 4), *b = malloc(((size_t)n + 1) * 4);
    CHECK(oracle_sais(t, a, n) == 0);
    if (n <= 20000) { CHECK(oracle_naive_sa(t, b, n) == 0); CHECK(memcmp(a, b, ((size_t)n + 1) * 4) == 0); }
    CHECK(oracle_verify_sa(t, n, a, n + 1) == 1);
    CHECK(oracle_verify_sa_mt(t, n, a, n + 1, 3) == 1);
    if (n <= 5000) CHECK(oracle_check_integrity(t, n, a, n + 1) == 1);
    if (n >= 3) {
        uint32_t x = a[1]; a[1] = a[2]; a[2] = x;
        CHECK(oracle_verify_sa(t, n, a, n + 1) == 0);
        CHECK(oracle_verify_sa_mt(t, n, a, n + 1, 2) == 0);
        x = a[1]; a[1] = a[2]; a[2] = x;
    }
    uint64_t st[32];
    CHECK(oracle_lcp_stats(t, n, a, st) == 0);
    free(a); free(b);
}

int main(void)
{
    static const struct { const char *s; int n; uint32_t sa[18]; } known[] = {
        { "", 0, { 0 } }, { "a", 1, { 1, 0 } }, { "aa", 2, { 2, 1, 0 } }, { "banana", 6, { 6, 5, 3, 1, 0, 4, 2 } },
        { "mississippi", 11, { 11, 10, 7, 4, 1, 0, 9, 8, 6, 3, 5, 2 } },
        { "splendid splendor", 17, { 17, 8, 7, 5, 14, 3, 12, 6, 2, 11, 4, 13, 15, 1, 10, 16, 0, 9 } },
    };
    for (size_t k = 0; k < sizeof known / sizeof known[0]; ++k) {
        uint32_t out[18];
        CHECK(oracle_sais((const uint8_t *)known[k].s, out, known[k].n) == 0);
        CHECK(memcmp(out, known[k].sa, ((size_t)known[k].n + 1) * 4) == 0);
        CHECK(oracle_naive_sa((const uint8_t *)known[k].s, out, known[k].n) == 0);
        CHECK(memcmp(out, known[k].sa, ((size_t)known[k].n + 1) * 4) == 0);
    }
    static const int sizes[] = { 0, 1, 2, 3, 63, 64, 65, 255, 257, 1000, 4095, 4097, 30011 };
    for (size_t k = 0; k < sizeof sizes / sizeof sizes[0]; ++k) {
        const int64_t n = sizes[k];
        uint8_t *t = malloc((size_t)n + 1);
        sa_gen_uniform(t, n, 1 + k); check_text(t, n);
        sa_gen_sigma(t, n, 2 + k, 3, 250); check_text(t, n);
        sa_gen_dna(t, n, 3 + k); check_text(t, n);
        sa_gen_dna_repeats(t, n, 4 + k, 0.3); check_text(t, n);
        CHECK(sa_gen_english(t, n, 5 + k, 500) == 0); check_text(t, n);
        CHECK(sa_gen_english_corpus(t, n, 6 + k, 500, 0.2) == 0); check_text(t, n);
        memset(t, 0xff, (size_t)n); check_text(t, n);
        for (int64_t i = 0; i < n; ++i) t[i] = (uint8_t)("ab"[i & 1]);
        check_text(t, n);
        free(t);
    }
    {   /* a corpus large enough for every generator layer (copies, second printings) */
        const int64_t n = (3 << 20) + 17;
        uint8_t *t = malloc((size_t)n);
        CHECK(sa_gen_english_corpus(t, n, 3, 50000, 0.08) == 0);
        check_text(t, n);
        uint32_t *bkt = malloc((256 * 257 + 1) * 4);
        oracle_bucket_table(t, n, bkt);
        CHECK(bkt[256 * 257] == (uint32_t)n + 1);
        free(bkt); free(t);
    }
    if (failures) { fprintf(stderr, "selftest: %d failures\n", failures); return 1; }
    puts("selftest ok");
    return 0;
}
