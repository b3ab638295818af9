/*
 * textgen.c -- seeded synthetic corpora for the benchmark configs of
 * BASELINE.json / SURVEY.md section 8d.  Own fixed PRNG (splitmix64 seeding a
 * xoshiro256**), so the bytes are identical on every toolchain and the CPU
 * baseline and the GPU path see the same input.
 *
 * The families mirror the reference's benchmark data (reference
 * benches/utils.rs:17-45, :206-210): uniform random bytes over 0..=255, and
 * offline stand-ins for the Pizza&Chili `english` and `dna` files, which the
 * reference downloads over HTTP (benches/utils.rs:153) and which are
 * unavailable here.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define GEN_API __attribute__((visibility("default")))

typedef struct { uint64_t s[4]; } rng_t;

static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static void rng_seed(rng_t *r, uint64_t seed)
{
    for (int i = 0; i < 4; ++i) r->s[i] = splitmix64(&seed);
}

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t rng_next(rng_t *r)
{
    uint64_t *s = r->s;
    uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}

static inline double rng_unit(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

/* C1 / C2 / C5: iid uniform bytes over 0..=255 (reference benches/utils.rs:206-210) */
GEN_API void sa_gen_uniform(uint8_t *out, int64_t n, uint64_t seed)
{
    rng_t r; rng_seed(&r, seed);
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t v = rng_next(&r); memcpy(out + i, &v, 8); }
    if (i < n) { uint64_t v = rng_next(&r); memcpy(out + i, &v, (size_t)(n - i)); }
}

/* iid uniform over the first `sigma` byte values starting at `base` (test helper) */
GEN_API void sa_gen_sigma(uint8_t *out, int64_t n, uint64_t seed, int32_t sigma, int32_t base)
{
    rng_t r; rng_seed(&r, seed);
    if (sigma < 1) sigma = 1;
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)(base + (int32_t)(rng_next(&r) % (uint64_t)sigma));
}

/* C4: iid uniform over A C G T */
GEN_API void sa_gen_dna(uint8_t *out, int64_t n, uint64_t seed)
{
    static const char acgt[4] = { 'A', 'C', 'G', 'T' };
    rng_t r; rng_seed(&r, seed);
    int64_t i = 0;
    while (i < n) {
        uint64_t v = rng_next(&r);
        for (int k = 0; k < 32 && i < n; ++k, v >>= 2) out[i++] = (uint8_t)acgt[v & 3];
    }
}

/* C4 harder variant: plant copies of earlier segments (1..100 KiB) with 1 % point mutations */
GEN_API void sa_gen_dna_repeats(uint8_t *out, int64_t n, uint64_t seed, double repeat_fraction)
{
    static const char acgt[4] = { 'A', 'C', 'G', 'T' };
    sa_gen_dna(out, n, seed);
    rng_t r; rng_seed(&r, seed ^ 0x5eedULL);
    int64_t planted = 0, target = (int64_t)(repeat_fraction * (double)n);
    while (planted < target && n > 4096) {
        int64_t len = 1024 + (int64_t)(rng_next(&r) % (99 * 1024));
        if (len > n / 4) len = n / 4;
        int64_t dst = (int64_t)(rng_next(&r) % (uint64_t)(n - len));
        if (dst < len) continue;
        int64_t src = (int64_t)(rng_next(&r) % (uint64_t)(dst - len + 1));
        memmove(out + dst, out + src, (size_t)len);
        for (int64_t k = 0; k < len / 100; ++k)
            out[dst + (int64_t)(rng_next(&r) % (uint64_t)len)] = (uint8_t)acgt[rng_next(&r) & 3];
        planted += len;
    }
}

/*
 * C3: English-like text.  Vocabulary of `vocab` lowercase words whose letters
 * follow English letter frequencies; word ids drawn from Zipf(s=1.0); separators
 * are mostly spaces with occasional ", " / ". " / newlines; the word after a
 * sentence end is capitalised.
 */
GEN_API int32_t sa_gen_english(uint8_t *out, int64_t n, uint64_t seed, int32_t vocab)
{
    static const char letters[] = "etaoinshrdlcumwfgypbvkjxqz";
    static const double freq[26] = { 12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8,
                                     2.4, 2.4, 2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07 };
    if (vocab < 16) vocab = 16;
    rng_t r; rng_seed(&r, seed);
    double lcum[26], tot = 0;
    for (int i = 0; i < 26; ++i) { tot += freq[i]; lcum[i] = tot; }
    /* vocabulary: rank k gets length ~ 2 + log-ish growth, so frequent words are short */
    enum { MAXW = 16 };
    char *words = (char *)malloc((size_t)vocab * MAXW);
    uint8_t *wlen = (uint8_t *)malloc((size_t)vocab);
    double *zcum = (double *)malloc((size_t)vocab * sizeof(double));
    if (!words || !wlen || !zcum) { free(words); free(wlen); free(zcum); return -2; }
    double z = 0;
    for (int32_t k = 0; k < vocab; ++k) {
        int len = 1 + (int)(log2((double)k + 2.0) * 0.55 + rng_unit(&r) * 3.0);
        if (len > MAXW - 1) len = MAXW - 1;
        wlen[k] = (uint8_t)len;
        for (int j = 0; j < len; ++j) {
            double u = rng_unit(&r) * tot;
            int c = 0;
            while (c < 25 && lcum[c] < u) ++c;
            words[(size_t)k * MAXW + j] = letters[c];
        }
        z += 1.0 / ((double)k + 1.0);
        zcum[k] = z;
    }
    int64_t i = 0;
    int cap = 1;
    while (i < n) {
        double u = rng_unit(&r) * z;
        int32_t lo = 0, hi = vocab - 1;
        while (lo < hi) { int32_t mid = (lo + hi) >> 1; if (zcum[mid] < u) lo = mid + 1; else hi = mid; }
        const char *w = words + (size_t)lo * MAXW;
        for (int j = 0; j < wlen[lo] && i < n; ++j) {
            char c = w[j];
            if (cap) { c = (char)(c - 32); cap = 0; }
            out[i++] = (uint8_t)c;
        }
        uint64_t v = rng_next(&r) % 100;
        if (v < 80) { if (i < n) out[i++] = ' '; }
        else if (v < 90) { if (i < n) out[i++] = ','; if (i < n) out[i++] = ' '; }
        else if (v < 98) { if (i < n) out[i++] = '.'; if (i < n) out[i++] = ' '; cap = 1; }
        else { if (i < n) out[i++] = '.'; if (i < n) out[i++] = '\n'; cap = 1; }
    }
    free(words); free(wlen); free(zcum);
    return 0;
}
