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

/*
 * C3 (round 2): English-like text WITH the repeat structure of a real corpus.  Pizza&Chili `english` is a concatenation
 * of Project Gutenberg books: besides word statistics it has recurring phrases, boiler-plate sentences, quotations and
 * whole passages / editions that occur twice, which give it a mean LCP in the thousands and a maximum near 10^6
 * (english.200MB: mean 9 390, max 987 770 as published with the corpus).  An iid word model has none of that (the
 * round-1 generator above: mean LCP 9, max 29).  Layers, all seeded:
 *   1. vocabulary + Zipf word frequencies as above;
 *   2. word-bigram structure: every word has 8 preferred successors; with probability 0.5 the next word is one of them;
 *   3. a pool of 20 000 stock sentences (5-24 words); 3 % of the sentences are drawn from it;
 *   4. passages copied from earlier text: length Pareto(1.1) from 200 bytes to min(4 MiB, n / 8), about `dup_fraction`
 *      of the text, each copy with a word-substitution rate of 0 (exact), 1e-4, 1e-3 or 1e-2 per byte;
 *   5. two "second printings": at 55 % of the text a passage of n / 256 bytes is repeated verbatim, at 80 % one of
 *      n / 128 bytes with one substituted word per 100 000 bytes (what gives the real file its maximum LCP near 10^6).
 * Statistics of the benchmark instance (256 MiB, seed 3) are measured, not assumed: profiles/r02_corpus_stats.json.
 */
typedef struct {
    rng_t r;
    int32_t vocab;
    char *words; uint8_t *wlen; double *zcum; double z;
    int32_t *succ;              /* vocab x 8 preferred successors */
} eng_model;

enum { EMAXW = 16, ESUCC = 8 };

static int32_t eng_zipf(eng_model *m)
{
    double u = rng_unit(&m->r) * m->z;
    int32_t lo = 0, hi = m->vocab - 1;
    while (lo < hi) { int32_t mid = (lo + hi) >> 1; if (m->zcum[mid] < u) lo = mid + 1; else hi = mid; }
    return lo;
}

static int32_t eng_next_word(eng_model *m, int32_t prev)
{
    if (prev >= 0 && (rng_next(&m->r) & 1)) {
        /* preferred successor, the first ones more often (1/2, 1/4, ...) */
        uint64_t v = rng_next(&m->r);
        int k = 0;
        while (k < ESUCC - 1 && (v & 1)) { ++k; v >>= 1; }
        return m->succ[(size_t)prev * ESUCC + k];
    }
    return eng_zipf(m);
}

/* appends one sentence (words + separators, final ". " or ".\n"); returns the new length; *prev = last word */
static int64_t eng_sentence(eng_model *m, uint8_t *out, int64_t i, int64_t n, int nwords, int32_t *prev)
{
    int cap = 1;
    for (int w = 0; w < nwords && i < n; ++w) {
        const int32_t id = eng_next_word(m, *prev);
        *prev = id;
        const char *wd = m->words + (size_t)id * EMAXW;
        for (int j = 0; j < m->wlen[id] && i < n; ++j) {
            char c = wd[j];
            if (cap) { c = (char)(c - 32); cap = 0; }
            out[i++] = (uint8_t)c;
        }
        if (w + 1 < nwords) {
            const uint64_t v = rng_next(&m->r) % 100;
            if (v < 88) { if (i < n) out[i++] = ' '; }
            else if (v < 97) { if (i < n) out[i++] = ','; if (i < n) out[i++] = ' '; }
            else { if (i < n) out[i++] = ';'; if (i < n) out[i++] = ' '; }
        }
    }
    if (i < n) out[i++] = '.';
    if (i < n) out[i++] = (rng_next(&m->r) % 8 == 0) ? '\n' : ' ';
    return i;
}

GEN_API int32_t sa_gen_english_corpus(uint8_t *out, int64_t n, uint64_t seed, int32_t vocab, double dup_fraction)
{
    static const char letters[] = "etaoinshrdlcumwfgypbvkjxqz";
    static const double freq[26] = { 12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8,
                                     2.4, 2.4, 2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07 };
    if (vocab < 16) vocab = 16;
    if (dup_fraction < 0) dup_fraction = 0;
    if (dup_fraction > 0.9) dup_fraction = 0.9;
    eng_model m;
    rng_seed(&m.r, seed);
    m.vocab = vocab;
    m.words = (char *)malloc((size_t)vocab * EMAXW);
    m.wlen = (uint8_t *)malloc((size_t)vocab);
    m.zcum = (double *)malloc((size_t)vocab * sizeof(double));
    m.succ = (int32_t *)malloc((size_t)vocab * ESUCC * sizeof(int32_t));
    enum { POOL = 20000 };
    int64_t *pool_off = (int64_t *)malloc((POOL + 1) * sizeof(int64_t));
    uint8_t *pool = (uint8_t *)malloc((size_t)POOL * 25 * (EMAXW + 2));
    if (!m.words || !m.wlen || !m.zcum || !m.succ || !pool_off || !pool) {
        free(m.words); free(m.wlen); free(m.zcum); free(m.succ); free(pool_off); free(pool);
        return -2;
    }
    double lcum[26], tot = 0;
    for (int i = 0; i < 26; ++i) { tot += freq[i]; lcum[i] = tot; }
    m.z = 0;
    for (int32_t k = 0; k < vocab; ++k) {
        int len = 1 + (int)(log2((double)k + 2.0) * 0.55 + rng_unit(&m.r) * 3.0);
        if (len > EMAXW - 1) len = EMAXW - 1;
        m.wlen[k] = (uint8_t)len;
        for (int j = 0; j < len; ++j) {
            double u = rng_unit(&m.r) * tot;
            int c = 0;
            while (c < 25 && lcum[c] < u) ++c;
            m.words[(size_t)k * EMAXW + j] = letters[c];
        }
        m.z += 1.0 / ((double)k + 1.0);
        m.zcum[k] = m.z;
    }
    for (int32_t k = 0; k < vocab; ++k)
        for (int j = 0; j < ESUCC; ++j) m.succ[(size_t)k * ESUCC + j] = eng_zipf(&m);
    /* stock sentences */
    {
        const int64_t cap = (int64_t)POOL * 25 * (EMAXW + 2);
        int64_t o = 0;
        for (int q = 0; q < POOL; ++q) {
            pool_off[q] = o;
            int32_t prev = -1;
            o = eng_sentence(&m, pool, o, cap, 5 + (int)(rng_next(&m.r) % 20), &prev);
        }
        pool_off[POOL] = o;
    }
    const int64_t lmin = 200;
    int64_t lmax = n / 8;
    if (lmax > ((int64_t)4 << 20)) lmax = (int64_t)4 << 20;
    /* a copy of mean length E[L] is started after every sentence with probability p so that copies fill dup_fraction:
       E[L] of Pareto(1.1) truncated at lmax */
    double mean_len = 0;
    if (lmax > lmin) {
        const double a = 1.1, r = pow((double)lmin / (double)lmax, a - 1.0);
        mean_len = a / (a - 1.0) * (double)lmin * (1.0 - r) / (1.0 - pow((double)lmin / (double)lmax, a));
    }
    const double mean_sentence = 85.0;           /* bytes, measured for this model */
    const double p_copy = (mean_len > 0 && dup_fraction > 0) ? (dup_fraction / (1.0 - dup_fraction)) * mean_sentence / mean_len : 0.0;
    int64_t i = 0;
    int32_t prev = -1;
    int printings = 0;                              /* layer 5: done so far */
    while (i < n) {
        const int reprint = (n >= ((int64_t)1 << 20)) && ((printings == 0 && i >= n / 100 * 55) || (printings == 1 && i >= n / 100 * 80));
        if (reprint || (i > 4 * lmin && lmax > lmin && rng_unit(&m.r) < p_copy)) {
            /* passage copied from earlier text, with word substitutions */
            const double u = rng_unit(&m.r);
            const double a = 1.1, lo_a = pow((double)lmin, -a), hi_a = pow((double)lmax, -a);
            int64_t L = (int64_t)pow(lo_a - u * (lo_a - hi_a), -1.0 / a);
            if (L > i / 2) L = i / 2;
            if (L < lmin) L = lmin;
            int64_t src = (int64_t)(rng_next(&m.r) % (uint64_t)(i - L + 1));
            static const double rates[4] = { 0.0, 1e-4, 1e-3, 1e-2 };
            double rate = rates[rng_next(&m.r) & 3];
            if (reprint) {
                L = printings == 0 ? n / 256 : n / 128;
                src = printings == 0 ? n / 10 : n / 10 * 3;
                rate = printings == 0 ? 0.0 : 1e-5;
                ++printings;
            }
            const int64_t end_src = src + L;
            int64_t next_edit = rate > 0 ? src + (int64_t)(-log(1.0 - rng_unit(&m.r)) / rate) : end_src + 1;
            while (src < end_src && i < n) {
                if (src >= next_edit) {
                    /* replace the word at src by a fresh one */
                    while (src < end_src && out[src] != ' ') ++src;
                    const int32_t id = eng_zipf(&m);
                    const char *wd = m.words + (size_t)id * EMAXW;
                    for (int j = 0; j < m.wlen[id] && i < n; ++j) out[i++] = (uint8_t)wd[j];
                    next_edit = src + 1 + (int64_t)(-log(1.0 - rng_unit(&m.r)) / rate);
                    continue;
                }
                out[i++] = out[src++];
            }
            if (i < n) out[i++] = '\n';
            prev = -1;
            continue;
        }
        if (rng_next(&m.r) % 100 < 3) {
            const int q = (int)(rng_next(&m.r) % POOL);
            for (int64_t k = pool_off[q]; k < pool_off[q + 1] && i < n; ++k) out[i++] = pool[k];
            prev = -1;
            continue;
        }
        i = eng_sentence(&m, out, i, n, 5 + (int)(rng_next(&m.r) % 20), &prev);
    }
    free(m.words); free(m.wlen); free(m.zcum); free(m.succ); free(pool_off); free(pool);
    return 0;
}
