// host/tuning.hpp -- every environment knob of the engine, parsed in ONE place, once per build.
//
// All knobs are ROUTE switches: they choose between kernels / regimes that produce the same (unique) suffix array, and
// every numeric one is clamped to the range the kernels accept, so no value of any SA_AMD_* variable can change a
// result (tests/test_gpu_parity.py::test_any_environment_is_bit_exact).  Kernels that produce wrong orders on purpose
// (timing ablations) and the truncated "initial sort only" build exist only in libsuffix_array_amd_diag.so
// (-DSA_AMD_DIAG), never in the product library.
#pragma once
#include <cstdlib>
#include <cstdint>

namespace sa {

constexpr int64_t SPARSE_DIV_DEFAULT = 64;  // sparse refinement when at most n / 64 suffixes are tied after the initial sort
constexpr int GROUP_CAP_MAX = 1024;         // == GS_CAP (kernels/refine.hpp), checked there

inline bool env_flag(const char *name)
{
    const char *e = getenv(name);
    return e && *e && !(e[0] == '0' && e[1] == 0);
}

// integer knob: unparsable text = default; clamped to [lo, hi]
inline int64_t env_int(const char *name, int64_t def, int64_t lo, int64_t hi)
{
    const char *e = getenv(name);
    if (!e || !*e) return def;
    char *end = nullptr;
    const long long v = strtoll(e, &end, 10);
    if (end == e) return def;
    return v < lo ? lo : (v > hi ? hi : (int64_t)v);
}

struct Tuning {
    int sort_variant = 0;            // SA_AMD_SORT_VARIANT: tile-scatter kernel shape, 64-bit keys (all shapes give the same order)
    int sort32_variant = 0;          // SA_AMD_SORT32_VARIANT: the same for the 32-bit first stage
    int key_bits_max = 64;           // SA_AMD_KEY_BITS: 16..64 bits of packed key in the initial sort
    int group_cap = GROUP_CAP_MAX;   // SA_AMD_GROUP_CAP: largest group ordered in LDS, 2..1024
    int64_t sparse_div = SPARSE_DIV_DEFAULT;   // SA_AMD_SPARSE_DIV
    bool sparse_div_set = false;
    bool force_dense = false;        // SA_AMD_FORCE_DENSE
    bool no_text_rounds = false;     // SA_AMD_NO_TEXT_ROUNDS
    bool no_local_sort = false;      // SA_AMD_NO_LOCAL_SORT
    bool no_top32 = false;           // SA_AMD_NO_TOP32
    bool force_top32 = false;        // SA_AMD_FORCE_TOP32
    bool no_fused_finish = false;    // SA_AMD_NO_FUSED_FINISH
    bool fused64 = false;            // SA_AMD_FUSED64
    bool no_packed_text = false;     // SA_AMD_NO_PACKED_TEXT
    bool no_binned_isa = false;      // SA_AMD_NO_BINNED_ISA
    bool binned_isa_always = false;  // SA_AMD_BINNED_ISA_ALWAYS
    bool no_run_skip = false;        // SA_AMD_NO_RUN_SKIP: never skip a radix pass whose digit is the same for every element
    int64_t run_skip_min = (int64_t)1 << 25;   // SA_AMD_RUN_SKIP_MIN: smallest refinement sort that looks for such passes
    int64_t dense_rekey_min = (int64_t)1 << 22;   // SA_AMD_DENSE_REKEY_MIN: smallest whole-list global sort that is re-keyed by group index
    bool no_first_tail = false;      // SA_AMD_NO_FIRST_TAIL: the dense route's first ranks are head ranks (the first doubling round then rewrites every rank)
    bool no_repeat_probe = false;    // SA_AMD_NO_REPEAT_PROBE: never start rank doubling right after the initial sort
    int64_t binned_min = (int64_t)1 << 26;   // SA_AMD_BINNED_MIN: fewest (suffix, rank) pairs a round bins before scattering
    int chase = 7;                   // SA_AMD_CHASE: rank look-ups per member and dense doubling round (1 = plain doubling), 1..15
    int chase_big = 3;               // SA_AMD_CHASE_BIG: the same for lists of at least chase_big_min members (look-ups are what a long
    int64_t chase_big_min = (int64_t)1 << 23;   // SA_AMD_CHASE_BIG_MIN  list pays for; a short one pays for launches and read-backs)
    int scatter_levels = 0;          // SA_AMD_SCATTER_LEVELS: radix passes before a binned ISA write (0 = by size, 1, 2)
    int max_text_rounds = 4;         // SA_AMD_MAX_TEXT_ROUNDS: text-keyed rounds before rank doubling with a full ISA, 0..8
    bool no_split = false;           // SA_AMD_NO_SPLIT: never split giant groups around their majority key (always the whole-list radix sort)
    int64_t split_min = (int64_t)1 << 22;        // SA_AMD_SPLIT_MIN: smallest list that is split
    int64_t split_group_min = (int64_t)1 << 15;  // SA_AMD_SPLIT_GROUP_MIN: smallest average group of such a list
    bool no_gram_keys = false;       // SA_AMD_NO_GRAM_KEYS: never key the initial sort by ranks of g-grams
    int64_t gram_min_n = (int64_t)1 << 22;   // SA_AMD_GRAM_MIN_N: smallest text whose g-grams are looked at (measured: -4 % at 4 MiB, +5 % at 1 MiB)
    int gram_g = 0;                  // SA_AMD_GRAM_G: gram length (0 = the longest whose table fits, else 2..8, still subject to the fit)
    int gram_tail_max = 8;           // SA_AMD_GRAM_TAIL: most plain symbols behind the gram ranks of a gram key (as many as the 64 bits hold, 0..8; 0 = none)
    int top32_partners_x100 = 50;    // SA_AMD_TOP32_PARTNERS_X100: the 32-bit first stage is taken when a suffix shares its top 32 key bits with fewer than this / 100 others (sample estimate)
    int top32_collisions_x100 = 400; // SA_AMD_TOP32_COLLISIONS_X100: ... or when fewer than this / 100 share only the top 32 bits (chance collisions) and fewer than
                                     //   top32_partners_x100 / 100 share all key bits (repeats)
    int64_t top32_probe_min_n = (int64_t)1 << 21;   // SA_AMD_TOP32_PROBE_MIN_N: smallest text the entropy probe looks at (below: full keys; 16 Mi before the mid-size timings of round 3)
    int small_max = 8192;            // SA_AMD_SMALL_MAX: texts of up to this many bytes are built by ONE launch of one workgroup (kernels/small.hpp), 0..8192
    bool no_onesweep = false;        // SA_AMD_NO_ONESWEEP: the three-kernel radix pass (histogram, spine, chunk-owned scatter) instead of the single-pass one
    int onesweep_flags = 0;          // SA_AMD_ONESWEEP_FLAGS: scheduling switches of the single-pass scatter (kernels/onesweep.hpp, OnesweepPass::flags), same result
    int onesweep64_shape = 0;        // SA_AMD_ONESWEEP64_SHAPE / SA_AMD_ONESWEEP32_SHAPE: tile shape of the single-pass scatter (host/pipeline.hpp,
    int onesweep32_shape = 0;        //   os_shapes64 / os_shapes32; out of range = default)
    bool no_unary_shortcut = false;  // SA_AMD_NO_UNARY_SHORTCUT: a text of one byte value goes through the sort and the rounds like any other (214 ms at 256 MiB instead of 0.3)
    int64_t count_next_min_n = 20000000;            // SA_AMD_COUNT_NEXT_MIN_N: radix sorts of fewer pairs do not count the next digit inside a pass -- all digits in front of the
                                                    //   first pass (k_radix_hist_all) or, where a pass may be skipped, each in a kernel of its own, for [below_n, min_n) pairs
    int64_t count_next_below_n = 393216;            // SA_AMD_COUNT_NEXT_BELOW_N (0 / 0: always inside the pass before)
    int network_min = 256;           // SA_AMD_NETWORK_MIN: k_group_sort orders the groups of a tile that owns one of more members than this by a bitonic network (0: never)
    bool no_upfront_counts = false;  // SA_AMD_NO_UPFRONT_COUNTS: short radix sorts count their digits pass by pass, not all of them in front of the first pass
    bool no_flat_rule = false;       // SA_AMD_NO_FLAT_RULE: short texts with a flat byte histogram keep all 64 key bits
    bool no_posted_readback = false; // SA_AMD_NO_POSTED_READBACK: counts come back by copy command + stream synchronise, not by a posted write the host spins on
    bool no_defer = false;           // SA_AMD_NO_DEFER: every refinement round reads the local pass's counts back in its middle (two blocking read-backs per round instead of one)
    bool no_big_group_sort = false;  // SA_AMD_NO_BIG_GROUP_SORT: groups of 1025..16384 members go through the global radix sort, not k_group_sort_big
    bool no_text_keys = false;       // SA_AMD_NO_TEXT_KEYS: the first global pass of the bucket route always reads a key array (never the text itself)
    bool no_value_bits = false;      // SA_AMD_NO_VALUE_BITS: the unused top bits of the 32-bit stage's values never carry key bits into the bucket sort
    bool no_bucket_sort = false;     // SA_AMD_NO_BUCKET_SORT: the 32-bit first stage always takes four global passes (never two + the in-LDS bucket sort)
    int64_t bucket_min_n = (int64_t)1 << 25;   // SA_AMD_BUCKET_MIN_N: smallest text whose 32-bit first stage sorts the low 16 key bits bucket by bucket in LDS
    bool no_bucket_finish = false;   // SA_AMD_NO_BUCKET_FINISH: the suffixes tied on the top 32 key bits are ordered by k_finish_sorted in a pass of its own, not inside k_bucket_sort
    bool bucket_finish_always = false;   // SA_AMD_BUCKET_FINISH_ALWAYS: ... inside k_bucket_sort even with the 20-pairs-per-thread shapes (measured slower)
    int bucket_bits = 0;             // SA_AMD_BUCKET_BITS: key bits ordered by the two global passes in front of the bucket sort (0 = by text size, 16, 18)
    int bucket_shape = -1;           // SA_AMD_BUCKET_SHAPE: workgroup shape of that sort tried first (host/pipeline.hpp, bk_shapes; -1 = the smallest default one that holds the largest bucket)
#ifdef SA_AMD_DIAG
    // the sample sort of the 64-bit stage (kernels/sample_sort.hpp): a measured dead end of round 4 (profiles/r04_sample_sort_64.txt),
    // kept in the diagnostic library with its tests -- correct, not faster than the eight LSD passes
    bool sample_sort = false;        // SA_AMD_SAMPLE_SORT=1: the 64-bit stage's initial sort is the sample sort for texts of at least ...
    int64_t sample_sort_min_n = (int64_t)1 << 26;   // ... SA_AMD_SAMPLE_SORT_MIN_N bytes
    bool sample_merge = false;       // SA_AMD_SAMPLE_MERGE=1: the ordinary bucket by a merge sort in LDS (variant 3, slower) instead of LSD passes
    int sample_log = 0;              // SA_AMD_SAMPLE_LOG: log2 of the number of sampled keys (16..24; 0 = by the size of the text)
    bool timing_only_initial_sort = false;     // SA_AMD_TIMING_ONLY_INITIAL_SORT (diag library only: the array is NOT finished)
#endif

    static Tuning from_env(int n_sort_variants, int n_sort32_variants, int n_os64 = 1, int n_os32 = 1)
    {
        Tuning t;
        t.sort_variant = (int)env_int("SA_AMD_SORT_VARIANT", 0, 0, 1 << 20);
        if (t.sort_variant >= n_sort_variants) t.sort_variant = 0;
        t.sort32_variant = (int)env_int("SA_AMD_SORT32_VARIANT", 0, 0, 1 << 20);
        if (t.sort32_variant >= n_sort32_variants) t.sort32_variant = 0;
        t.key_bits_max = (int)env_int("SA_AMD_KEY_BITS", 64, 16, 64);
        t.group_cap = (int)env_int("SA_AMD_GROUP_CAP", GROUP_CAP_MAX, 2, GROUP_CAP_MAX);
        t.sparse_div_set = getenv("SA_AMD_SPARSE_DIV") != nullptr;
        t.sparse_div = env_int("SA_AMD_SPARSE_DIV", SPARSE_DIV_DEFAULT, 1, (int64_t)1 << 40);
        t.force_dense = env_flag("SA_AMD_FORCE_DENSE");
        t.no_text_rounds = env_flag("SA_AMD_NO_TEXT_ROUNDS");
        t.no_local_sort = env_flag("SA_AMD_NO_LOCAL_SORT");
        t.no_top32 = env_flag("SA_AMD_NO_TOP32");
        t.force_top32 = env_flag("SA_AMD_FORCE_TOP32");
        t.no_fused_finish = env_flag("SA_AMD_NO_FUSED_FINISH");
        t.fused64 = env_flag("SA_AMD_FUSED64");
        t.no_packed_text = env_flag("SA_AMD_NO_PACKED_TEXT");
        t.no_binned_isa = env_flag("SA_AMD_NO_BINNED_ISA");
        t.binned_isa_always = env_flag("SA_AMD_BINNED_ISA_ALWAYS");
        t.no_run_skip = env_flag("SA_AMD_NO_RUN_SKIP");
        t.run_skip_min = env_int("SA_AMD_RUN_SKIP_MIN", (int64_t)1 << 25, 1, (int64_t)1 << 40);
        t.no_first_tail = env_flag("SA_AMD_NO_FIRST_TAIL");
        t.no_repeat_probe = env_flag("SA_AMD_NO_REPEAT_PROBE");
        t.dense_rekey_min = env_int("SA_AMD_DENSE_REKEY_MIN", (int64_t)1 << 22, 1, (int64_t)1 << 40);
        t.max_text_rounds = (int)env_int("SA_AMD_MAX_TEXT_ROUNDS", 4, 0, 8);
        t.chase = (int)env_int("SA_AMD_CHASE", 7, 1, 15);
        t.chase_big = (int)env_int("SA_AMD_CHASE_BIG", 3, 1, 15);
        t.chase_big_min = env_int("SA_AMD_CHASE_BIG_MIN", (int64_t)1 << 23, 1, (int64_t)1 << 40);
        t.scatter_levels = (int)env_int("SA_AMD_SCATTER_LEVELS", 0, 0, 2);
        t.binned_min = env_int("SA_AMD_BINNED_MIN", (int64_t)1 << 26, 1, (int64_t)1 << 40);
        t.no_split = env_flag("SA_AMD_NO_SPLIT");
        t.split_min = env_int("SA_AMD_SPLIT_MIN", (int64_t)1 << 22, 1, (int64_t)1 << 40);
        t.split_group_min = env_int("SA_AMD_SPLIT_GROUP_MIN", (int64_t)1 << 15, 1, (int64_t)1 << 40);
        t.no_gram_keys = env_flag("SA_AMD_NO_GRAM_KEYS");
        t.gram_min_n = env_int("SA_AMD_GRAM_MIN_N", (int64_t)1 << 22, 1, (int64_t)1 << 40);
        t.gram_g = (int)env_int("SA_AMD_GRAM_G", 0, 0, 8);
        t.gram_tail_max = (int)env_int("SA_AMD_GRAM_TAIL", 8, 0, 8);
        t.top32_partners_x100 = (int)env_int("SA_AMD_TOP32_PARTNERS_X100", 50, 0, 1000000);
        t.top32_collisions_x100 = (int)env_int("SA_AMD_TOP32_COLLISIONS_X100", 400, 0, 1000000);
        t.top32_probe_min_n = env_int("SA_AMD_TOP32_PROBE_MIN_N", (int64_t)1 << 21, 8192, (int64_t)1 << 40);
        t.small_max = (int)env_int("SA_AMD_SMALL_MAX", 8192, 0, 8192);
        t.no_onesweep = env_flag("SA_AMD_NO_ONESWEEP");
        t.onesweep_flags = (int)env_int("SA_AMD_ONESWEEP_FLAGS", 0, 0, 255);
        t.onesweep64_shape = (int)env_int("SA_AMD_ONESWEEP64_SHAPE", 0, 0, 1 << 20);
        if (t.onesweep64_shape >= n_os64) t.onesweep64_shape = 0;
        t.onesweep32_shape = (int)env_int("SA_AMD_ONESWEEP32_SHAPE", 0, 0, 1 << 20);
        if (t.onesweep32_shape >= n_os32) t.onesweep32_shape = 0;
        t.no_big_group_sort = env_flag("SA_AMD_NO_BIG_GROUP_SORT");
        t.no_defer = env_flag("SA_AMD_NO_DEFER");
        t.no_posted_readback = env_flag("SA_AMD_NO_POSTED_READBACK");
        t.no_flat_rule = env_flag("SA_AMD_NO_FLAT_RULE");
        t.no_upfront_counts = env_flag("SA_AMD_NO_UPFRONT_COUNTS");
        t.network_min = (int)env_int("SA_AMD_NETWORK_MIN", 256, 0, 1 << 20);
        t.count_next_min_n = env_int("SA_AMD_COUNT_NEXT_MIN_N", 20000000, 0, (int64_t)1 << 40);
        t.count_next_below_n = env_int("SA_AMD_COUNT_NEXT_BELOW_N", 393216, 0, (int64_t)1 << 40);
        t.no_unary_shortcut = env_flag("SA_AMD_NO_UNARY_SHORTCUT");

        t.no_text_keys = env_flag("SA_AMD_NO_TEXT_KEYS");
        t.no_value_bits = env_flag("SA_AMD_NO_VALUE_BITS");
        t.no_bucket_sort = env_flag("SA_AMD_NO_BUCKET_SORT");
        t.bucket_min_n = env_int("SA_AMD_BUCKET_MIN_N", (int64_t)1 << 25, 1, (int64_t)1 << 40);
        t.no_bucket_finish = env_flag("SA_AMD_NO_BUCKET_FINISH");
        t.bucket_finish_always = env_flag("SA_AMD_BUCKET_FINISH_ALWAYS");
        t.bucket_bits = (int)env_int("SA_AMD_BUCKET_BITS", 0, 0, 18);
        if (t.bucket_bits != 16 && t.bucket_bits != 18) t.bucket_bits = 0;
        t.bucket_shape = (int)env_int("SA_AMD_BUCKET_SHAPE", -1, -1, 64);
#ifdef SA_AMD_DIAG
        t.sample_sort = env_flag("SA_AMD_SAMPLE_SORT");
        t.sample_merge = env_flag("SA_AMD_SAMPLE_MERGE");
        t.sample_sort_min_n = env_int("SA_AMD_SAMPLE_SORT_MIN_N", (int64_t)1 << 26, (int64_t)1 << 17, (int64_t)1 << 40);
        t.sample_log = (int)env_int("SA_AMD_SAMPLE_LOG", 0, 0, 24);
        if (t.sample_log != 0 && t.sample_log < 16) t.sample_log = 16;
        t.timing_only_initial_sort = env_flag("SA_AMD_TIMING_ONLY_INITIAL_SORT");
#endif
        return t;
    }
};

}  // namespace sa
