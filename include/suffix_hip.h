/* include/suffix_hip.h -- C ABI of libsuffix_hip.so, the MI355X (gfx950) engine
 * that replaces the suffix-array hot path of BurntSushi/suffix v1.3.0.
 *
 * The reference is pure Rust with no FFI of its own; these entry points are
 * what a Rust `extern "C"` block binds at the private seams listed below
 * (citations are /root/reference/src/table.rs; the Rust side a maintainer
 * would add is shown in INTEGRATION.md and rust/suffix-hip/src/lib.rs).
 *
 * Conventions
 *  - caller allocates, callee fills (mirrors `vec![0u32; n]` at :381);
 *  - every function returns an `int` status, 0 == SFX_OK (the reference panics,
 *    :380 / :117; the shim turns non-zero into panic!);
 *  - plain pointers and sizes only; `void* stream` is a hipStream_t (NULL = the
 *    default stream);
 *  - "*_dev" entry points take DEVICE pointers (inputs already resident in HBM,
 *    outputs left in HBM) plus a caller-provided device workspace; the others
 *    take HOST pointers and stage through HBM themselves;
 *  - re-entrant: no global mutable state except the optional profiler.
 */
#ifndef SUFFIX_HIP_H
#define SUFFIX_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
    SFX_OK = 0,
    SFX_ERR_ARG = 1,          /* null pointer / inconsistent sizes            */
    SFX_ERR_TOO_LARGE = 2,    /* n > u32::MAX on a u32 entry point (:380)     */
    SFX_ERR_NO_DEVICE = 3,    /* no HIP device visible                        */
    SFX_ERR_HIP = 4,          /* a HIP runtime call or kernel failed          */
    SFX_ERR_WORKSPACE = 5,    /* caller workspace smaller than *_workspace_bytes */
    SFX_ERR_INTERNAL = 6,     /* engine invariant violated (bug)              */
    SFX_ERR_NEEDS_RANKS = 7   /* range build only: the slice holds repeats too long for text-symbol
                                 refinement; build the whole suffix array instead (suffix_amd/dist.py does) */
};

const char* sfx_strerror(int status);
int sfx_device_count(void);
/* text of the last HIP error seen by this thread ("" if none) */
const char* sfx_last_hip_error(void);
/* Process-wide switches (additive; the defaults are what every number in DESIGN.md is quoted with; the library reads no
 * environment).  SFX_OPT_TINY_MAX: texts of up to this many bytes are built by ONE workgroup in ONE launch (sfx_tiny.hip:
 * alphabet, LSD radix sort of the suffixes' first key bits and the ordering of tied suffixes all inside one CU's LDS) --
 * default and largest value 16384, 0 = never (the tests use it to run small inputs through the general build as well).
 * sfx_set_option returns SFX_ERR_ARG for an unknown option or a value out of range. */
#define SFX_OPT_TINY_MAX 1
int      sfx_set_option(int option, uint64_t value);
uint64_t sfx_get_option(int option);

/* ---- SuffixTable::new -> sais_table (:378-386): suffix array, u32 indices ---- */
/* Host buffers.  Replaces the body of sais_table after `vec![0u32; n]` (:381-385).
 * n == 0 and n == 1 succeed (:395-402).  sa_out[r] = start of the r-th smallest
 * byte suffix, "shorter prefix sorts first" (naive_table :367-376). */
int sfx_build_sa_u32(const uint8_t* text, uint64_t n, uint32_t* sa_out);
/* Device-resident variant: d_text (n bytes) -> d_sa (n u32), all in HBM. */
uint64_t sfx_sa_workspace_bytes(uint64_t n);
int sfx_build_sa_u32_dev(const uint8_t* d_text, uint64_t n, uint32_t* d_sa,
                         void* d_workspace, uint64_t workspace_bytes, void* stream);

/* u64 index array (BASELINE config 4: "u64 indices").  SuffixTable itself is u32-only
 * (`Cow<[u32]>` :57, assert :380), so positions fit 32 bits: the u32 engine runs and the
 * array is widened on the device.  n > u32::MAX is SFX_ERR_TOO_LARGE as on the u32 entry. */
int sfx_build_sa_u64(const uint8_t* text, uint64_t n, uint64_t* sa_out);
int sfx_widen_u32_to_u64_dev(const uint32_t* d_in, uint64_t count, uint64_t* d_out, void* stream);

/* The host-pointer entry points keep a few released device buffers in a mutex-guarded pool
 * so that repeated calls on similar sizes skip hipMalloc/hipFree; this returns them. */
void sfx_release_cached_buffers(void);

/* ---- lcp_lens (:130-138 -> lcp_lens_quadratic :348-361): LCP array ---------- */
/* lcp_out[0] = 0, lcp_out[r] = |lcp(text[sa[r-1]..], text[sa[r]..])| in bytes.
 * The array is the same whichever route computes it: from 1 MiB of text up a sample of
 * adjacent pairs picks the reference's direct comparison (one window gather per suffix, capped;
 * the right choice for low-LCP text) or the linear Phi/PLCP form; a pair that reaches the cap
 * of the direct route sends the whole array through Phi/PLCP, so the cost stays linear in n.
 * The _dev entry synchronises the stream once or twice to read that choice back. */
int sfx_build_lcp_u32(const uint8_t* text, uint64_t n, const uint32_t* sa, uint32_t* lcp_out);
uint64_t sfx_lcp_workspace_bytes(uint64_t n);
int sfx_build_lcp_u32_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa,
                          uint32_t* d_lcp, void* d_workspace, uint64_t workspace_bytes,
                          void* stream);

/* ---- SuffixTable::new + lcp_lens in one call (:78-85 + :130-138; the pair suffix_tree/src/lib.rs:71,
 * :413 makes) -------------------------------------------------------------------
 * The same two arrays as sfx_build_sa_u32 followed by sfx_build_lcp_u32.  Where the initial key sort
 * already tells two neighbours apart (98 % of the pairs of uniform DNA) their LCP is read off the sorted
 * keys inside the build; only the remaining pairs are compared on the text. */
int sfx_build_sa_lcp_u32(const uint8_t* text, uint64_t n, uint32_t* sa_out, uint32_t* lcp_out);
uint64_t sfx_sa_lcp_workspace_bytes(uint64_t n);
int sfx_build_sa_lcp_u32_dev(const uint8_t* d_text, uint64_t n, uint32_t* d_sa, uint32_t* d_lcp,
                             void* d_workspace, uint64_t workspace_bytes, void* stream);

/* ---- positions / contains / any_position (:223-293), batched ---------------- */
/* Device-resident index = text + suffix array kept in HBM across calls. */
typedef struct sfx_index sfx_index;
/* sa == NULL => build it on the device.  Host pointers.  A caller-supplied table is checked for
 * entries >= n (SFX_ERR_ARG; the reference's from_parts is unchecked and "fails in weird ways", :105-107 --
 * a memory-safe panic there, so the engine must not read out of bounds either).  The index also holds a
 * BUCKET DIRECTORY: for every prefix of dbits bits of dense symbol codes (dbits = log2 n - 2, at most 28:
 * about one bucket per four suffixes, n bytes of HBM) the first rank whose suffix is not smaller -- a query
 * looks its own first dbits bits up and searches only inside that bucket -- and, memory permitting (17 n
 * bytes), a static 16-ary B+TREE over the first 16 bytes of every suffix in table order: a query of <= 16
 * bytes is answered from its nodes alone, a longer one bisects the ranks that share its first 16 bytes.
 * Batches of >= 4096 queries keep a per-thread scratch list (12 bytes per query) between calls. */
int sfx_index_create(const uint8_t* text, uint64_t n, const uint32_t* sa, sfx_index** out);
/* the same over text and suffix array that already live in HBM (borrowed, not copied: keep them alive and
 * unchanged while the index exists); only the directory and the tree are built.  Queries with device buffers: */
int sfx_index_create_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, void* stream, sfx_index** out);
int sfx_index_query_dev(const sfx_index* ix, const uint8_t* d_qbytes, const uint64_t* d_qoff, uint64_t nq,
                        uint32_t* d_start, uint32_t* d_end, uint8_t* d_found, uint32_t* d_any, void* stream);
void sfx_index_destroy(sfx_index* ix);
uint64_t sfx_index_len(const sfx_index* ix);
/* copy the index's suffix array back to the host (n u32) */
int sfx_index_table(const sfx_index* ix, uint32_t* sa_out);
/* Queries are concatenated in `qbytes`; query k is qbytes[qoff[k] .. qoff[k+1]).
 * positions(q) == table[start_out[k] .. end_out[k]) exactly as :244-258; an empty
 * result is reported as start == end == 0.  found_out[k] = contains(q) (:197-199);
 * any_out[k] = any_position(q) or UINT32_MAX for None (:279-293; which occurrence
 * is "arbitrary" by contract, :261-262).  Output arrays may be NULL to skip. */
int sfx_positions_batch(const sfx_index* ix, const uint8_t* qbytes, const uint64_t* qoff,
                        uint64_t nq, uint32_t* start_out, uint32_t* end_out);
int sfx_contains_batch(const sfx_index* ix, const uint8_t* qbytes, const uint64_t* qoff,
                       uint64_t nq, uint8_t* found_out, uint32_t* any_out);
/* all-device variant of the above (d_qbytes, d_qoff, outputs in HBM) */
int sfx_query_batch_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa,
                        const uint8_t* d_qbytes, const uint64_t* d_qoff, uint64_t nq,
                        uint32_t* d_start, uint32_t* d_end, uint8_t* d_found,
                        uint32_t* d_any, void* stream);

/* ---- suffix-tree topology from SA + LCP (suffix_tree/src/lib.rs:392-505 `to_suffix_tree`), flat ------
 * The internal nodes of the suffix tree are the lcp-intervals of the LCP array; the reference finds them
 * with a serial stack sweep, the engine with nearest-smaller-value searches over a min-pyramid.  For every
 * BOUNDARY p in [0, n) (between ranks p - 1 and p; value lcp[p]) the call writes
 *   lb[p], rb[p]   the rank range of the node the boundary belongs to (its string depth is lcp[p]);
 *   node[p]        that node's id = its leftmost boundary with that depth (0 = the root: [0, n-1], depth 0,
 *                  to which every boundary with lcp[p] == 0 belongs);
 *   parent[p]      the id of that node's parent (UINT32_MAX for every boundary of the root: p = 0 and every p with
 *                  lcp[p] == 0 -- the root is nobody's child, its own included);
 * and for every RANK r:  leaf_parent[r] = id of the node the leaf of suffix sa[r] hangs under.
 * Node k's edge label is text[sa[lb[k]] + depth(parent) .. sa[lb[k]] + depth(k)); children are the nodes /
 * leaves whose parent is k -- the same tree as `to_suffix_tree`, as arrays.  All device pointers, n u32 each.
 * Boundary 0 (in front of the first suffix) has depth 0 by definition: d_lcp[0] is not looked at. */
uint64_t sfx_lcp_intervals_workspace_bytes(uint64_t n);
int sfx_lcp_intervals_dev(const uint32_t* d_lcp, uint64_t n, uint32_t* d_lb, uint32_t* d_rb, uint32_t* d_node,
                          uint32_t* d_parent, uint32_t* d_leaf_parent, void* d_workspace, uint64_t workspace_bytes,
                          void* stream);
/* ---- generalized suffix array (README.md:60-74): documents concatenated with a separator byte into one
 * text, one SuffixTable over it; a match position is mapped back to (document, offset) by a binary search
 * over the sorted document start offsets.  doc / offset may be NULL to skip. */
int sfx_doc_lookup_dev(const uint32_t* d_positions, uint64_t count, const uint64_t* d_doc_starts, uint64_t ndocs,
                       uint32_t* d_doc, uint32_t* d_offset, void* stream);

/* ---- range-partitioned construction (multi-GPU, one rank per GPU) ----------- */
/* Every rank holds the whole text in HBM (all-gathered over RCCL) and owns the
 * text shard [shard_begin, shard_end).
 * Step 1  sfx_byte_histogram_dev: 256 u64 byte counts of the rank's shard
 *         (cf. Bins::find_sizes :686-704).  Ranks all-reduce(sum) them; the
 *         result defines the dense symbol codes, identically on every rank.
 * Step 2  sfx_key_histogram_dev: every suffix has a key = its first k symbols
 *         packed big-endian (k fixed by the global byte COUNTS and n -- an alphabet of
 *         more than 16 symbols that is used unevenly gets the wider key: the rule looks
 *         at the order-0 entropy -- so steps 2 and 4 must be given the same all-reduced
 *         counts, not presence flags); this counts,
 *         for the suffixes starting in the rank's shard, the top `top_bits`
 *         (<= 14) bits of that key into 2^top_bits u64 bins.  Ranks all-reduce
 *         them = the bucket-boundary histogram exchange.
 * Step 3  the host splits the bins into contiguous, balanced ranges, one per rank.
 * Step 4  sfx_build_sa_range_u32_dev: sort ONLY the suffixes whose bin lies in
 *         [bin_lo, bin_hi).  Writes them, fully sorted, to d_sa_part (capacity
 *         entries available) and their number to *count_out (host pointer):
 *         d_sa_part is this rank's contiguous slice of the global suffix array. */
int sfx_byte_histogram_dev(const uint8_t* d_text, uint64_t shard_begin, uint64_t shard_end,
                           uint64_t* d_bins256, void* stream);
int sfx_key_histogram_dev(const uint8_t* d_text, uint64_t n, uint64_t shard_begin,
                          uint64_t shard_end, const uint64_t* d_global_byte_bins256,
                          int top_bits, uint64_t* d_bins, void* stream);
uint64_t sfx_sa_range_workspace_bytes(uint64_t n, uint64_t capacity);
int sfx_build_sa_range_u32_dev(const uint8_t* d_text, uint64_t n,
                               const uint64_t* d_global_byte_bins256, int top_bits,
                               uint32_t bin_lo, uint32_t bin_hi, uint64_t capacity,
                               uint32_t* d_sa_part, uint64_t* count_out, void* d_workspace,
                               uint64_t workspace_bytes, void* stream);

/* Packed-text variant of steps 1 and 4 (saves xGMI volume: the all-gather moves
 * bits/8 of the raw bytes, and no rank packs the whole text):
 *   sfx_pack_text_dev   symbols of d_text[0..count) re-coded with the GLOBAL alphabet
 *         (d_global_byte_bins256) and packed big-endian, floor(32/bits) per u32 word,
 *         bits = ceil(log2 sigma); writes exactly n_words words (zeros past the text).
 *         A shard whose length is a multiple of floor(32/bits) packs into count /
 *         floor(32/bits) words that can be concatenated across ranks; the concatenation
 *         must end in >= 3 zero words.  d_scratch256: 256 bytes of device scratch.
 *   sfx_build_sa_range_packed_u32_dev   step 4 on that packed text (n = symbols). */
int sfx_pack_text_dev(const uint8_t* d_text, uint64_t count, const uint64_t* d_global_byte_bins256,
                      uint8_t* d_scratch256, uint32_t* d_words, uint64_t n_words, void* stream);
int sfx_build_sa_range_packed_u32_dev(const uint32_t* d_packed, uint64_t n,
                                      const uint64_t* d_global_byte_bins256, int top_bits,
                                      uint32_t bin_lo, uint32_t bin_hi, uint64_t capacity,
                                      uint32_t* d_sa_part, uint64_t* count_out, void* d_workspace,
                                      uint64_t workspace_bytes, void* stream);

/* Per-rank pieces of the partitioned index (every rank holds the text and ONE contiguous
 * slice d_sa_part[0..count) of the suffix array):
 *  - LCP of the slice: lcp_part[r] = |lcp(text[sa_part[r-1]..], text[sa_part[r]..])|, with
 *    prev_suffix = the last suffix of the previous rank's slice for r == 0 (UINT32_MAX for
 *    the first slice, giving 0 as :352 does).  Direct comparison, exactly :348-361.
 *  - queries against the slice: start/end are positions INSIDE the slice (0/0 if the
 *    slice holds no match); the global interval is the concatenation over ranks.        */
int sfx_build_lcp_range_u32_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa_part,
                                uint64_t count, uint32_t prev_suffix, uint32_t* d_lcp_part,
                                void* stream);
int sfx_query_batch_range_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa_part,
                              uint64_t count, const uint8_t* d_qbytes, const uint64_t* d_qoff,
                              uint64_t nq, uint32_t* d_start, uint32_t* d_end, uint8_t* d_found,
                              uint32_t* d_any, void* stream);

/* ---- memory-system micro-benchmarks (the scatter/gather roofline
 * must be measured) -------------------------------------------------------------
 * Allocates its own device buffers of about `bytes`, runs `reps` timed launches
 * (after one warm-up) and reports algorithmic GB/s (10^9 B/s) in *gbps_out.
 *   SFX_MB_COPY        streaming 16-byte copy (read + write counted)
 *   SFX_MB_SCATTER4    random 4-byte writes into a `bytes`-sized array
 *                      (head_insert/tail_insert :723-736, ISA[suffix] = rank)
 *   SFX_MB_GATHER1     random 1-byte reads (T[s-1] in induce, :429)
 *   SFX_MB_GATHER4     random 4-byte reads (ISA[suffix + h])
 *   SFX_MB_RUNSCATTER  runs of `param` bytes (power of two >= 8) copied to
 *                      pseudo-random places, run-aligned if param2 != 0, else
 *                      offset by 8 bytes: the write side of a radix pass        */
enum { SFX_MB_COPY = 0, SFX_MB_SCATTER4 = 1, SFX_MB_GATHER1 = 2, SFX_MB_GATHER4 = 3, SFX_MB_RUNSCATTER = 4 };
int sfx_microbench(int kind, uint64_t bytes, int param, int param2, int reps, double* gbps_out);

/* ---- profiling (per-kernel HIP-event timing; off by default) ---------------- */
/* When enabled every kernel launch is bracketed by hipEvents on its stream.
 * sfx_profile_report writes up to `cap` records and returns how many exist. */
typedef struct {
    char     name[48];
    uint64_t launches;
    double   total_ms;
    double   algo_bytes;      /* algorithmic bytes summed over launches */
} sfx_kernel_stat;
void sfx_profile_enable(int on);
void sfx_profile_reset(void);
int  sfx_profile_report(sfx_kernel_stat* out, int cap);
/* per-build statistics of the most recent SA construction on this thread */
typedef struct {
    uint64_t n;
    uint32_t sigma;           /* distinct byte values                          */
    uint32_t bits_per_symbol;
    uint32_t key_bits;        /* width of the initial k-mer key (32 or 64)     */
    uint32_t symbols_per_key; /* k (compressed 64-bit keys: the average, 64 / mean code length) */
    uint32_t rounds;          /* refinement rounds after the initial sort      */
    uint32_t reserved;        /* bit 0: sfx_build_sa_lcp_u32 stopped reading LCP values off the sort (most suffixes tied on the initial key);
                                 bit 1: the initial keys are context codes (a code per class of the preceding symbol) */
    uint64_t active_after_initial;
    uint64_t radix_passes;
    uint64_t elements_sorted; /* sum over passes of elements moved             */
    uint64_t small_bucket_resolved; /* suffixes placed by direct comparison of small buckets */
    uint64_t tile_sorted;     /* elements ordered by the in-LDS bucket sort, summed over rounds   */
    uint64_t large_sorted;    /* elements of buckets too large for LDS, summed over rounds        */
    uint32_t text_rounds;     /* refinement rounds keyed by text symbols                          */
    uint32_t rank_rounds;     /* refinement rounds keyed by ranks (prefix doubling)               */
    uint64_t deep_gathers;    /* 64-bit key gathers of the deep text rounds (one random line each) */
} sfx_build_stats;
/* Writes sizeof(sfx_build_stats) bytes AS OF THE LIBRARY'S header.  The struct has grown (round 3 appended deep_gathers:
 * 104 -> 112 bytes) and may grow again, always at its end: a consumer compiled against an older header must use
 * sfx_build_stats_read instead, or it is written past its struct. */
void sfx_last_build_stats(sfx_build_stats* out);
/* The size-aware form: copies the first min(out_bytes, sizeof(sfx_build_stats)) bytes and returns the library's
 * sizeof(sfx_build_stats) (so a caller can tell which trailing fields it got).  out == NULL: only the size. */
uint64_t sfx_build_stats_read(void* out, uint64_t out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SUFFIX_HIP_H */
