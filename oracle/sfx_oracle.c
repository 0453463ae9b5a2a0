/* oracle/sfx_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the hot path of BurntSushi/suffix, used ONLY as
 * the checker in tests/, in __graft_entry__.smoke() and as bench.py's
 * `cpu_baseline` leg.  Nothing under suffix_amd/ may link, import or call it.
 *
 * Parity pin: the reference (Rust) cannot be compiled in this image (no
 * rustc/cargo), so this restatement is pinned against (1) every known-answer
 * literal of the reference's own tests (/root/reference/tests/tests.rs:22-70,
 * :100-168, :181-213 and the doc-tests), (2) the reference's own oracle,
 * `naive_table` (src/table.rs:367-376), restated below as orc_naive_sa, on
 * thousands of random byte/UTF-8 strings, and (3) the two FASTA fixtures'
 * SA/LCP sha256 values derived from the definition.
 * LCP: the reference has no test asserting lcp_lens() values ("parity
 * unpinned" by the reference itself); the definition at src/table.rs:352-359
 * is the pin.
 *
 * Functions and the reference code each one follows:
 *   orc_naive_sa        src/table.rs:367-376  naive_table
 *   orc_sais            src/table.rs:378-386  sais_table -> :388-574 sais
 *   orc_lcp_quadratic   src/table.rs:348-365  lcp_lens_quadratic + lcp_len
 *   orc_lcp_kasai       src/table.rs:314-346  (commented-out linear variant)
 *   orc_positions       src/table.rs:223-259  positions + :900-914 binary_search
 *   orc_any_position    src/table.rs:279-293  any_position (contains = found)
 *   orc_suffix_tree_sweep  suffix_tree/src/lib.rs:392-505  to_suffix_tree (the SA + LCP sweep, as flat arrays)
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { TY_ASC = 0, TY_DESC = 1, TY_VALLEY = 2 };      /* SuffixType :580-585 */

/* PartialEq for SuffixType, src/table.rs:663-669: Valley counts as Ascending */
static inline int ty_same_class(uint8_t a, uint8_t b)
{
    return (a == TY_DESC) == (b == TY_DESC);
}

/* SuffixTypes (:576-578) + Bins (:671-675), shared by all recursion levels */
typedef struct {
    uint8_t  *types;
    uint32_t *sizes;  size_t sizes_len;
    uint32_t *ptrs;   size_t ptrs_len, ptrs_cap;
    uint32_t *alphas; size_t n_alpha, alphas_cap;
} orc_ctx;

static void orc_grow_sizes(orc_ctx *cx, size_t want)
{
    size_t cap = cx->sizes_len ? cx->sizes_len : 256;
    while (cap < want) cap *= 2;
    cx->sizes = (uint32_t *)realloc(cx->sizes, cap * sizeof(uint32_t));
    memset(cx->sizes + cx->sizes_len, 0, (cap - cx->sizes_len) * sizeof(uint32_t));
    cx->sizes_len = cap;
}
static void orc_reserve_alphas(orc_ctx *cx, size_t want)
{
    if (want > cx->alphas_cap) {
        cx->alphas = (uint32_t *)realloc(cx->alphas, want * sizeof(uint32_t));
        cx->alphas_cap = want;
    }
}
static void orc_reset_ptrs(orc_ctx *cx, size_t len)
{
    if (len > cx->ptrs_cap) {
        cx->ptrs = (uint32_t *)realloc(cx->ptrs, len * sizeof(uint32_t));
        cx->ptrs_cap = len;
    }
    memset(cx->ptrs, 0, len * sizeof(uint32_t));
    cx->ptrs_len = len;
}
/* find_head_pointers :706-712 / find_tail_pointers :714-720 */
static void orc_head_ptrs(orc_ctx *cx)
{
    uint32_t sum = 0;
    for (size_t k = 0; k < cx->n_alpha; k++) {
        uint32_t c = cx->alphas[k];
        cx->ptrs[c] = sum;
        sum += cx->sizes[c];
    }
}
static void orc_tail_ptrs(orc_ctx *cx)
{
    uint32_t sum = 0;
    for (size_t k = 0; k < cx->n_alpha; k++) {
        uint32_t c = cx->alphas[k];
        sum += cx->sizes[c];
        cx->ptrs[c] = sum - 1;
    }
}
/* head_insert :723-727 / tail_insert :730-736 */
static inline void orc_head_insert(orc_ctx *cx, uint32_t *sa, uint32_t i, uint32_t c)
{
    sa[cx->ptrs[c]++] = i;
}
static inline void orc_tail_insert(orc_ctx *cx, uint32_t *sa, uint32_t i, uint32_t c)
{
    uint32_t p = cx->ptrs[c];
    sa[p] = i;
    if (p > 0) cx->ptrs[c] = p - 1;
}

static void sais_u32(orc_ctx *cx, const uint32_t *t, uint32_t n, uint32_t *sa);

#define CH_T uint8_t
#define FN(name) name##_u8
#include "sais_level.inc"
#undef CH_T
#undef FN
#define CH_T uint32_t
#define FN(name) name##_u32
#include "sais_level.inc"
#undef CH_T
#undef FN

/* sais_table, src/table.rs:378-386.  Returns 0, or -1 if n > u32::MAX (:380). */
int orc_sais(const uint8_t *text, uint64_t n, uint32_t *sa)
{
    if (n > 0xFFFFFFFFull) return -1;
    orc_ctx cx;
    memset(&cx, 0, sizeof cx);
    cx.types = (uint8_t *)malloc(n ? n : 1);
    sais_u8(&cx, text, (uint32_t)n, sa);
    free(cx.types); free(cx.sizes); free(cx.ptrs); free(cx.alphas);
    return 0;
}

/* naive_table, src/table.rs:367-376: comparison sort of all byte suffixes;
 * a proper prefix sorts first (Rust slice Ord). */
typedef struct { const uint8_t *text; uint64_t n; } cmp_env;
static int cmp_suffix(const void *pa, const void *pb, void *envp)
{
    const cmp_env *e = (const cmp_env *)envp;
    uint32_t a = *(const uint32_t *)pa, b = *(const uint32_t *)pb;
    uint64_t la = e->n - a, lb = e->n - b;
    int r = memcmp(e->text + a, e->text + b, la < lb ? la : lb);
    if (r) return r;
    return (la < lb) ? -1 : (la > lb);
}
int orc_naive_sa(const uint8_t *text, uint64_t n, uint32_t *sa)
{
    if (n > 0xFFFFFFFFull) return -1;
    for (uint64_t i = 0; i < n; i++) sa[i] = (uint32_t)i;
    cmp_env env = { text, n };
    qsort_r(sa, n, sizeof(uint32_t), cmp_suffix, &env);
    return 0;
}

/* lcp_lens_quadratic + lcp_len, src/table.rs:348-365 */
void orc_lcp_quadratic(const uint8_t *text, uint64_t n, const uint32_t *sa, uint32_t *lcp)
{
    if (n == 0) return;
    lcp[0] = 0;
    for (uint64_t r = 1; r < n; r++) {
        uint64_t a = sa[r - 1], b = sa[r], k = 0;
        while (a + k < n && b + k < n && text[a + k] == text[b + k]) k++;
        lcp[r] = (uint32_t)k;
    }
}

/* The linear-time variant the reference keeps commented out (:314-346), on
 * bytes.  `inv` is scratch of n entries (the inverse SA the reference builds at
 * :131-134 and then never uses). */
void orc_lcp_kasai(const uint8_t *text, uint64_t n, const uint32_t *sa,
                   uint32_t *inv, uint32_t *lcp)
{
    if (n == 0) return;
    for (uint64_t r = 0; r < n; r++) inv[sa[r]] = (uint32_t)r;
    lcp[0] = 0;
    uint64_t h = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint32_t r = inv[i];
        if (r == 0) { h = 0; continue; }
        uint64_t j = sa[r - 1];
        while (i + h < n && j + h < n && text[i + h] == text[j + h]) h++;
        lcp[r] = (uint32_t)h;
        if (h) h--;
    }
}

/* Rust `a <= b` / `a < b` on byte slices: lexicographic, prefix sorts first. */
static int bytes_cmp(const uint8_t *a, uint64_t la, const uint8_t *b, uint64_t lb)
{
    int r = memcmp(a, b, la < lb ? la : lb);
    if (r) return r;
    return (la < lb) ? -1 : (la > lb);
}
static int starts_with(const uint8_t *s, uint64_t ls, const uint8_t *q, uint64_t lq)
{
    return ls >= lq && memcmp(s, q, lq) == 0;
}

/* positions, src/table.rs:223-259 (with binary_search :900-914): writes the
 * half-open SA interval [*start, *end).  Empty result => *start == *end == 0. */
void orc_positions(const uint8_t *text, uint64_t n, const uint32_t *sa,
                   const uint8_t *q, uint64_t m, uint64_t *start, uint64_t *end)
{
    *start = *end = 0;
    if (n == 0 || m == 0) return;                                   /* :228-229 */
    const uint8_t *s0 = text + sa[0];       uint64_t l0 = n - sa[0];
    const uint8_t *sl = text + sa[n - 1];   uint64_t ll = n - sa[n - 1];
    if ((bytes_cmp(q, m, s0, l0) < 0 && !starts_with(s0, l0, q, m)) /* :230-231 */
        || bytes_cmp(q, m, sl, ll) > 0)                             /* :232 */
        return;
    uint64_t lo = 0, hi = n;                                        /* :244-246 */
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (bytes_cmp(q, m, text + sa[mid], n - sa[mid]) <= 0) hi = mid; else lo = mid + 1;
    }
    uint64_t st = lo;
    lo = 0; hi = n - st;                                            /* :247-250 */
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        uint32_t s = sa[st + mid];
        if (!starts_with(text + s, n - s, q, m)) hi = mid; else lo = mid + 1;
    }
    uint64_t en = st + lo;
    if (st > en) return;                                            /* :254-255 */
    if (st == en) return;
    *start = st; *end = en;
}

/* any_position, src/table.rs:279-293: std binary_search_by comparing the first
 * m bytes of each suffix with the query.  Returns 1 and *pos when found.  The
 * index returned is "arbitrary" by contract (:261-262); callers may only check
 * that it is a real occurrence. */
/* positions() for a batch of queries (query k = qbytes[qoff[k] .. qoff[k+1])): the same routine in a
 * loop, spread over the host cores with OpenMP -- a checker convenience for the 10^6-query config */
void orc_positions_batch(const uint8_t *text, uint64_t n, const uint32_t *sa, const uint8_t *qbytes,
                         const uint64_t *qoff, uint64_t nq, uint32_t *start, uint32_t *end)
{
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t k = 0; k < (int64_t)nq; k++) {
        uint64_t s, e;
        orc_positions(text, n, sa, qbytes + qoff[k], qoff[k + 1] - qoff[k], &s, &e);
        start[k] = (uint32_t)s;
        end[k] = (uint32_t)e;
    }
}

int orc_any_position(const uint8_t *text, uint64_t n, const uint32_t *sa,
                     const uint8_t *q, uint64_t m, uint32_t *pos)
{
    if (m == 0) return 0;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        uint32_t s = sa[mid];
        uint64_t ls = n - s, take = ls < m ? ls : m;
        int r = bytes_cmp(text + s, take, q, m);
        if (r == 0) { *pos = s; return 1; }
        if (r < 0) lo = mid + 1; else hi = mid;
    }
    return 0;
}

/* to_suffix_tree, /root/reference/suffix_tree/src/lib.rs:392-505, as flat arrays.  The reference walks the suffix
 * array with the LCP array and keeps the path from the root to the last leaf (`ancestor_lcp_len` :393-411 climbs it):
 * an lcp equal to the depth of the node it stops at adds a leaf there (:423-442), a larger one splits the edge to that
 * node's right-most child with a new internal node of depth lcp (:443-499).  Every internal node is an lcp-interval;
 * here the open path is an explicit stack and the tree comes out as, per boundary p (between ranks p - 1 and p):
 *   lb[p], rb[p]   rank range of the node that boundary p belongs to (depth lcp[p]; the root for lcp[p] == 0, p == 0)
 *   node[p]        id of that node = the boundary that CREATED it (:455 `Node::internal`), its leftmost one of that depth; 0 = root
 *   parent[p]      id of its parent (UINT32_MAX for the root)
 * and per rank r: leaf_parent[r] = id of the node the leaf of suffix sa[r] ends up under (a leaf that is the right-most
 * child when the next boundary is deeper moves under the new node, :470-473).  lcp[0] is taken as 0. */
void orc_suffix_tree_sweep(const uint32_t *lcp, uint64_t n, uint32_t *lb, uint32_t *rb, uint32_t *node,
                           uint32_t *parent, uint32_t *leaf_parent)
{
    if (n == 0) return;
    typedef struct { uint32_t depth, lb, id, parent; } open_node;
    open_node *stack = (open_node *)malloc((size_t)(n + 1) * sizeof(open_node));
    uint32_t *rb_of = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));      /* rb of the node with id = boundary */
    uint32_t *par_of = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    uint32_t *lb_of = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    size_t sp = 0;
    stack[sp++] = (open_node){0u, 0u, 0u, UINT32_MAX};                       /* the root (SuffixTree::init :418) */
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t l = i ? lcp[i] : 0u;
        uint32_t child_lb = (uint32_t)(i ? i - 1 : 0);                       /* leftmost rank under the right-most child of vins */
        int64_t child_id = -1;                                               /* ... that child, if it is an internal node */
        while (stack[sp - 1].depth > l) {                                    /* ancestor_lcp_len :397-409 */
            const open_node c = stack[--sp];
            rb_of[c.id] = (uint32_t)(i - 1);                                 /* closed: nothing after rank i - 1 is below it */
            lb_of[c.id] = c.lb;
            par_of[c.id] = c.parent;
            child_lb = c.lb;
            child_id = (int64_t)c.id;
        }
        open_node *vins = &stack[sp - 1];
        if (vins->depth == l) {                                              /* Ordering::Equal :423: a new leaf under vins */
            node[i] = vins->id;
            leaf_parent[i] = vins->id;
        } else {                                                             /* Ordering::Less :443: split, new internal node */
            const open_node in = {l, child_lb, (uint32_t)i, vins->id};
            stack[sp++] = in;
            node[i] = (uint32_t)i;
            leaf_parent[i] = (uint32_t)i;
            /* the old right-most child hangs under the new node now (:470-473): leaf i - 1, or the node closed last */
            if (child_id >= 0) par_of[child_id] = (uint32_t)i;
            else leaf_parent[i - 1] = (uint32_t)i;
        }
    }
    while (sp) {
        const open_node c = stack[--sp];
        rb_of[c.id] = (uint32_t)(n - 1);
        lb_of[c.id] = c.lb;
        par_of[c.id] = c.parent;
    }
    for (uint64_t p = 0; p < n; p++) {
        const uint32_t id = node[p];
        lb[p] = lb_of[id];
        rb[p] = rb_of[id];
        parent[p] = par_of[id];
    }
    free(stack); free(rb_of); free(par_of); free(lb_of);
}
