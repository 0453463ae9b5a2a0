/* tests/gen/sfxgen.c -- TEST / BENCH INPUT GENERATORS, not product code.
 *
 * Deterministic synthetic texts for BASELINE.json's configs, fast enough that bench.py can
 * make the 1 GB inputs inside a driver run (numpy needed minutes per GB):
 *   - splitmix64 everywhere, seed = 0x5AF1C5 + config index;
 *   - integer arithmetic only (integer Zipf / letter weights): the same bytes on every box,
 *     whatever the compiler or the thread count;
 *   - the output is made in independent 1 MiB blocks (block k has its own stream seeded from
 *     (seed, k)), so blocks are generated in parallel (OpenMP) and any prefix of whole blocks
 *     of a longer text equals the shorter text's blocks.
 * Python binding: tests/_gen.py (ctypes).  Build: tests/gen/Makefile.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK (1u << 20)
#define VOCAB 50000
#define MAXW 48 /* bytes per vocabulary word: 12 code points x 4 bytes */

static inline uint64_t sm_next(uint64_t* s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t sm_mix(uint64_t seed, uint64_t k)
{
    uint64_t s = seed ^ (k * 0xD6E8FEB86659FD93ull);
    return sm_next(&s);
}

/* English letter frequencies x100 (etaoinshrdlcumwfgypbvkjxqz) */
static const char LETTERS[27] = "etaoinshrdlcumwfgypbvkjxqz";
static const uint32_t LFREQ[26] = {1270, 910, 820, 750, 700, 670, 630, 610, 600, 430, 400, 280, 280,
                                   240, 240, 220, 200, 200, 190, 150, 100, 80, 15, 15, 10, 7};

typedef struct {
    uint8_t bytes[VOCAB][MAXW];
    uint8_t len[VOCAB];
    uint64_t zipf_cum[VOCAB]; /* cumulative floor(2^40 / rank) */
    uint64_t zipf_total;
} vocab_t;

static int put_utf8(uint8_t* o, uint32_t cp)
{
    if (cp < 0x80) { o[0] = (uint8_t)cp; return 1; }
    if (cp < 0x800) { o[0] = 0xC0 | (cp >> 6); o[1] = 0x80 | (cp & 0x3F); return 2; }
    if (cp < 0x10000) { o[0] = 0xE0 | (cp >> 12); o[1] = 0x80 | ((cp >> 6) & 0x3F); o[2] = 0x80 | (cp & 0x3F); return 3; }
    o[0] = 0xF0 | (cp >> 18); o[1] = 0x80 | ((cp >> 12) & 0x3F); o[2] = 0x80 | ((cp >> 6) & 0x3F); o[3] = 0x80 | (cp & 0x3F);
    return 4;
}
static uint32_t draw_letter(uint64_t* s)
{
    uint32_t tot = 0, i;
    for (i = 0; i < 26; i++) tot += LFREQ[i];
    uint32_t u = (uint32_t)(sm_next(s) % tot);
    for (i = 0; i < 26; i++) {
        if (u < LFREQ[i]) return (uint32_t)LETTERS[i];
        u -= LFREQ[i];
    }
    return 'e';
}
/* script of a word: 0 ASCII (40 %), 1 Cyrillic / Greek (20 %), 2 CJK U+4E00..U+9FFF (30 %), 3 U+1F300..U+1F5FF (10 %) */
static uint32_t draw_cp(uint64_t* s, int script)
{
    uint64_t r = sm_next(s);
    switch (script) {
    case 1: return (r & 1) ? 0x0410 + (uint32_t)((r >> 8) % 64) : 0x0391 + (uint32_t)((r >> 8) % 57);
    case 2: return 0x4E00 + (uint32_t)((r >> 8) % 0x5200);
    case 3: return 0x1F300 + (uint32_t)((r >> 8) % 0x300);
    default: return 0;
    }
}
static vocab_t* make_vocab(uint64_t seed, int mixed_scripts)
{
    vocab_t* v = (vocab_t*)malloc(sizeof(vocab_t));
    if (!v) return NULL;
    uint64_t s = sm_mix(seed, 0xC0FFEEull);
    uint64_t cum = 0;
    for (int w = 0; w < VOCAB; w++) {
        int ncp = 1 + (int)(sm_next(&s) % 12);
        int script = 0;
        if (mixed_scripts) {
            uint32_t u = (uint32_t)(sm_next(&s) % 100);
            script = u < 40 ? 0 : (u < 60 ? 1 : (u < 90 ? 2 : 3));
        }
        int len = 0;
        for (int c = 0; c < ncp; c++)
            len += put_utf8(v->bytes[w] + len, script == 0 ? draw_letter(&s) : draw_cp(&s, script));
        v->len[w] = (uint8_t)len;
        cum += (1ull << 40) / (uint64_t)(w + 1);
        v->zipf_cum[w] = cum;
    }
    v->zipf_total = cum;
    return v;
}
static inline int zipf_draw(const vocab_t* v, uint64_t* s)
{
    uint64_t u = sm_next(s) % v->zipf_total;
    int lo = 0, hi = VOCAB - 1; /* first index with zipf_cum > u */
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (v->zipf_cum[mid] > u) hi = mid; else lo = mid + 1;
    }
    return lo;
}

/* one block of word text: Zipf words joined by " " / ", " (8 %) / ". " (6 %, next word capitalised if
 * ASCII), the separator's space becomes '\n' once a line has passed 80 bytes, 2 % numeric tokens.
 * A token that does not fit the block is replaced by spaces (keeps UTF-8 whole). */
static void word_block(const vocab_t* v, uint64_t seed, uint64_t k, uint8_t* out, uint32_t size)
{
    uint64_t s = sm_mix(seed, k + 1);
    uint32_t pos = 0, col = 0;
    int cap = 1;
    uint8_t tok[MAXW + 8];
    while (pos < size) {
        uint32_t len;
        uint64_t r = sm_next(&s);
        if (r % 100 < 2) { /* numeric token: 1-6 digits */
            uint32_t nd = 1 + (uint32_t)((r >> 16) % 6), x = (uint32_t)(r >> 32);
            for (len = 0; len < nd; len++, x /= 10) tok[len] = (uint8_t)('0' + x % 10);
        } else {
            int w = zipf_draw(v, &s);
            len = v->len[w];
            memcpy(tok, v->bytes[w], len);
            if (cap && tok[0] >= 'a' && tok[0] <= 'z') tok[0] -= 32;
        }
        uint32_t p = (uint32_t)((r >> 8) % 100);
        cap = 0;
        if (p < 6) { tok[len++] = '.'; cap = 1; }
        else if (p < 14) tok[len++] = ',';
        col += len + 1;
        if (col >= 80) { tok[len++] = '\n'; col = 0; }
        else tok[len++] = ' ';
        if (pos + len > size) {
            memset(out + pos, ' ', size - pos);
            break;
        }
        memcpy(out + pos, tok, len);
        pos += len;
    }
}

static int run_blocks(uint8_t* out, uint64_t n, uint64_t seed, int mixed, int threads)
{
    vocab_t* v = make_vocab(seed, mixed);
    if (!v) return 1;
    const int64_t nblocks = (int64_t)((n + BLOCK - 1) / BLOCK);
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (int64_t k = 0; k < nblocks; k++) {
        const uint64_t off = (uint64_t)k * BLOCK;
        const uint32_t size = (uint32_t)(n - off < BLOCK ? n - off : BLOCK);
        word_block(v, seed, (uint64_t)k, out + off, size);
    }
    free(v);
    return 0;
}

/* config 3: English-like ASCII (sigma <= 96) */
int sfxgen_english(uint8_t* out, uint64_t n, uint64_t seed, int threads)
{
    return run_blocks(out, n, seed, 0, threads > 0 ? threads : 1);
}
/* config 5: valid UTF-8, words of 1-, 2-, 3- and 4-byte code points */
int sfxgen_utf8(uint8_t* out, uint64_t n, uint64_t seed, int threads)
{
    return run_blocks(out, n, seed, 1, threads > 0 ? threads : 1);
}

/* high-LCP input: `ndocs` distinct 1 MiB English-like documents, repeated round-robin until n bytes,
 * every copy with its own point mutations (one substituted byte about every `every` bytes):
 * near-duplicate documents, mean LCP in the hundreds */
int sfxgen_near_duplicates(uint8_t* out, uint64_t n, uint64_t seed, uint32_t ndocs, uint32_t every, int threads)
{
    if (ndocs == 0 || every == 0) return 1;
    vocab_t* v = make_vocab(seed, 0);
    if (!v) return 1;
    const int64_t nblocks = (int64_t)((n + BLOCK - 1) / BLOCK);
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (int64_t k = 0; k < nblocks; k++) {
        const uint64_t off = (uint64_t)k * BLOCK;
        const uint32_t size = (uint32_t)(n - off < BLOCK ? n - off : BLOCK);
        uint8_t* blk = out + off;
        if (size == BLOCK) {
            word_block(v, seed, (uint64_t)k % ndocs, blk, BLOCK);
        } else { /* a truncated last block is still a prefix of its document */
            uint8_t* tmp = (uint8_t*)malloc(BLOCK);
            word_block(v, seed, (uint64_t)k % ndocs, tmp, BLOCK);
            memcpy(blk, tmp, size);
            free(tmp);
        }
        if ((uint64_t)k >= ndocs) { /* the first copy of each document stays pristine */
            uint64_t s = sm_mix(seed ^ 0xA5A5A5A5ull, (uint64_t)k + 1);
            uint64_t p = sm_next(&s) % (2ull * every);
            while (p < size) {
                blk[p] = (uint8_t)('a' + sm_next(&s) % 26);
                p += 1 + sm_next(&s) % (2ull * every);
            }
        }
    }
    free(v);
    return 0;
}

/* config 2 / 4: uniform DNA, each 64-bit draw yields 32 symbols, 2 bits each, LSB first, 0->A 1->C 2->G 3->T
 * (one sequential splitmix64 stream: draw j is state seed + (j+1) * golden, so chunks start anywhere) */
int sfxgen_dna(uint8_t* out, uint64_t n, uint64_t seed, int threads)
{
    static const char sym[4] = {'A', 'C', 'G', 'T'};
    const int64_t nwords = (int64_t)((n + 31) / 32);
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t j = 0; j < nwords; j++) {
        uint64_t s = seed + (uint64_t)j * 0x9E3779B97F4A7C15ull;
        uint64_t w = sm_next(&s);
        const uint64_t base = (uint64_t)j * 32;
        const int cnt = (int)(n - base < 32 ? n - base : 32);
        for (int c = 0; c < cnt; c++) out[base + c] = (uint8_t)sym[(w >> (2 * c)) & 3];
    }
    return 0;
}

/* bytes [begin, begin + n) of the same stream (a rank's shard of ONE text: bench.py --total-size) */
int sfxgen_dna_slice(uint8_t* out, uint64_t begin, uint64_t n, uint64_t seed, int threads)
{
    static const char sym[4] = {'A', 'C', 'G', 'T'};
    const int64_t w0 = (int64_t)(begin / 32), w1 = (int64_t)((begin + n + 31) / 32);
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t j = w0; j < w1; j++) {
        uint64_t s = seed + (uint64_t)j * 0x9E3779B97F4A7C15ull;
        uint64_t w = sm_next(&s);
        for (int c = 0; c < 32; c++) {
            const uint64_t p = (uint64_t)j * 32 + (uint64_t)c;
            if (p >= begin && p < begin + n) out[p - begin] = (uint8_t)sym[(w >> (2 * c)) & 3];
        }
    }
    return 0;
}

/* config 5 queries: nq substrings of `text` (valid UTF-8) starting at code-point boundaries, 1-16 code
 * points long; the second half has its last code point replaced by another one of the same script
 * (mostly misses).  qbytes must hold 64 * nq bytes, qoff nq + 1 entries.  Returns total bytes via qoff[nq]. */
static inline int is_cont(uint8_t b) { return (b & 0xC0) == 0x80; }
static int cp_len(uint8_t lead) { return lead < 0x80 ? 1 : (lead < 0xE0 ? 2 : (lead < 0xF0 ? 3 : 4)); }
int sfxgen_queries(const uint8_t* text, uint64_t n, uint64_t nq, uint64_t seed, uint8_t* qbytes, uint64_t* qoff)
{
    uint64_t s = sm_mix(seed, 0x5EA7C4ull);
    uint64_t total = 0;
    qoff[0] = 0;
    for (uint64_t k = 0; k < nq; k++) {
        uint64_t p = n ? sm_next(&s) % n : 0;
        while (p < n && is_cont(text[p])) p++;
        uint32_t ncp = 1 + (uint32_t)(sm_next(&s) % 16);
        uint64_t e = p, last = p;
        for (uint32_t c = 0; c < ncp && e < n; c++) {
            last = e;
            e += (uint64_t)cp_len(text[e]);
        }
        if (e > n) e = n; /* (a valid text never cuts a code point) */
        uint64_t len = e - p;
        memcpy(qbytes + total, text + p, len);
        if (k >= nq / 2 && len > 0) {
            const int cl = (int)(e - last);
            uint8_t rep[4];
            uint32_t cp = cl == 1 ? (uint32_t)('a' + sm_next(&s) % 26) : draw_cp(&s, cl - 1);
            (void)put_utf8(rep, cp);
            memcpy(qbytes + total + (last - p), rep, (size_t)cl);
        }
        total += len;
        qoff[k + 1] = total;
    }
    return 0;
}
