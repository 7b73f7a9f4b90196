/* mcd_host.c -- host-side helper of the CSV writer (plain C, no GPU): numpy's str() of a 1-D float32 row.
 *
 * The drivers' CSV (describe_og_neurons.py:112-116 / describe_broad_neurons.py:112-116) stores the 10 similarities
 * of a neuron as str(np.ndarray): numpy's array2string, i.e. per element the SHORTEST digit string that identifies
 * the float32 (Dragon4, unique=True) cut at 8 fractional digits, padded to common widths, wrapped at 75 columns.
 * At 9216 neurons that is 92 160 Dragon4 calls from Python: 0.1 s, 170x the whole HIP core.  This file produces the
 * same characters for the plain regime (finite, 1e-4 <= |x| < 2^24, max/min <= 999) with exact double arithmetic:
 * for such x and d <= 8, x*10^d and the rounding-interval ends (x +- half the gap to the neighbouring float32)*10^d
 * are exact doubles (<= 44 significant bits), so Dragon4's stopping rule -- stop at the first digit position where
 * the truncated or the incremented digit string lies strictly inside the interval, then round as it does -- is a few
 * comparisons.  Rows outside the regime return -1 and are formatted by numpy itself.
 * tests/test_host_logic_cpu.py compares the two character for character on random rows. */
#include <math.h>
#include <stdint.h>
#include <string.h>

static const double P10[9] = {1.0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8};
static const uint64_t I10[9] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull};

static int ndigits(uint64_t v) {
    int n = 1;
    while (v >= 10) { v /= 10; ++n; }
    return n;
}

/* |v| -> integer part, fractional digits F with d of them (trailing zeros trimmed) */
static void shortest(float v, uint64_t* ip, uint64_t* fp, int* dp) {
    const float av = fabsf(v);
    const double x = (double)av;
    const double hi = (x + (double)nextafterf(av, INFINITY)) * 0.5, lo = (x + (double)nextafterf(av, 0.0f)) * 0.5;
    int d = 0, low = 0, high = 0;
    double y = 0, fl = 0;
    for (;; ++d) {
        y = x * P10[d];
        fl = floor(y);
        low = fl > lo * P10[d];
        high = fl + 1.0 < hi * P10[d];
        if (low || high || d == 8) break;
    }
    int up;
    if (low == high) {                       /* both candidates identify x, or the 8-digit cut: nearest, ties to even */
        const double a = y - fl;
        up = a > 0.5 || (a == 0.5 && (((uint64_t)fl) & 1ull));
    } else {
        up = high;
    }
    uint64_t r = (uint64_t)fl + (uint64_t)up;
    while (d > 0 && r % 10 == 0) { r /= 10; --d; }
    *ip = r / I10[d];
    *fp = r % I10[d];
    *dp = d;
}

/* str(row) of numpy for a float32 row of n <= 64 elements; returns the length written (no terminator) or -1 */
int mcd_fmt_f32_row(const float* row, int n, char* out, int cap) {
    uint64_t ip[64], fp[64];
    int d[64], neg[64];
    if (n < 1 || n > 64) return -1;
    double amin = INFINITY, amax = 0.0;
    for (int i = 0; i < n; ++i) {
        const double a = fabs((double)row[i]);
        if (!(a >= 1.001e-4 && a < 16777216.0)) return -1;   /* also rejects NaN */
        if (a < amin) amin = a;
        if (a > amax) amax = a;
    }
    if (amax / amin > 999.0) return -1;
    int pl = 0, pr = 0;
    for (int i = 0; i < n; ++i) {
        shortest(row[i], &ip[i], &fp[i], &d[i]);
        neg[i] = row[i] < 0;
        const int li = ndigits(ip[i]) + neg[i];
        if (li > pl) pl = li;
        if (d[i] > pr) pr = d[i];
    }
    const int wlen = pl + 1 + pr;
    if (cap < 4 + n * (wlen + 2)) return -1;
    /* numpy _formatArray/_extendLine: lines start with one space, wrap before a word that would pass column 74 */
    char* o = out;
    int linelen = 1;          /* the leading space; replaced by '[' on the first line */
    *o++ = '[';
    for (int i = 0; i < n; ++i) {
        if (linelen + wlen > 74 && linelen > 1) {
            while (o > out && o[-1] == ' ') --o;   /* rstrip */
            *o++ = '\n';
            *o++ = ' ';
            linelen = 1;
        }
        char w[96];
        int k = wlen;
        /* fraction, right padded with spaces */
        for (int j = 0; j < pr - d[i]; ++j) w[--k] = ' ';
        uint64_t f = fp[i];
        for (int j = 0; j < d[i]; ++j) { w[--k] = (char)('0' + f % 10); f /= 10; }
        w[--k] = '.';
        uint64_t v = ip[i];
        do { w[--k] = (char)('0' + v % 10); v /= 10; } while (v);
        if (neg[i]) w[--k] = '-';
        while (k > 0) w[--k] = ' ';
        memcpy(o, w, (size_t)wlen);
        o += wlen;
        linelen += wlen;
        if (i + 1 < n) { *o++ = ' '; ++linelen; }
    }
    *o++ = ']';
    return (int)(o - out);
}

/* all rows of a [rows, n] matrix; lens[r] = length or -1; out has stride `cap` bytes per row */
void mcd_fmt_f32_rows(const float* a, int64_t rows, int n, char* out, int cap, int32_t* lens) {
    for (int64_t r = 0; r < rows; ++r) lens[r] = mcd_fmt_f32_row(a + r * n, n, out + r * cap, cap);
}

/* str(row) of numpy for an int64 row (describe_*_neurons.py: the `images` column, 5 image indices): every element
 * right-justified to the widest, one space between, no wrapping when the row fits 74 columns (else -1: numpy does it) */
int mcd_fmt_i64_row(const int64_t* row, int n, char* out, int cap) {
    if (n < 1 || n > 64) return -1;
    int w = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t v = row[i];
        if (v == INT64_MIN) return -1;
        const int l = ndigits((uint64_t)(v < 0 ? -v : v)) + (v < 0);
        if (l > w) w = l;
    }
    if ((w + 1) * n + 1 > 74 || cap < (w + 1) * n + 2) return -1;
    char* o = out;
    *o++ = '[';
    for (int i = 0; i < n; ++i) {
        char t[24];
        int k = w;
        const int64_t v = row[i];
        uint64_t a = (uint64_t)(v < 0 ? -v : v);
        do { t[--k] = (char)('0' + a % 10); a /= 10; } while (a);
        if (v < 0) t[--k] = '-';
        while (k > 0) t[--k] = ' ';
        memcpy(o, t, (size_t)w);
        o += w;
        if (i + 1 < n) *o++ = ' ';
    }
    *o++ = ']';
    return (int)(o - out);
}

void mcd_fmt_i64_rows(const int64_t* a, int64_t rows, int n, char* out, int cap, int32_t* lens) {
    for (int64_t r = 0; r < rows; ++r) lens[r] = mcd_fmt_i64_row(a + r * n, n, out + r * cap, cap);
}

/* ---- whole CSV rows of the og/broad drivers -------------------------------------------------------------------
 * layer,unit,description,similarity,images with Python's csv QUOTE_MINIMAL dialect (what pandas' to_csv drives): a
 * field is wrapped in '"' (inner '"' doubled) when it contains ',', '"', '\n' or '\r'.
 *   description = "[" + ", ".join(repr(word)) + "]"   (str() of the list of k concept strings; reprs come from Python)
 *   similarity  = str(np.float32[k])                  (mcd_fmt_f32_row)
 *   images      = str(np.int64[k_img])                (mcd_fmt_i64_row)
 * Returns the bytes written, or -1 when a row needs numpy's own formatting or `cap` is too small (the caller then
 * takes the csv-module path for the file). */
static char* emit_field(char* o, const char* t, int64_t n) {
    int special = 0;
    for (int64_t i = 0; i < n; ++i) {
        const char ch = t[i];
        if (ch == ',' || ch == '"' || ch == '\n' || ch == '\r') { special = 1; break; }
    }
    if (!special) { memcpy(o, t, (size_t)n); return o + n; }
    *o++ = '"';
    for (int64_t i = 0; i < n; ++i) {
        if (t[i] == '"') *o++ = '"';
        *o++ = t[i];
    }
    *o++ = '"';
    return o;
}

int64_t mcd_csv_og_rows(const char* layer, int layer_len, int64_t nrows, const int32_t* ids, int k,
                        const char* const* reprs, const int32_t* repr_lens, int n_words, const float* vals,
                        const int64_t* imgs, int k_img, char* out, int64_t cap) {
    char* o = out;
    char cell[4096];
    int64_t max_desc = 2;
    for (int j = 0; j < n_words; ++j)
        if (repr_lens[j] + 2 > max_desc) max_desc = repr_lens[j] + 2;
    max_desc = max_desc * k + 2;                       /* "[" + k * (repr + ", ") + "]" */
    if (k < 1 || k > 64 || k_img < 1 || k_img > 64) return -1;
    for (int64_t r = 0; r < nrows; ++r) {
        if ((o - out) + 2 * (layer_len + max_desc) + 3 * 4096 + 64 > cap) return -1;
        o = emit_field(o, layer, layer_len);
        *o++ = ',';
        {   /* unit */
            char t[24];
            int n = 0;
            int64_t v = r;
            do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
            while (n) *o++ = t[--n];
        }
        *o++ = ',';
        {   /* description: built after its own start position, then quoted in place if needed */
            char* d = o + max_desc + 8;                 /* scratch beyond the widest possible quoted field */
            char* e = d;
            *e++ = '[';
            for (int j = 0; j < k; ++j) {
                const int32_t w = ids[r * k + j];
                if (w < 0 || w >= n_words) return -1;
                if (j) { *e++ = ','; *e++ = ' '; }
                memcpy(e, reprs[w], (size_t)repr_lens[w]);
                e += repr_lens[w];
            }
            *e++ = ']';
            o = emit_field(o, d, e - d);
        }
        *o++ = ',';
        int n = mcd_fmt_f32_row(vals + r * k, k, cell, (int)sizeof(cell));
        if (n < 0) return -1;
        o = emit_field(o, cell, n);
        *o++ = ',';
        n = mcd_fmt_i64_row(imgs + r * k_img, k_img, cell, (int)sizeof(cell));
        if (n < 0) return -1;
        o = emit_field(o, cell, n);
        *o++ = '\n';
    }
    return (int64_t)(o - out);
}
