// a9: the result tables of a phenotype (phenotypes.get_ML_df, modeling.py:1112-1145): <test>_results_<pheno>.tsv -- one line per
// k-mer that passed the scan, ordered by the p-value STRINGS (:1128), each line the k-mer, round(stat, 2), "%.2E" % p,
// [the two group means,] the number of samples with the k-mer and "| " + their names -- and its first n_kmers lines again as
// <...>_top<n>.tsv.  Host code: no kernel; it is here because the Python of r01-r03 spent 0.6 s of a 1.7-s run on it
// (140,000 lines, 30 M sample names).  The lines are formatted by a few threads, chunk by chunk, and
// written in order; every byte is what the Python writer produced -- repr() of a double is re-stated below and the golden
// TSVs of the reference (tests/golden/ds_*) check the whole files.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "psk_internal.h"

namespace {

// Python's repr(float): the shortest digits that round-trip, in positional notation while the decimal exponent is in
// [-4, 16), else d.ddde+XX; "x.0" for whole numbers
void py_repr(double v, std::string &out)
{
    if (std::isnan(v)) { out += "nan"; return; }
    if (std::isinf(v)) { out += v < 0 ? "-inf" : "inf"; return; }
    char sci[40];
    auto r = std::to_chars(sci, sci + sizeof sci, v, std::chars_format::scientific);
    *r.ptr = 0;
    const char *e = strchr(sci, 'e');
    const int e10 = atoi(e + 1);
    if (e10 >= -4 && e10 < 16) {
        char fix[64];
        auto q = std::to_chars(fix, fix + sizeof fix, v, std::chars_format::fixed);
        *q.ptr = 0;
        out += fix;
        if (!strchr(fix, '.')) out += ".0";
    } else out += sci;   // (to_chars writes at least two exponent digits with their sign, as Python does)
}

inline double round2(double x) { return std::nearbyint(x * 100.0) / 100.0; }   // numpy's round(x, 2): rint(x * 100) / 100

struct Tables {
    int kind, k, wpr, n_samples;
    const uint64_t *words, *bits;
    const double *stat, *p, *mean_x, *mean_y;
    const int32_t *n_with;
    const uint8_t *valid;
    const char *names;
    const int64_t *name_off;
    std::vector<uint64_t> vmask;   // the valid samples as bit words
};

void format_row(const Tables &t, int64_t i, const char *pstr, std::string &out)
{
    static const char code[4] = {'A', 'C', 'G', 'T'};
    char km[33];
    for (int b = 0; b < t.k; b++) km[b] = code[(t.words[i] >> (2 * (t.k - 1 - b))) & 3];
    out.append(km, t.k);
    out += '\t';
    py_repr(round2(t.stat[i]), out);
    out += '\t';
    out += pstr;
    out += '\t';
    if (t.kind == 1) {
        py_repr(round2(t.mean_x[i]), out);
        out += '\t';
        py_repr(round2(t.mean_y[i]), out);
        out += '\t';
    }
    char num[16];
    auto r = std::to_chars(num, num + sizeof num, t.n_with[i]);
    out.append(num, r.ptr - num);
    out += "\t|";
    const uint64_t *row = t.bits + (size_t)i * t.wpr;
    for (int w = 0; w < t.wpr; w++) {
        uint64_t x = row[w] & t.vmask[w];
        while (x) {
            const int s = w * 64 + __builtin_ctzll(x);
            x &= x - 1;
            out += ' ';
            out.append(t.names + t.name_off[s], (size_t)(t.name_off[s + 1] - t.name_off[s]));
        }
    }
    out += '\n';
}

int write_all(FILE *f, const std::string &s) { return fwrite(s.data(), 1, s.size(), f) == s.size() ? 0 : -1; }

}  // namespace

extern "C" int psk_write_result_tables(psk_ctx *ctx, const char *path, const char *path_top, int64_t n_top, const char *header, int kind,
                                       int64_t n_rows, const uint64_t *words, int k, const double *stat, const double *p,
                                       const double *mean_x, const double *mean_y, const int32_t *n_with, const uint64_t *bits, int wpr,
                                       int n_samples, const uint8_t *valid, const char *names, const int64_t *name_off, int64_t *order_out)
{
    auto fail = [&](int code, const char *what) { return ctx ? psk_fail(ctx, code, "psk_write_result_tables: %s", what) : code; };
    if (!path || !header || n_rows < 0 || k < 1 || k > 32 || wpr < 1 || n_samples < 1 || n_samples > 64 * wpr || (kind != 0 && kind != 1))
        return fail(PSK_EINVAL, "bad argument");
    if (n_rows && (!words || !stat || !p || !n_with || !bits || !valid || !names || !name_off || !order_out || (kind == 1 && (!mean_x || !mean_y))))
        return fail(PSK_EINVAL, "null buffer");
    Tables t;
    t.kind = kind; t.k = k; t.wpr = wpr; t.n_samples = n_samples; t.words = words; t.bits = bits; t.stat = stat; t.p = p;
    t.mean_x = mean_x; t.mean_y = mean_y; t.n_with = n_with; t.valid = valid; t.names = names; t.name_off = name_off;
    t.vmask.assign(wpr, 0);
    for (int s = 0; s < n_samples; s++)
        if (valid[s]) t.vmask[s >> 6] |= 1ull << (s & 63);
    // the p-value strings, and the order of the lines: by the STRING (the reference sorts the DataFrame's columns by the
    // formatted p-value row), ties by k-mer text -- for one k that is the order of the words
    std::vector<char> ps((size_t)n_rows * 16);
    for (int64_t i = 0; i < n_rows; i++) snprintf(&ps[(size_t)i * 16], 16, "%.2E", p[i]);
    for (int64_t i = 0; i < n_rows; i++) order_out[i] = i;
    std::stable_sort(order_out, order_out + n_rows, [&](int64_t a, int64_t b) {
        const int c = strcmp(&ps[(size_t)a * 16], &ps[(size_t)b * 16]);
        return c ? c < 0 : words[a] < words[b];
    });
    // the lines, formatted in chunks by a few threads
    unsigned hw = std::thread::hardware_concurrency();
    const int n_thr = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), std::max<int64_t>(1, n_rows / 2048));
    std::vector<std::string> part(n_thr);
    auto work = [&](int q) {
        const int64_t lo = n_rows * q / n_thr, hi = n_rows * (q + 1) / n_thr;
        std::string &o = part[q];
        o.reserve((size_t)(hi - lo) * 64);
        for (int64_t r = lo; r < hi; r++) format_row(t, order_out[r], &ps[(size_t)order_out[r] * 16], o);
    };
    std::vector<std::thread> thr;
    for (int q = 1; q < n_thr; q++) thr.emplace_back(work, q);
    work(0);
    for (auto &th : thr) th.join();
    FILE *f = fopen(path, "wb");
    if (!f) return fail(PSK_EINVAL, "cannot open the result table for writing");
    int bad = fputs(header, f) < 0 || fputc('\n', f) == EOF;
    for (int q = 0; q < n_thr && !bad; q++) bad = write_all(f, part[q]);
    bad = fclose(f) != 0 || bad;
    if (bad) return fail(PSK_EINVAL, "writing the result table failed");
    if (path_top) {
        std::string top;
        const int64_t m = std::min(n_top, n_rows);
        for (int64_t r = 0; r < m; r++) format_row(t, order_out[r], &ps[(size_t)order_out[r] * 16], top);
        f = fopen(path_top, "wb");
        if (!f) return fail(PSK_EINVAL, "cannot open the top table for writing");
        bad = fputs(header, f) < 0 || fputc('\n', f) == EOF || write_all(f, top);
        bad = fclose(f) != 0 || bad;
        if (bad) return fail(PSK_EINVAL, "writing the top table failed");
    }
    return PSK_OK;
}

// a11: the lines of k-mers_and_coefficients_in_<model>_model_<pheno>.txt (write_model_coefficients_to_file, modeling.py:1414-1455)
// appended to `path` (the caller has written the header line): k-mer, repr(coefficient), the number of samples that carry it,
// "| " + their names.  The Python writer joined a million names per 2,048-sample run (0.07 s of a 1.0-s run).
extern "C" int psk_write_model_coefficients(psk_ctx *ctx, const char *path, int64_t n_kmers, const char *kmers, const int64_t *kmer_off,
                                            const double *coefs, const int64_t *x, int64_t n_samples, const char *names,
                                            const int64_t *name_off)
{
    auto fail = [&](int code, const char *what) { return ctx ? psk_fail(ctx, code, "psk_write_model_coefficients: %s", what) : code; };
    if (!path || n_kmers < 0 || n_samples < 0) return fail(PSK_EINVAL, "bad argument");
    if (n_kmers && (!kmers || !kmer_off || !coefs || (n_samples && (!x || !names || !name_off)))) return fail(PSK_EINVAL, "null buffer");
    std::vector<std::string> who((size_t)n_kmers);
    std::vector<int64_t> count((size_t)n_kmers, 0);
    for (int64_t i = 0; i < n_samples; i++) {   // x is samples x k-mers, row-major: one pass in memory order
        const int64_t *row = x + (size_t)i * n_kmers;
        const char *nm = names + name_off[i];
        const size_t len = (size_t)(name_off[i + 1] - name_off[i]);
        for (int64_t j = 0; j < n_kmers; j++)
            if (row[j] != 0) {
                std::string &w = who[(size_t)j];
                if (count[(size_t)j]++) w += ' ';
                w.append(nm, len);
            }
    }
    std::string out;
    out.reserve((size_t)n_kmers * 64);
    char num[24];
    for (int64_t j = 0; j < n_kmers; j++) {
        out.append(kmers + kmer_off[j], (size_t)(kmer_off[j + 1] - kmer_off[j]));
        out += '\t';
        py_repr(coefs[j], out);
        out += '\t';
        auto r = std::to_chars(num, num + sizeof num, count[(size_t)j]);
        out.append(num, r.ptr - num);
        out += "\t| ";
        out += who[(size_t)j];
        out += '\n';
    }
    FILE *f = fopen(path, "ab");
    if (!f) return fail(PSK_EINVAL, "cannot open the coefficient file for appending");
    int bad = write_all(f, out);
    bad = fclose(f) != 0 || bad;
    return bad ? fail(PSK_EINVAL, "writing the coefficient file failed") : PSK_OK;
}
