// Host side of the cached-results layout: JSON text of fp64 tensors, written the way Python's `json.dump` writes the
// reference's `.mc` / `.mcm` caches (mcsim.py:457-459, :501: nested lists, ", " separators, `NaN` / `Infinity` tokens),
// so that the reference's own `json.load` cache-hit branches (mcsim.py:396-397, :504-506) read the files back to the
// SAME doubles.  Digits are the shortest round-trip representation (std::to_chars), formatted by all host threads:
// a paper-scale fidelity cache (4.4e6 values, ~90 MB of text) takes ~25 ms instead of the ~2 s of the Python encoder,
// which would otherwise dominate a cold `get_metrics_dict` once the arithmetic runs on the GPU.
//
// No GPU code here: plain C++17, part of librobchar_hip.so because the cache files are part of the drop-in boundary.
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/robchar_hip.h"

namespace {

constexpr int kMaxTok = 26;   // longest token: "-2.2250738585072014e-308" = 24 chars, + ", "

inline char* put_value(char* p, double v) {
    if (v != v) {
        memcpy(p, "NaN", 3);
        return p + 3;
    }
    if (std::isinf(v)) {
        if (v < 0) *p++ = '-';
        memcpy(p, "Infinity", 8);
        return p + 8;
    }
    // `float.__repr__` (what json.dump writes): the shortest round-trip DIGITS, laid out in positional notation while the
    // decimal point sits at -4 < decpt <= 16 and in exponent notation ("1e-05", "1.5e+16": at least two exponent digits)
    // outside.  std::to_chars(shortest) produces the same digits but picks whichever of the two notations is shorter
    // ("1e-04" for 0.0001, "12345678901234567000" beyond 1e16), so only its positional answers with at most 16 integer
    // digits are taken as they are; everything else is re-laid from the scientific form by Python's rule.
    char* const first = p;
    char* const end = std::to_chars(p, p + 25, v).ptr;
    const char* q = first + (*first == '-');
    int intdigits = 0;
    while (q < end && *q >= '0' && *q <= '9') ++q, ++intdigits;
    if (q == end) {                           // positional, integral: Python appends ".0" (and json.load keeps a float)
        if (intdigits <= 16) {
            p = end;
            *p++ = '.';
            *p++ = '0';
            return p;
        }
    } else if (*q == '.') {
        bool has_e = false;
        for (const char* r = q + 1; r < end; ++r) has_e |= (*r == 'e');
        if (!has_e && intdigits <= 16) return end;
    }
    char sci[32];                             // "-d.ddddddddddddddddde-308"
    char* const se = std::to_chars(sci, sci + 32, v, std::chars_format::scientific).ptr;
    const char* s = sci;
    p = first;
    if (*s == '-') *p++ = *s++;
    char dig[20];
    int nd = 0;
    dig[nd++] = *s++;
    if (*s == '.') {
        ++s;
        while (*s != 'e') dig[nd++] = *s++;
    }
    const char* const epos = s;               // at 'e'
    int ex = 0;
    for (const char* r = epos + 2; r < se; ++r) ex = ex * 10 + (*r - '0');
    if (epos[1] == '-') ex = -ex;
    const int decpt = ex + 1;
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) {
            *p++ = '0';
            *p++ = '.';
            for (int i = 0; i < -decpt; ++i) *p++ = '0';
            memcpy(p, dig, nd);
            p += nd;
        } else if (decpt >= nd) {
            memcpy(p, dig, nd);
            p += nd;
            for (int i = nd; i < decpt; ++i) *p++ = '0';
            *p++ = '.';
            *p++ = '0';
        } else {
            memcpy(p, dig, decpt);
            p += decpt;
            *p++ = '.';
            memcpy(p, dig + decpt, nd - decpt);
            p += nd - decpt;
        }
        return p;
    }
    const size_t n = (size_t)(se - (sci + (sci[0] == '-')));      // exponent notation: to_chars' own text is Python's
    memcpy(p, sci + (sci[0] == '-'), n);
    return p + n;
}

struct Shape {
    int ndim;
    long long rows, K;                  // rows = prod(shape[:-1]) (1 for a vector), K = shape[-1]
    std::vector<long long> period;      // for each outer dim d (0 .. ndim-2): rows per element of that dim
};

bool make_shape(int ndim, const long long* shape, Shape* s) {
    if (ndim < 1 || ndim > 8 || !shape) return false;
    s->ndim = ndim;
    s->K = shape[ndim - 1];
    s->rows = 1;
    for (int d = 0; d < ndim; ++d)
        if (shape[d] < 0) return false;
    s->period.assign(ndim > 1 ? ndim - 1 : 0, 1);
    // period[d] = number of rows spanned by ONE index of outer dim d = prod(shape[d+1 .. ndim-2])
    for (int d = ndim - 2; d >= 0; --d) {
        s->period[d] = s->rows;
        s->rows *= shape[d];
    }
    return true;
}

// text of rows [r0, r1): every row carries the brackets that open / close around it and its separator
size_t encode_rows(const double* data, const Shape& s, const long long* shape, long long r0, long long r1, char* out) {
    char* p = out;
    for (long long r = r0; r < r1; ++r) {
        for (int d = 0; d < s.ndim - 1; ++d)                          // outer lists that START at this row
            if (r % (s.period[d] * shape[d]) == 0 && d > 0) *p++ = '[';
        // (dim 0's bracket is written once by the caller; dims 1.. open whenever their parent index advances)
        *p++ = '[';
        const double* row = data + r * s.K;
        for (long long k = 0; k < s.K; ++k) {
            p = put_value(p, row[k]);
            if (k + 1 < s.K) {
                *p++ = ',';
                *p++ = ' ';
            }
        }
        *p++ = ']';
        for (int d = s.ndim - 2; d >= 1; --d)                         // outer lists that END after this row
            if ((r + 1) % (s.period[d - 1]) == 0) *p++ = ']';
        if (r + 1 < s.rows) {
            *p++ = ',';
            *p++ = ' ';
        }
    }
    return (size_t)(p - out);
}

// Formats rows [0, rows) in rounds of T blocks (one block per thread, ~64k values each, into per-thread buffers that
// persist across calls: no per-call page faults) and hands the finished blocks to `sink` in order.
// Where the text goes.  put(): one piece, in order.  put_many(): the finished blocks of a round, in order - copied by
// as many threads as there are blocks (memcpy into the caller's buffer / pwrite at the blocks' file offsets: the
// page-cache copy of a 100 MB cache file is otherwise the slowest part of writing it).
struct MemSink {
    char* p;
    bool put(const char* src, size_t n) {
        memcpy(p, src, n);
        p += n;
        return true;
    }
    bool put_many(char* const* ptr, const size_t* len, int nb) {
        std::vector<std::thread> th;
        char* q = p;
        for (int t = 0; t < nb; ++t) {
            if (t + 1 < nb && len[t] > (1u << 16))
                th.emplace_back([=]() { memcpy(q, ptr[t], len[t]); });
            else
                memcpy(q, ptr[t], len[t]);
            q += len[t];
        }
        for (auto& x : th) x.join();
        p = q;
        return true;
    }
};

struct FdSink {
    int fd;
    static bool write_all(int fd, const char* src, size_t n, off_t at, bool positioned) {
        while (n > 0) {
            const ssize_t w = positioned ? ::pwrite(fd, src, n, at) : ::write(fd, src, n);
            if (w < 0) {
                if (errno == EINTR) continue;
                return false;
            }
            src += w;
            n -= (size_t)w;
            at += w;
        }
        return true;
    }
    bool put(const char* src, size_t n) { return write_all(fd, src, n, 0, false); }
    bool put_many(char* const* ptr, const size_t* len, int nb) {
        const off_t base = ::lseek(fd, 0, SEEK_CUR);
        if (base < 0 || nb == 1) {                       // not seekable (a pipe): plain ordered writes
            for (int t = 0; t < nb; ++t)
                if (!put(ptr[t], len[t])) return false;
            return true;
        }
        std::vector<off_t> at(nb + 1, base);
        for (int t = 0; t < nb; ++t) at[t + 1] = at[t] + (off_t)len[t];
        std::vector<char> ok(nb, 1);
        std::vector<std::thread> th;
        for (int t = 1; t < nb; ++t)
            th.emplace_back([&, t]() { ok[t] = write_all(fd, ptr[t], len[t], at[t], true) ? 1 : 0; });
        ok[0] = write_all(fd, ptr[0], len[0], at[0], true) ? 1 : 0;
        for (auto& x : th) x.join();
        for (int t = 0; t < nb; ++t)
            if (!ok[t]) return false;
        return ::lseek(fd, at[nb], SEEK_SET) == at[nb];
    }
};

template <typename Sink>
long long encode_blocks(const double* data, const Shape& s, const long long* shape, int nthreads, Sink& sink) {
    int T = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    if (T > 64) T = 64;
    const long long n = s.rows * s.K;
    if (n < (1 << 15)) T = 1;
    long long block_rows = s.K > 0 ? (65536 / s.K) : 65536;          // at most ~64k values per block (buffer size) ...
    if (block_rows > (s.rows + T - 1) / T) block_rows = (s.rows + T - 1) / T;   // ... and at least T blocks: every thread busy
    if (block_rows < 1) block_rows = 1;
    const long long nblocks = (s.rows + block_rows - 1) / block_rows;
    if ((long long)T > nblocks) T = (int)nblocks;
    const size_t cap = (size_t)(block_rows * (s.K * kMaxTok + 2LL * s.ndim + 4));
    static thread_local std::vector<std::unique_ptr<char[]>> bufs;       // owned by the calling thread, reused
    static thread_local std::vector<size_t> caps;
    if ((int)bufs.size() < T) {
        bufs.resize(T);
        caps.resize(T, 0);
    }
    for (int t = 0; t < T; ++t)
        if (caps[t] < cap) {
            bufs[t].reset(new char[cap]);                                // uninitialised: only written pages are touched
            caps[t] = cap;
        }
    char* ptr[64];                               // plain pointers: a thread_local is per thread, the workers must not name it
    for (int t = 0; t < T; ++t) ptr[t] = bufs[t].get();
    long long total = 0;
    std::vector<size_t> len(T);
    for (long long b0 = 0; b0 < nblocks; b0 += T) {
        const int nb = (int)std::min<long long>(T, nblocks - b0);
        auto work = [&](int t) {
            const long long r0 = (b0 + t) * block_rows, r1 = std::min(s.rows, r0 + block_rows);
            len[t] = encode_rows(data, s, shape, r0, r1, ptr[t]);
        };
        if (nb == 1) {
            work(0);
        } else {
            std::vector<std::thread> th;
            for (int t = 1; t < nb; ++t) th.emplace_back(work, t);
            work(0);
            for (auto& x : th) x.join();
        }
        if (!sink.put_many(ptr, len.data(), nb)) return RC_EINVAL;       // the round's blocks, in order (copied in parallel)
        for (int t = 0; t < nb; ++t) total += (long long)len[t];
    }
    return total;
}

// the whole array: outer bracket, degenerate shapes, then the row blocks
template <typename Sink>
long long encode_array(const double* data, int ndim, const long long* shape, int nthreads, Sink& sink) {
    Shape s;
    if (!make_shape(ndim, shape, &s)) return RC_EINVAL;
    if (s.rows * s.K > 0 && !data) return RC_EINVAL;
    if (ndim == 1) {
        // a plain vector: one "row", no outer list
        return encode_blocks(data, s, shape, 1, sink);
    }
    for (int d = 0; d < ndim - 1; ++d) {
        if (shape[d] == 0) {                      // some outer dim is empty: e.g. shape (3, 0, 5) -> [[], [], []]
            std::string txt;
            struct R {
                static void go(std::string& t, int dim, int stop, const long long* sh) {
                    t += '[';
                    if (dim < stop)
                        for (long long i = 0; i < sh[dim]; ++i) {
                            go(t, dim + 1, stop, sh);
                            if (i + 1 < sh[dim]) t += ", ";
                        }
                    t += ']';
                }
            };
            R::go(txt, 0, d, shape);
            return sink.put(txt.data(), txt.size()) ? (long long)txt.size() : (long long)RC_EINVAL;
        }
    }
    if (!sink.put("[", 1)) return RC_EINVAL;
    const long long body = encode_blocks(data, s, shape, nthreads, sink);
    if (body < 0) return body;
    if (!sink.put("]", 1)) return RC_EINVAL;
    return body + 2;
}

}  // namespace

extern "C" {

long long rc_json_bound_f64(int ndim, const long long* shape) {
    Shape s;
    if (!make_shape(ndim, shape, &s)) return RC_EINVAL;
    long long n = 1;
    for (int d = 0; d < ndim; ++d) n *= shape[d];
    // values + per-row brackets / separators + a generous constant for degenerate shapes
    return n * kMaxTok + (s.rows + 1) * (2LL * ndim + 4) + 64;
}

long long rc_json_encode_f64(const double* data, int ndim, const long long* shape, char* out, long long cap,
                             int nthreads) {
    if (!out || cap < rc_json_bound_f64(ndim, shape)) return RC_EINVAL;
    MemSink sink{out};
    return encode_array(data, ndim, shape, nthreads, sink);
}

long long rc_json_write_f64(int fd, const double* data, int ndim, const long long* shape, int nthreads) {
    if (fd < 0) return RC_EINVAL;
    FdSink sink{fd};
    return encode_array(data, ndim, shape, nthreads, sink);
}

}  // extern "C"
