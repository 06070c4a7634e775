// dph_host.cpp -- the host half of MIPS.search_phrase in C++: what the reference does in python
// loops over 2*B*k candidates after the kernels (index.py:373-421) -- interleave start / end candidates, metadata
// look-up, dict assembly, answer slice, paragraph cropping (`adjust`, :167-176), sentence cropping (`adjust_sent`,
// :178-187, with the rule-based sentencizer), per-query sort and the dummy filter -- one call per batch.
// Strings stay python objects (the result dicts are what callers consume); all positions are CODE POINTS, so the string
// work goes through the PyUnicode API (find / substring on the str object itself, no UTF-8 round trip).
// Built in-tree as densephrases_amd/_dph_host*.so (densephrases_amd/build.py); no GPU code here.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace py = pybind11;

namespace {

struct DocView {
    PyObject* title;      // str (borrowed from the DocMeta kept alive in `keep`)
    PyObject* context;    // str
    const int64_t* f2o; Py_ssize_t n_f2o;
    const int32_t* w2cs; Py_ssize_t n_w2cs;
    const int32_t* w2ce; Py_ssize_t n_w2ce;
};

PyObject* g_delim = nullptr;   // " [PAR] "
PyObject *k_context, *k_title, *k_doc_idx, *k_start_pos, *k_end_pos, *k_start_idx, *k_end_idx, *k_score, *k_start_vec,
    *k_end_vec, *k_answer, *k_dummy;

void init_keys() {
    if (g_delim) return;
    g_delim = PyUnicode_InternFromString(" [PAR] ");
    k_context = PyUnicode_InternFromString("context");
    k_title = PyUnicode_InternFromString("title");
    k_doc_idx = PyUnicode_InternFromString("doc_idx");
    k_start_pos = PyUnicode_InternFromString("start_pos");
    k_end_pos = PyUnicode_InternFromString("end_pos");
    k_start_idx = PyUnicode_InternFromString("start_idx");
    k_end_idx = PyUnicode_InternFromString("end_idx");
    k_score = PyUnicode_InternFromString("score");
    k_start_vec = PyUnicode_InternFromString("start_vec");
    k_end_vec = PyUnicode_InternFromString("end_vec");
    k_answer = PyUnicode_InternFromString("answer");
    k_dummy = PyUnicode_InternFromString("dummy");
}

PyObject* result_template() {                         // {key: None} for the eleven keys of a result dict
    static PyObject* t = nullptr;
    if (!t) {
        t = _PyDict_NewPresized(11);
        if (!t) throw py::error_already_set();
        for (PyObject* k : {k_context, k_title, k_doc_idx, k_start_pos, k_end_pos, k_start_idx, k_end_idx, k_score, k_start_vec, k_end_vec, k_answer})
            if (PyDict_SetItem(t, k, Py_None) < 0) throw py::error_already_set();
    }
    return t;
}

inline void set_steal(PyObject* d, PyObject* key, PyObject* val) {      // dict[key] = val, consuming the reference to val
    if (!val) throw py::error_already_set();
    if (PyDict_SetItem(d, key, val) < 0) { Py_DECREF(val); throw py::error_already_set(); }
    Py_DECREF(val);
}

// ---------------------------------------------------------------------------------------------- sentence units
// [(sentence_start, sentence_end)) in code points: the rule of densephrases_amd/sentencizer.py (documented there): a
// restatement from memory of the part of spaCy 2.3's English tokenizer and rule-based `sentencizer` (index.py:65-66,
// 178-187) that decides sentence boundaries -- special-case abbreviations, prefix / suffix peeling with the tokenizer's
// period rule, the lower.Upper infix, ellipses as one token, and Sentencizer.predict (a punct_chars token arms the split,
// the next token that is not punctuation starts the sentence).  Approximate: spaCy itself is absent offline.
#include "dph_sentencizer_tables.inc"
#include "dph_unicode_punct.inc"

template <size_t N>
inline bool in_sorted(const Py_UCS4 (&t)[N], Py_UCS4 c) { return std::binary_search(t, t + N, c); }

inline bool is_punct_cp(Py_UCS4 c) {                      // Unicode general category P*
    size_t lo = 0, hi = sizeof(DPH_PUNCT_RANGES) / sizeof(DPH_PUNCT_RANGES[0]);
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (DPH_PUNCT_RANGES[mid][1] < c) lo = mid + 1; else hi = mid;
    }
    return lo < sizeof(DPH_PUNCT_RANGES) / sizeof(DPH_PUNCT_RANGES[0]) && DPH_PUNCT_RANGES[lo][0] <= c;
}

struct Text {
    int kind; const void* data; Py_ssize_t n;
    explicit Text(PyObject* o) : kind(PyUnicode_KIND(o)), data(PyUnicode_DATA(o)), n(PyUnicode_GET_LENGTH(o)) {}
    Text(const Text& t, Py_ssize_t a, Py_ssize_t b) : kind(t.kind), data((const char*)t.data + a * t.kind), n(b - a) {}      // t[a:b], no copy
    Py_UCS4 at(Py_ssize_t i) const { return PyUnicode_READ(kind, data, i); }
};
enum TokKind { TOK_TERM = 0, TOK_PUNCT = 1, TOK_OTHER = 2 };
struct Tok { Py_ssize_t a, b; int kind; };

const std::unordered_set<std::string>& abbreviations() {
    static const std::unordered_set<std::string> set(S_ABBREV, S_ABBREV + sizeof(S_ABBREV) / sizeof(S_ABBREV[0]));
    return set;
}
inline bool special(const Text& t, Py_ssize_t a, Py_ssize_t b) {
    if (b - a == 2 && t.at(a + 1) == '.' && t.at(a) >= 'a' && t.at(a) <= 'z') return true;      // "a." .. "z."
    if ((b - a == 5 || b - a == 6) && t.at(b - 1) == '.' && t.at(b - 2) == 'm' && t.at(b - 3) == '.' &&
        (t.at(b - 4) == 'a' || t.at(b - 4) == 'p')) {      // "1a.m." .. "12p.m." (two tokens in spaCy, neither a full stop)
        const Py_UCS4 d0 = t.at(a), d1 = b - a == 6 ? t.at(a + 1) : 0;
        if (b - a == 5) return d0 >= '1' && d0 <= '9';
        return d0 == '1' && d1 >= '0' && d1 <= '2';
    }
    if (b - a < 3 || b - a > 8) return false;
    char buf[9];
    for (Py_ssize_t i = a; i < b; ++i) { const Py_UCS4 c = t.at(i); if (c >= 128) return false; buf[i - a] = (char)c; }
    return abbreviations().count(std::string(buf, (size_t)(b - a))) != 0;
}
inline int kind_of(const Text& t, Py_ssize_t a, Py_ssize_t b) {
    if (b - a == 1 && in_sorted(S_PUNCT_CHARS, t.at(a))) return TOK_TERM;
    for (Py_ssize_t i = a; i < b; ++i) if (!is_punct_cp(t.at(i))) return TOK_OTHER;
    return TOK_PUNCT;
}
inline Py_ssize_t prefix_len(const Text& t, Py_ssize_t a, Py_ssize_t b) {
    const Py_UCS4 c = t.at(a);
    if (c == '.') {
        Py_ssize_t j = a;
        while (j < b && t.at(j) == '.') ++j;
        return j - a >= 2 ? j - a : 0;                     // ellipsis
    }
    if (in_sorted(S_PREFIX, c)) return 1;
    if (c == '+' && !(a + 1 < b && t.at(a + 1) >= '0' && t.at(a + 1) <= '9')) return 1;
    return 0;
}
inline Py_ssize_t suffix_len(const Text& t, Py_ssize_t a, Py_ssize_t b) {
    const Py_UCS4 c = t.at(b - 1);
    if (c == '.') {
        Py_ssize_t k = b;
        while (k > a && t.at(k - 1) == '.') --k;
        if (b - k >= 2) return b - k;                      // ellipsis: ONE token
        if (b - 2 >= a) {
            const Py_UCS4 p = t.at(b - 2);
            if (Py_UNICODE_ISLOWER(p) || (p >= '0' && p <= '9') || in_sorted(S_BEFORE_PERIOD, p) ||
                (b - 3 >= a && Py_UNICODE_ISUPPER(p) && Py_UNICODE_ISUPPER(t.at(b - 3))))
                return 1;
        }
        return 0;
    }
    if (in_sorted(S_SUFFIX, c)) return 1;
    if (b - a >= 2 && (c == 's' || c == 'S') && (t.at(b - 2) == '\'' || t.at(b - 2) == 0x2019)) return 2;
    return 0;
}
// tokens of the whitespace-free chunk [a, b), in text order (Tokenizer._split_affixes, then the [lower|quote].[Upper|quote] infix)
void tokens_of_chunk(const Text& t, Py_ssize_t a, Py_ssize_t b, std::vector<Tok>& out, std::vector<Tok>& tail) {
    tail.clear();
    while (a < b && !special(t, a, b)) {
        const Py_ssize_t pre = prefix_len(t, a, b);
        if (pre && a + pre < b && special(t, a + pre, b)) { out.push_back({a, a + pre, kind_of(t, a, a + pre)}); a += pre; break; }
        const Py_ssize_t suf = suffix_len(t, a, b);
        if (suf && a < b - suf && special(t, a, b - suf)) { tail.push_back({b - suf, b, kind_of(t, b - suf, b)}); b -= suf; break; }
        if (pre && suf && pre + suf <= b - a) {
            out.push_back({a, a + pre, kind_of(t, a, a + pre)}); a += pre;
            tail.push_back({b - suf, b, kind_of(t, b - suf, b)}); b -= suf;
        } else if (pre) {
            out.push_back({a, a + pre, kind_of(t, a, a + pre)}); a += pre;
        } else if (suf) {
            tail.push_back({b - suf, b, kind_of(t, b - suf, b)}); b -= suf;
        } else {
            break;
        }
    }
    if (a < b) {
        if (special(t, a, b)) {
            out.push_back({a, b, TOK_OTHER});
        } else {
            // infixes (Tokenizer._attach_tokens): an ellipsis ("..+" or U+2026), or a '.' between [lower-case or quote] and
            // [upper-case or quote]; a match at the start of the piece being collected is skipped, as spaCy does
            Py_ssize_t s0 = a, i = a;
            while (i < b) {
                Py_ssize_t n = 0;
                if (t.at(i) == '.') {
                    Py_ssize_t j = i;
                    while (j < b && t.at(j) == '.') ++j;
                    if (j - i >= 2) n = j - i;
                    else if (i > a && i + 1 < b && (Py_UNICODE_ISLOWER(t.at(i - 1)) || in_sorted(S_QUOTES, t.at(i - 1))) &&
                             (Py_UNICODE_ISUPPER(t.at(i + 1)) || in_sorted(S_QUOTES, t.at(i + 1))))
                        n = 1;
                } else if (t.at(i) == 0x2026) {
                    n = 1;
                }
                if (n == 0) { ++i; continue; }
                if (i > s0) {
                    out.push_back({s0, i, kind_of(t, s0, i)});
                    out.push_back({i, i + n, kind_of(t, i, i + n)});
                    s0 = i + n;
                }
                i += n;
            }
            if (s0 < b) out.push_back({s0, b, kind_of(t, s0, b)});
        }
    }
    out.insert(out.end(), tail.rbegin(), tail.rend());
}
void split_sentences(const Text& t, std::vector<std::pair<Py_ssize_t, Py_ssize_t>>& out, std::vector<Tok>& toks, std::vector<Tok>& tail) {
    out.clear();
    toks.clear();
    // whitespace (Tokenizer.__call__): ONE blank after a token belongs to that token; the rest of a run of whitespace -- and a
    // run at the very start -- is a token of its own (not punctuation, not a full stop: it can start a sentence)
    for (Py_ssize_t i = 0; i < t.n;) {
        Py_ssize_t j = i;
        if (Py_UNICODE_ISSPACE(t.at(i))) {
            while (j < t.n && Py_UNICODE_ISSPACE(t.at(j))) ++j;
            const Py_ssize_t a = (i > 0 && t.at(i) == ' ') ? i + 1 : i;
            if (a < j) toks.push_back({a, j, TOK_OTHER});
        } else {
            while (j < t.n && !Py_UNICODE_ISSPACE(t.at(j))) ++j;
            tokens_of_chunk(t, i, j, toks, tail);
        }
        i = j;
    }
    if (toks.empty()) return;
    bool seen = false;
    Py_ssize_t first = toks[0].a;
    for (size_t k = 0; k < toks.size(); ++k) {
        if (seen && toks[k].kind == TOK_OTHER) {
            out.emplace_back(first, toks[k - 1].b);
            first = toks[k].a;
            seen = false;
        } else if (toks[k].kind == TOK_TERM) {
            seen = true;
        }
    }
    out.emplace_back(first, toks.back().b);
}
void split_sentences(PyObject* text, std::vector<std::pair<Py_ssize_t, Py_ssize_t>>& out) {
    std::vector<Tok> toks, tail;
    split_sentences(Text(text), out, toks, tail);
}

// ' [PAR] ' in t[a:b) -- the first match from the left (dir > 0) or the last one (dir < 0) lying wholly inside, -1 when there is
// none: PyUnicode_Find(context, ' [PAR] ', a, b, dir) without the interpreter (the candidate loop runs with the GIL released)
inline Py_ssize_t find_par(const Text& t, Py_ssize_t a, Py_ssize_t b, int dir) {
    static const Py_UCS4 pat[7] = {' ', '[', 'P', 'A', 'R', ']', ' '};
    if (a < 0) a = 0;
    if (b > t.n) b = t.n;
    auto at = [&](Py_ssize_t i) { for (int j = 0; j < 7; ++j) if (t.at(i + j) != pat[j]) return false; return true; };
    if (dir > 0) { for (Py_ssize_t i = a; i + 7 <= b; ++i) if (t.at(i + 1) == '[' && at(i)) return i; }
    else { for (Py_ssize_t i = b - 7; i >= a; --i) if (t.at(i + 1) == '[' && at(i)) return i; }
    return -1;
}

// A few worker threads for the part of the host half that needs no interpreter (positions, paragraph / sentence bounds, per-query
// sort and de-duplication).  run(n, f): f(chunk) for chunk in [0, n), the caller working too; DPH_HOST_THREADS sets the number of
// threads (default min(8, cores / 2); 1 = everything on the caller).
class Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, done;
    const std::function<void(int)>* job = nullptr;
    std::atomic<int> next{0};
    int n_chunks = 0, gen = 0, busy = 0;
    bool stop = false;
    void work() { for (int c; (c = next.fetch_add(1)) < n_chunks;) (*job)(c); }
    void loop() {
        int seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
            }
            work();
            { std::lock_guard<std::mutex> l(m); if (--busy == 0) done.notify_all(); }
        }
    }
public:
    static int wanted() {
        if (const char* e = getenv("DPH_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) return v > 64 ? 64 : v; }
        unsigned hc = std::thread::hardware_concurrency();
        // (a pod's cgroup may allow fewer CPUs than it shows: cpu.max = "<quota> <period>"; more busy threads than that are throttled)
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long q = 0, p_ = 0;
            if (fscanf(f, "%lld %lld", &q, &p_) == 2 && q > 0 && p_ > 0) hc = std::min<unsigned>(hc, (unsigned)((q + p_ - 1) / p_) * 2u);
            fclose(f);
        }
        return (int)std::max(1u, std::min(8u, hc / 2));
    }
    int threads() const { return (int)th.size() + 1; }
    void run(int n, const std::function<void(int)>& f) {
        const int want = wanted();
        while ((int)th.size() + 1 < want) th.emplace_back([this] { loop(); });
        if (n <= 1 || th.empty()) { for (int c = 0; c < n; ++c) f(c); return; }
        {
            std::lock_guard<std::mutex> l(m);
            job = &f; n_chunks = n; next = 0; busy = (int)th.size(); ++gen;
        }
        cv.notify_all();
        work();
        std::unique_lock<std::mutex> l(m);
        done.wait(l, [&] { return busy == 0; });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

}  // namespace

// One per MIPS instance: keeps the per-document views (title / context objects, f2o and word2char arrays) of the documents
// it has seen, so a document costs one python call the first time it appears, not once per batch (the reference's RAM
// branch keeps whole metadata dicts around the same way, index.py:106-122; SURVEY 8(f) rank 4 asks for the doc cache).
struct Prepared;
py::list aggregate(py::list results, const std::string& strat, py::object normalize);

// where a candidate's answer, paragraph and sentences lie (phase 1 below)
struct Cand {
    int state = 0;                                   // 0 dropped, 1 alive, 2 / 3 start / end index outside the document
    Py_ssize_t a0 = 0, a1 = 0;                       // answer = context[a0:a1]
    Py_ssize_t lo = 0, hi = 0;                       // paragraph = context[lo:hi]
    Py_ssize_t start_pos = 0, end_pos = 0;           // relative to the returned context
    std::vector<std::pair<Py_ssize_t, Py_ssize_t>> parts;     // return_sent: the sentences of the paragraph that are joined (paragraph coordinates)
    bool joined = false;
};

struct __attribute__((visibility("hidden"))) HostHalf {
    // A cached document: the view plus the owners of everything it points into.  Shared: a batch in flight keeps its documents alive
    // whatever later batches rotate out of the cache.  (An Entry dies under the GIL: its members are python references.)
    struct __attribute__((visibility("hidden"))) Entry {
        DocView v;
        py::object title, context, f2o, ws, we;
    };
    using EntryP = std::shared_ptr<Entry>;
    using Map = std::unordered_map<int64_t, EntryP>;
    py::function doc_meta;
    size_t cap;                                          // documents kept, both generations together
    // Two generations: look-ups hit `docs` (the current one) or `old` (an entry found there moves over); when the current
    // one is full it becomes the old one and the previous old one is dropped.  A working set that fits the capacity is never
    // thrown away wholesale (round 3 cleared everything when the map was full: two alternating batches of 20 k documents
    // against a capacity of 32 k re-fetched every document through python on every batch, 130-320 ms per batch of 512).
    Map docs, old;
    size_t fetched = 0;                                  // python doc_meta calls so far (statistics)

    HostHalf(py::function f, size_t cache_docs) : doc_meta(std::move(f)), cap(cache_docs < 2 ? 2 : cache_docs) {}

    EntryP fetch(int64_t d) {                            // (GIL held)
        py::object m = doc_meta(d);
        ++fetched;
        EntryP e = std::make_shared<Entry>();
        e->title = m.attr("title");
        e->context = m.attr("context");
        auto f2o = py::array_t<int64_t, py::array::c_style | py::array::forcecast>(m.attr("f2o_start"));
        auto ws = py::array_t<int32_t, py::array::c_style | py::array::forcecast>(m.attr("word2char_start"));
        auto we = py::array_t<int32_t, py::array::c_style | py::array::forcecast>(m.attr("word2char_end"));
        if (!PyUnicode_Check(e->title.ptr()) || !PyUnicode_Check(e->context.ptr())) throw std::invalid_argument("assemble: title / context must be str");
        e->v = DocView{e->title.ptr(), e->context.ptr(), f2o.data(), f2o.shape(0), ws.data(), ws.shape(0), we.data(), we.shape(0)};
        e->f2o = std::move(f2o); e->ws = std::move(ws); e->we = std::move(we);
        return e;
    }

    // Make every document of `ids` (>= 0) resident in the CURRENT generation and hand a reference to each distinct one to `keep`.
    // Runs on the caller's thread or on a batch's worker thread: the maps are touched by one thread at a time (the callers make
    // sure), the GIL is taken only for what needs it -- the python doc_meta callback of a document seen for the first time, and a
    // rotation (dropping a generation may destroy entries, i.e. python references).
    void make_resident(const int64_t* ids, Py_ssize_t n, std::vector<EntryP>& keep, bool have_gil) {
        std::unordered_set<int64_t> need;                  // distinct documents of the batch not in the current generation
        for (Py_ssize_t g = 0; g < n; ++g) if (ids[g] >= 0 && !docs.count(ids[g])) need.insert(ids[g]);
        if (!need.empty()) {
            bool to_fetch = false;
            for (int64_t d : need) if (!old.count(d)) { to_fetch = true; break; }
            if (2 * need.size() > cap) cap = 2 * need.size();  // one batch must fit a generation
            const bool rotate = docs.size() + need.size() > cap / 2;
            std::unique_ptr<py::gil_scoped_acquire> gil;
            if (!have_gil && (to_fetch || rotate)) gil.reset(new py::gil_scoped_acquire());
            if (rotate) {
                // the documents of THIS batch that sit in the generation about to be dropped are carried over first ...
                Map next;
                for (Py_ssize_t g = 0; g < n; ++g) {
                    if (ids[g] < 0) continue;
                    auto it = docs.find(ids[g]);
                    if (it != docs.end()) { next.emplace(it->first, std::move(it->second)); docs.erase(it); }
                }
                // ... and so are the ones waiting in the OLD generation, which the rotation is about to throw away (looking them up
                // after it could never hit: they would come back through the python doc_meta callback every rotating batch)
                for (int64_t d : need) {
                    auto it = old.find(d);
                    if (it != old.end()) { next.emplace(d, std::move(it->second)); old.erase(it); }
                }
                old = std::move(docs);
                docs = std::move(next);
            }
            for (int64_t d : need) {
                if (docs.count(d)) continue;
                auto it = old.find(d);
                if (it != old.end()) { docs.emplace(d, std::move(it->second)); old.erase(it); }
                else docs.emplace(d, fetch(d));
            }
        }
        std::unordered_set<int64_t> seen;
        for (Py_ssize_t g = 0; g < n; ++g) if (ids[g] >= 0 && seen.insert(ids[g]).second) keep.push_back(docs.at(ids[g]));
    }

    void resident_and_views(Prepared& P, bool have_gil);
    static void phase1(Prepared& P);
    py::list materialize(Prepared& P, py::object start_vecs, py::object end_vecs, py::object normalize);
    py::list assemble(int num_queries, int top_k, py::array_t<int64_t, py::array::c_style | py::array::forcecast> doc_i,
                      py::array_t<int64_t, py::array::c_style | py::array::forcecast> start_i,
                      py::array_t<int64_t, py::array::c_style | py::array::forcecast> end_i,
                      py::array_t<double, py::array::c_style | py::array::forcecast> score_i, py::object start_vecs,
                      py::object end_vecs, bool return_sent, py::object agg_strat, py::object normalize);
    std::shared_ptr<Prepared> prepare_async(int num_queries, int top_k, uintptr_t I_ptr, uintptr_t best_ptr, uintptr_t pred_ptr, uintptr_t status_ptr,
                                            uintptr_t id2docword_fn, uintptr_t handle, bool return_sent, py::object agg_strat);
};

// A batch between the kernels and its result dicts: the interleaved candidates, where every candidate's answer / paragraph /
// sentences lie, and which candidates survive the per-query sort and de-duplication, in result order -- everything that needs no
// interpreter.  `assemble` builds one on the calling thread; `prepare_async` builds it on a thread of its own (from the record the
// GPU half left in pinned host memory) while the caller turns the PREVIOUS batch into python objects.
struct __attribute__((visibility("hidden"))) Prepared {
    HostHalf* host = nullptr;
    int num_queries = 0, top_k = 0, mode = 0;          // mode: 0 no aggregation, 1 .. 4 = opt1 .. opt4
    bool return_sent = false;
    std::vector<int64_t> D, S, E;                      // [2 * B * k] interleaved (start-candidate, end-candidate)
    std::vector<double> SC;
    std::vector<HostHalf::EntryP> keep;                // the batch's documents (alive until this object dies, under the GIL)
    std::vector<const DocView*> views;                 // one per candidate
    std::vector<Cand> cand;
    std::vector<std::vector<Py_ssize_t>> order;        // per query: the candidates that become results, in result order
    double num_docs = 0.0;                             // mean distinct documents per query (index.py:213-215)
    bool needs_exact = false;                          // a row came back uncertified: the caller repairs the batch the synchronous way
    int error_kind = 0;                                // 1 out_of_range, 2 runtime
    std::string error;
    std::thread worker;
    ~Prepared() {
        if (worker.joinable()) { py::gil_scoped_release nogil; worker.join(); }     // (the worker may want the GIL for a document it has not seen)
    }
};

static int mode_of(py::object agg_strat) {
    if (agg_strat.is_none()) return 0;
    const std::string st = agg_strat.cast<std::string>();
    const int m_ = st == "opt1" ? 1 : st == "opt2" ? 2 : st == "opt3" ? 3 : st == "opt4" ? 4 : -1;
    if (m_ < 0) throw py::type_error("wrong aggregation strategy");
    return m_;
}

void HostHalf::resident_and_views(Prepared& P, bool have_gil) {
    const Py_ssize_t n = (Py_ssize_t)P.D.size();
    // every distinct document of the batch is made resident before views are handed out
    make_resident(P.D.data(), n, P.keep, have_gil);
    P.views.assign((size_t)n, nullptr);                // one hash look-up per run of equal documents
    int64_t last = -1;
    const DocView* lv = nullptr;
    for (Py_ssize_t g = 0; g < n; ++g) {
        if (P.D[g] < 0) continue;
        if (P.D[g] != last) { last = P.D[g]; lv = &docs.at(last)->v; }
        P.views[(size_t)g] = lv;
    }
}

// ---- phase 1, WITHOUT the interpreter (a few threads when the batch is large): where every candidate's answer, paragraph and
//      sentences lie -- the metadata look-ups are cache misses into 10^4 documents, the ' [PAR] ' searches and the sentence rule walk
//      text; then per query the sort by score and, for the strategies whose key needs no python (opt1 / opt2 / opt3),
//      MIPS.aggregate_results' de-duplication.  Only the SURVIVORS become python objects in phase 2: under opt1 about half of the
//      2 * top_k candidates of a query are the same span found twice (start- and end-candidate).
void HostHalf::phase1(Prepared& P) {
    const Py_ssize_t n = (Py_ssize_t)P.D.size();
    const int64_t *D = P.D.data(), *S = P.S.data(), *E = P.E.data();
    const double* SC = P.SC.data();
    const int num_queries = P.num_queries, mode = P.mode;
    const bool return_sent = P.return_sent;
    const bool native_agg = mode >= 1 && mode <= 3;
    const Py_ssize_t per = 2 * (Py_ssize_t)P.top_k;
    std::vector<Cand>& cand = P.cand;
    std::vector<const DocView*>& views = P.views;
    std::vector<std::vector<Py_ssize_t>>& order = P.order;
    cand.assign((size_t)n, Cand());
    order.assign((size_t)num_queries, std::vector<Py_ssize_t>());
    {
        static Pool* pool = nullptr;
        static long pool_pid = 0;
        static std::mutex pool_m;                          // (two batches' phases never overlap by design; the lock says so)
        std::lock_guard<std::mutex> pool_lock(pool_m);
        if (!pool || pool_pid != (long)getpid()) { pool = new Pool(); pool_pid = (long)getpid(); }      // (a forked child starts its own threads)
        const int chunk_q = std::max(1, 2048 / (int)std::max<Py_ssize_t>(per, 1));        // queries per chunk: ~2048 candidates
        const int n_chunks = (num_queries + chunk_q - 1) / chunk_q;
        auto body = [&](int ch) {
            std::vector<std::pair<Py_ssize_t, Py_ssize_t>> sents;
            std::vector<Tok> toks, tail;
            for (Py_ssize_t g = (Py_ssize_t)ch * chunk_q * per; g < std::min<Py_ssize_t>(n, (Py_ssize_t)(ch + 1) * chunk_q * per); ++g) {
                Cand& c = cand[(size_t)g];
                const double sc = SC[g];
                if (D[g] < 0 || !(sc > -1e5)) continue;         // dummy (index.py:400-401) or masked out: dropped at :420 anyway
                const DocView& m = *views[(size_t)g];
                const int64_t s = S[g], e = E[g];
                if (s < 0 || s >= m.n_f2o || m.f2o[s] < 0 || m.f2o[s] >= m.n_w2cs) { c.state = 2; continue; }
                Py_ssize_t start_pos = m.w2cs[m.f2o[s]], end_pos;
                if (m.n_w2ce > 0 && e >= 0) {
                    if (e >= m.n_f2o || m.f2o[e] < 0 || m.f2o[e] >= m.n_w2ce) { c.state = 3; continue; }
                    end_pos = m.w2ce[m.f2o[e]];
                } else {
                    end_pos = start_pos + 1;
                }
                const Text ctx_all(m.context);
                const Py_ssize_t clen = ctx_all.n;
                // answer = context[start_pos:end_pos] (python slice semantics)                          index.py:406-407
                c.a0 = std::min(std::max<Py_ssize_t>(start_pos, 0), clen);
                c.a1 = std::max(std::min(end_pos, clen), c.a0);
                // adjust: crop to the ' [PAR] '-delimited paragraph around the span                      index.py:167-176
                Py_ssize_t lo = find_par(ctx_all, 0, std::max<Py_ssize_t>(std::min(start_pos, clen), 0), -1);
                lo = lo == -1 ? 0 : lo + 7;
                Py_ssize_t hi = find_par(ctx_all, std::min(std::max<Py_ssize_t>(end_pos, 0), clen), clen, 1);
                hi = hi == -1 ? clen : hi;
                hi = std::max(hi, lo);
                c.lo = lo; c.hi = hi;
                start_pos -= lo;
                end_pos -= lo;
                if (return_sent) {                               // adjust_sent                             index.py:178-187
                    split_sentences(Text(ctx_all, lo, hi), sents, toks, tail);
                    if (!sents.empty()) {
                        Py_ssize_t a = -1, b = -1;
                        for (Py_ssize_t i = 0; i < (Py_ssize_t)sents.size(); ++i) {
                            if (sents[(size_t)i].first <= start_pos) a = i;
                            if (sents[(size_t)i].first <= end_pos - 1) b = i;
                        }
                        const Py_ssize_t nn = (Py_ssize_t)sents.size();
                        auto wrap = [&](Py_ssize_t v) { return v < 0 ? v + nn : v; };       // python list[-1]
                        const Py_ssize_t lo_s = std::min(a, b), hi_s = std::max(a, b);
                        for (Py_ssize_t i = lo_s; i <= hi_s; ++i) c.parts.push_back(sents[(size_t)wrap(i)]);
                        c.joined = true;
                        const Py_ssize_t base = sents[(size_t)wrap(lo_s)].first;
                        start_pos -= base;
                        end_pos -= base;
                    }
                }
                c.start_pos = start_pos; c.end_pos = end_pos;
                c.state = 1;
            }
            // per query: sorted(key=-score) (stable), then -- opt1 / opt2 / opt3 -- aggregate_results: the FIRST result with a key
            // keeps its score, later ones drop to -1e8 and are filtered out (<= -1e5) after the re-sort, which leaves the firsts in
            // the order they already have.  Keys: the reference's strings f'{title}_{start_pos}_{end_pos}' / context / f'{title}' are
            // equal exactly when (title text, start_pos, end_pos) / the context text / the title text are (title = [one str] here).
            std::u32string key;
            auto same_text = [](const Text& x, const Text& y) {
                if (x.n != y.n) return false;
                if (x.kind == y.kind) return memcmp(x.data, y.data, (size_t)x.n * (size_t)x.kind) == 0;
                for (Py_ssize_t i = 0; i < x.n; ++i) if (x.at(i) != y.at(i)) return false;
                return true;
            };
            auto make_key = [&](Py_ssize_t g) {               // the key as a string of its own (large top_k, joined sentences)
                const Cand& c = cand[(size_t)g];
                const DocView& m = *views[(size_t)g];
                key.clear();
                if (mode == 2) {
                    const Text t(m.context);
                    if (c.joined) {
                        for (size_t pi = 0; pi < c.parts.size(); ++pi) {
                            if (pi) key.push_back(U' ');
                            for (Py_ssize_t i = c.lo + c.parts[pi].first; i < c.lo + c.parts[pi].second; ++i) key.push_back((char32_t)t.at(i));
                        }
                    } else {
                        for (Py_ssize_t i = c.lo; i < c.hi; ++i) key.push_back((char32_t)t.at(i));
                    }
                } else {
                    const Text t(m.title);
                    for (Py_ssize_t i = 0; i < t.n; ++i) key.push_back((char32_t)t.at(i));
                    if (mode == 1) {
                        // (two fixed-width fields behind the title: no (title, start, end) can imitate another)
                        for (int sh = 0; sh < 64; sh += 16) key.push_back((char32_t)(((uint64_t)c.start_pos >> sh) & 0xFFFFu) + 0x110000u);
                        for (int sh = 0; sh < 64; sh += 16) key.push_back((char32_t)(((uint64_t)c.end_pos >> sh) & 0xFFFFu) + 0x110000u);
                    }
                }
            };
            for (int qi = ch * chunk_q; qi < std::min(num_queries, (ch + 1) * chunk_q); ++qi) {
                std::vector<Py_ssize_t>& o = order[(size_t)qi];
                for (Py_ssize_t g = (Py_ssize_t)qi * per; g < (Py_ssize_t)(qi + 1) * per; ++g) if (cand[(size_t)g].state == 1) o.push_back(g);
                std::stable_sort(o.begin(), o.end(), [&](Py_ssize_t x, Py_ssize_t y) { return SC[x] > SC[y]; });
                if (!native_agg) continue;
                size_t w = 0;
                if (per <= 128 && !(mode == 2 && return_sent)) {
                    // a few dozen candidates: every one against the survivors so far, cheapest field first -- no key is built
                    for (Py_ssize_t g : o) {
                        const Cand& c = cand[(size_t)g];
                        const DocView& m = *views[(size_t)g];
                        bool dup = false;
                        for (size_t j = 0; j < w && !dup; ++j) {
                            const Cand& d = cand[(size_t)o[j]];
                            const DocView& md = *views[(size_t)o[j]];
                            if (mode == 1) dup = c.start_pos == d.start_pos && c.end_pos == d.end_pos && (m.title == md.title || same_text(Text(m.title), Text(md.title)));
                            else if (mode == 3) dup = m.title == md.title || same_text(Text(m.title), Text(md.title));
                            else dup = (m.context == md.context && c.lo == d.lo && c.hi == d.hi) ||
                                       same_text(Text(Text(m.context), c.lo, c.hi), Text(Text(md.context), d.lo, d.hi));
                        }
                        if (!dup) o[w++] = g;
                    }
                } else {
                    std::unordered_set<std::u32string> seen;
                    for (Py_ssize_t g : o) {
                        make_key(g);
                        if (seen.insert(key).second) o[w++] = g;
                    }
                }
                o.resize(w);
            }
        };
        if (n >= 8192) pool->run(n_chunks, body);
        else for (int ch = 0; ch < n_chunks; ++ch) body(ch);
    }
    for (Py_ssize_t g = 0; g < n && !P.error_kind; ++g) {
        if (cand[(size_t)g].state == 2) { P.error_kind = 1; P.error = "assemble: start index outside the document"; }
        if (cand[(size_t)g].state == 3) { P.error_kind = 1; P.error = "assemble: end index outside the document"; }
    }
}

// ---- phase 2, python objects for the survivors.  Dicts, lists and strings are born below: the cyclic collector would wake up
//      every 700 allocations and, once its older generations fill, walk every cached document -- nothing created here can be part
//      of a cycle, so it rests meanwhile
py::list HostHalf::materialize(Prepared& P, py::object start_vecs, py::object end_vecs, py::object normalize) {
    init_keys();
    if (P.worker.joinable()) { py::gil_scoped_release nogil; P.worker.join(); }
    if (P.error_kind == 1) throw std::out_of_range(P.error);
    if (P.error_kind) throw std::runtime_error(P.error);
    if (P.needs_exact) throw std::runtime_error("materialize: the batch holds uncertified rows (Prepared.needs_exact): repair it the synchronous way");
    const int64_t *D = P.D.data(), *S = P.S.data(), *E = P.E.data();
    const double* SC = P.SC.data();
    const bool with_vecs = !start_vecs.is_none();
    struct GcPause {
        bool was;
        GcPause() : was(PyGC_IsEnabled() != 0) { if (was) PyGC_Disable(); }
        ~GcPause() { if (was) PyGC_Enable(); }
    } gc_pause;
    py::object sep = py::str(" ");
    // token and character positions are small: their int objects come from a table built once (CPython itself only keeps -5 .. 256)
    static PyObject* small_int[4096];
    if (!small_int[0])
        for (int i = 0; i < 4096; ++i) { small_int[i] = PyLong_FromLong(i); if (!small_int[i]) throw py::error_already_set(); }
    auto int_obj = [&](long long v) -> PyObject* {           // a NEW reference
        if (v >= 0 && v < 4096) { Py_INCREF(small_int[v]); return small_int[v]; }
        return PyLong_FromLongLong(v);
    };
    py::list out(P.num_queries);
    for (int qi = 0; qi < P.num_queries; ++qi) {
        const std::vector<Py_ssize_t>& ord = P.order[(size_t)qi];
        py::list l(ord.size());
        Py_ssize_t li = 0;
        for (Py_ssize_t g : ord) {
            const Cand& c = P.cand[(size_t)g];
            const DocView& m = *P.views[(size_t)g];
            py::object answer_o = py::reinterpret_steal<py::object>(PyUnicode_Substring(m.context, c.a0, c.a1));
            if (!answer_o) throw py::error_already_set();
            py::object ctx = py::reinterpret_steal<py::object>(PyUnicode_Substring(m.context, c.lo, c.hi));
            if (!ctx) throw py::error_already_set();
            if (c.joined) {
                py::list parts;
                for (const auto& sp : c.parts)
                    parts.append(py::reinterpret_steal<py::object>(PyUnicode_Substring(m.context, c.lo + sp.first, c.lo + sp.second)));
                ctx = py::reinterpret_steal<py::object>(PyUnicode_Join(sep.ptr(), parts.ptr()));
                if (!ctx) throw py::error_already_set();
            }
            // 11 keys: a copy of a template that holds them all (one table copy instead of eleven insertions; the values are set below)
            PyObject* rd = PyDict_Copy(result_template());
            if (!rd) throw py::error_already_set();
            PyList_SET_ITEM(l.ptr(), li++, rd);                      // (the list owns it from here: an exception below frees everything)
            if (PyDict_SetItem(rd, k_context, ctx.ptr()) < 0) throw py::error_already_set();
            PyObject* tl = PyList_New(1);
            if (!tl) throw py::error_already_set();
            Py_INCREF(m.title);
            PyList_SET_ITEM(tl, 0, m.title);
            set_steal(rd, k_title, tl);
            set_steal(rd, k_doc_idx, int_obj(D[g]));
            set_steal(rd, k_start_pos, int_obj((long long)c.start_pos));
            set_steal(rd, k_end_pos, int_obj((long long)c.end_pos));
            set_steal(rd, k_start_idx, int_obj(S[g]));
            set_steal(rd, k_end_idx, int_obj(E[g]));
            set_steal(rd, k_score, PyFloat_FromDouble(SC[g]));
            if (with_vecs) {
                py::object sv = start_vecs[py::int_(g)], ev = end_vecs[py::int_(g)];
                if (PyDict_SetItem(rd, k_start_vec, sv.ptr()) < 0 || PyDict_SetItem(rd, k_end_vec, ev.ptr()) < 0) throw py::error_already_set();
            }                                                        // (else: None, as the template has them)
            if (PyDict_SetItem(rd, k_answer, answer_o.ptr()) < 0) throw py::error_already_set();
        }
        // MIPS.search(aggregate=True) de-duplicates every query's list right away (index.py:476-480): same call, same pause.  opt4's
        // key is normalize_answer(answer) -- python -- so that strategy goes through the general routine below
        py::object ql = P.mode == 4 ? py::object(aggregate(std::move(l), "opt4", normalize)) : py::object(std::move(l));
        PyList_SET_ITEM(out.ptr(), qi, ql.release().ptr());
    }
    return out;
}

// doc_i, start_i, end_i: int64 [2*B*k] interleaved (start-candidate, end-candidate); score_i: float64 [2*B*k];
// start_vecs / end_vecs: float32 [2*B*k, 768] or None; doc_meta: callable doc_idx -> object with title, context,
// f2o_start (int64), word2char_start / word2char_end (int32).  Returns list[num_queries] of lists of dicts, each sorted
// by score (descending, stable) with the dummies (score <= -1e5) dropped.
py::list HostHalf::assemble(int num_queries, int top_k, py::array_t<int64_t, py::array::c_style | py::array::forcecast> doc_i,
                            py::array_t<int64_t, py::array::c_style | py::array::forcecast> start_i,
                            py::array_t<int64_t, py::array::c_style | py::array::forcecast> end_i,
                            py::array_t<double, py::array::c_style | py::array::forcecast> score_i, py::object start_vecs,
                            py::object end_vecs, bool return_sent, py::object agg_strat, py::object normalize) {
    init_keys();
    const Py_ssize_t n = doc_i.shape(0);
    if (start_i.shape(0) != n || end_i.shape(0) != n || score_i.shape(0) != n || n != (Py_ssize_t)num_queries * 2 * top_k)
        throw std::invalid_argument("assemble: array lengths must be 2 * num_queries * top_k");
    Prepared P;
    P.host = this; P.num_queries = num_queries; P.top_k = top_k; P.return_sent = return_sent; P.mode = mode_of(agg_strat);
    P.D.assign(doc_i.data(), doc_i.data() + n); P.S.assign(start_i.data(), start_i.data() + n);
    P.E.assign(end_i.data(), end_i.data() + n); P.SC.assign(score_i.data(), score_i.data() + n);
    resident_and_views(P, true);
    {
        py::gil_scoped_release nogil;
        phase1(P);
    }
    return materialize(P, start_vecs, end_vecs, normalize);
}

// The same from the RECORD the GPU half left in pinned host memory (densephrases_amd/dist.py RecordLayout: I i64 [2B, k], best f64
// [2B, k], pred i32 [2B, k], status i32 [2B]), on a thread of its own: ids -> (doc, word) through libdph's dph_id2docword (its address
// and the index handle come from the caller: this module does not link libdph), interleaving start / end candidates as
// MIPS.search_phrase does (index.py:373-398), the document cache, phase 1.  The caller goes on -- it materialises the PREVIOUS batch
// under the GIL meanwhile -- and collects with `materialize`.
std::shared_ptr<Prepared> HostHalf::prepare_async(int num_queries, int top_k, uintptr_t I_ptr, uintptr_t best_ptr, uintptr_t pred_ptr,
                                                  uintptr_t status_ptr, uintptr_t id2docword_fn, uintptr_t handle, bool return_sent,
                                                  py::object agg_strat) {
    init_keys();
    if (num_queries < 0 || top_k <= 0 || !I_ptr || !best_ptr || !pred_ptr || !status_ptr || !id2docword_fn || !handle)
        throw std::invalid_argument("prepare_async: bad arguments");
    auto P = std::make_shared<Prepared>();
    P->host = this; P->num_queries = num_queries; P->top_k = top_k; P->return_sent = return_sent; P->mode = mode_of(agg_strat);
    Prepared* p = P.get();
    P->worker = std::thread([this, p, I_ptr, best_ptr, pred_ptr, status_ptr, id2docword_fn, handle]() {
        try {
            const int B = p->num_queries, k = p->top_k;
            const int64_t bk = (int64_t)B * k;
            const int64_t* I = (const int64_t*)I_ptr;
            const double* best = (const double*)best_ptr;
            const int32_t* pred = (const int32_t*)pred_ptr;
            const int32_t* status = (const int32_t*)status_ptr;
            for (int r = 0; r < 2 * B; ++r) if (status[r] == 1) { p->needs_exact = true; return; }
            std::vector<int32_t> doc((size_t)(2 * bk)), word((size_t)(2 * bk));
            using fn_t = int (*)(void*, const int64_t*, int64_t, int32_t*, int32_t*);
            const int rc = ((fn_t)id2docword_fn)((void*)handle, I, 2 * bk, doc.data(), word.data());
            if (rc != 0) { p->error_kind = 2; p->error = "prepare_async: dph_id2docword failed (" + std::to_string(rc) + ")"; return; }
            p->D.resize((size_t)(2 * bk)); p->S.resize((size_t)(2 * bk)); p->E.resize((size_t)(2 * bk)); p->SC.resize((size_t)(2 * bk));
            for (int64_t c = 0; c < bk; ++c) {
                // start candidate c of the start rows [0, B): its end is pred[:B]; end candidate c of the end rows [B, 2B): its start is pred[B:]
                p->D[(size_t)(2 * c)] = doc[(size_t)c];            p->D[(size_t)(2 * c + 1)] = doc[(size_t)(bk + c)];
                p->S[(size_t)(2 * c)] = word[(size_t)c];           p->S[(size_t)(2 * c + 1)] = pred[bk + c];
                p->E[(size_t)(2 * c)] = pred[c];                   p->E[(size_t)(2 * c + 1)] = word[(size_t)(bk + c)];
                p->SC[(size_t)(2 * c)] = best[c];                  p->SC[(size_t)(2 * c + 1)] = best[bk + c];
            }
            // distinct documents per query among its 2 k candidates, mean over the batch (index.py:213-215)
            double tot = 0.0;
            std::vector<int32_t> tmp((size_t)(2 * k));
            for (int q = 0; q < B; ++q) {
                for (int j = 0; j < k; ++j) { tmp[(size_t)j] = doc[(size_t)((int64_t)q * k + j)]; tmp[(size_t)(k + j)] = doc[(size_t)(bk + (int64_t)q * k + j)]; }
                std::sort(tmp.begin(), tmp.end());
                tot += (double)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
            }
            p->num_docs = B > 0 ? tot / B : 0.0;
            resident_and_views(*p, false);
            phase1(*p);
        } catch (py::error_already_set& e) {
            py::gil_scoped_acquire gil;
            p->error_kind = 2; p->error = e.what();
            e.restore(); PyErr_Clear();
        } catch (const std::exception& e) {
            p->error_kind = 2; p->error = e.what();
        }
    });
    return P;
}

// MIPS.aggregate_results (index.py:424-448) for one query: de-duplicate by the strategy's key (the FIRST result with a
// key keeps its score, later ones drop to -1e8; opt4 also merges their titles into the first), sort by score, drop
// everything <= -1e5.  The keys are the reference's very strings (f'{title}_{start}_{end}', context, f'{title}',
// normalize_answer(answer)) so that even odd inputs collide exactly as they do there.
py::list aggregate(py::list results, const std::string& strat, py::object normalize) {
    init_keys();
    int mode = strat == "opt1" ? 1 : strat == "opt2" ? 2 : strat == "opt3" ? 3 : strat == "opt4" ? 4 : 0;
    if (!mode) throw py::type_error("wrong aggregation strategy");      // the reference raises NotImplementedError: mapped by the caller
    py::dict first;
    const Py_ssize_t n = PyList_GET_SIZE(results.ptr());
    std::vector<double> score((size_t)n);
    // opt1 / opt3 key on f'{title}_{start_pos}_{end_pos}' / f'{title}' with title a LIST: formatting a list's repr per result is most
    // of this function's time.  When EVERY result of the call has a title list of exactly one str (and exact ints as positions) --
    // what assemble produces -- the tuple (title[0], start_pos, end_pos) / (title[0],) separates the results exactly like the
    // strings do: same title text and same positions <=> same string.  One odd result sends the whole call down the string path
    // (a mixed call could otherwise miss a collision between an odd title's string and a regular one's).
    bool fast_keys = mode == 1 || mode == 3;
    for (Py_ssize_t i = 0; fast_keys && i < n; ++i) {
        PyObject* r = PyList_GET_ITEM(results.ptr(), i);
        PyObject* title = PyDict_Check(r) ? PyDict_GetItemWithError(r, k_title) : nullptr;
        if (!title || !PyList_CheckExact(title) || PyList_GET_SIZE(title) != 1 || !PyUnicode_CheckExact(PyList_GET_ITEM(title, 0))) fast_keys = false;
        else if (mode == 1) {
            PyObject *sp = PyDict_GetItemWithError(r, k_start_pos), *ep = PyDict_GetItemWithError(r, k_end_pos);
            if (!sp || !ep || !PyLong_CheckExact(sp) || !PyLong_CheckExact(ep)) fast_keys = false;
        }
    }
    if (PyErr_Occurred()) PyErr_Clear();             // (the string path below reports what is missing)
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* r = PyList_GET_ITEM(results.ptr(), i);
        PyObject* title = PyDict_GetItemWithError(r, k_title);
        if (!title) throw py::key_error("title");
        py::object key;
        if (fast_keys && mode == 1) {
            key = py::reinterpret_steal<py::object>(PyTuple_Pack(3, PyList_GET_ITEM(title, 0), PyDict_GetItemWithError(r, k_start_pos),
                                                                  PyDict_GetItemWithError(r, k_end_pos)));
        } else if (fast_keys && mode == 3) {
            key = py::reinterpret_steal<py::object>(PyTuple_Pack(1, PyList_GET_ITEM(title, 0)));
        } else if (mode == 1) {
            PyObject *sp = PyDict_GetItemWithError(r, k_start_pos), *ep = PyDict_GetItemWithError(r, k_end_pos);
            if (!sp || !ep) throw py::key_error("start_pos / end_pos");
            key = py::reinterpret_steal<py::object>(PyUnicode_FromFormat("%S_%S_%S", title, sp, ep));
        } else if (mode == 2) {
            PyObject* c = PyDict_GetItemWithError(r, k_context);
            if (!c) throw py::key_error("context");
            key = py::reinterpret_steal<py::object>(PyObject_Str(c));
        } else if (mode == 3) {
            key = py::reinterpret_steal<py::object>(PyObject_Str(title));
        } else {
            PyObject* a = PyDict_GetItemWithError(r, k_answer);
            if (!a) throw py::key_error("answer");
            key = normalize(py::reinterpret_borrow<py::object>(a));
        }
        if (!key) throw py::error_already_set();
        PyObject* sc = PyDict_GetItemWithError(r, k_score);
        if (!sc) throw py::key_error("score");
        score[(size_t)i] = PyFloat_AsDouble(sc);
        PyObject* f = PyDict_GetItemWithError(first.ptr(), key.ptr());
        if (!f) {
            if (PyErr_Occurred()) throw py::error_already_set();
            set_steal(first.ptr(), key.ptr(), PyLong_FromSsize_t(i));
        } else {
            score[(size_t)i] = -1e8;
            set_steal(r, k_score, PyFloat_FromDouble(-1e8));
            if (mode == 4) {
                PyObject* ft = PyDict_GetItemWithError(PyList_GET_ITEM(results.ptr(), PyLong_AsSsize_t(f)), k_title);
                PyObject* t0 = PySequence_GetItem(title, 0);
                if (!ft || !t0) throw py::error_already_set();
                const int has = PySequence_Contains(ft, t0);
                Py_DECREF(t0);
                if (has < 0) throw py::error_already_set();
                if (!has) {
                    PyObject* merged = PySequence_InPlaceConcat(ft, title);      // results[first]["title"] += r["title"]
                    if (!merged) throw py::error_already_set();
                    Py_DECREF(merged);
                }
            }
        }
    }
    std::vector<Py_ssize_t> order((size_t)n);
    for (Py_ssize_t i = 0; i < n; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](Py_ssize_t a, Py_ssize_t b) { return score[(size_t)a] > score[(size_t)b]; });
    py::list out;
    for (Py_ssize_t i : order)
        if (score[(size_t)i] > -1e5) out.append(py::reinterpret_borrow<py::object>(PyList_GET_ITEM(results.ptr(), i)));
    return out;
}

PYBIND11_MODULE(_dph_host, m) {
    m.doc() = "C++ host half of MIPS.search_phrase (dict assembly, paragraph / sentence cropping, per-query sort, de-duplication)";
    py::class_<Prepared, std::shared_ptr<Prepared>>(m, "Prepared", "a batch between the kernels and its result dicts (HostHalf.prepare_async)")
        .def("wait", [](Prepared& P) {
            if (P.worker.joinable()) { py::gil_scoped_release nogil; P.worker.join(); }
            if (P.error_kind == 1) throw std::out_of_range(P.error);
            if (P.error_kind) throw std::runtime_error(P.error);
            return py::make_tuple(P.needs_exact, P.num_docs);
        }, "blocks (GIL released) until the batch is prepared; (needs_exact, mean distinct documents per query)");
    py::class_<HostHalf>(m, "HostHalf")
        .def(py::init<py::function, size_t>(), py::arg("doc_meta"), py::arg("cache_docs") = 262144)
        .def("prepare_async", &HostHalf::prepare_async, py::arg("num_queries"), py::arg("top_k"), py::arg("I_ptr"), py::arg("best_ptr"), py::arg("pred_ptr"),
             py::arg("status_ptr"), py::arg("id2docword_fn"), py::arg("handle"), py::arg("return_sent") = false, py::arg("agg_strat") = py::none())
        .def("materialize", [](HostHalf& h, std::shared_ptr<Prepared> P, py::object normalize) { return h.materialize(*P, py::none(), py::none(), normalize); },
             py::arg("prepared"), py::arg("normalize") = py::none())
        .def("assemble", &HostHalf::assemble, py::arg("num_queries"), py::arg("top_k"), py::arg("doc_i"), py::arg("start_i"),
             py::arg("end_i"), py::arg("score_i"), py::arg("start_vecs"), py::arg("end_vecs"), py::arg("return_sent") = false,
             py::arg("agg_strat") = py::none(), py::arg("normalize") = py::none())
        .def("cached_docs", [](const HostHalf& h) { return h.docs.size() + h.old.size(); })
        .def("fetched_docs", [](const HostHalf& h) { return h.fetched; }, "python doc_meta calls so far");
    m.def("split_sentences", [](py::str text) {
        std::vector<std::pair<Py_ssize_t, Py_ssize_t>> sents;
        split_sentences(text.ptr(), sents);
        py::list out;
        for (const auto& sp : sents) out.append(py::make_tuple(sp.first, sp.second));
        return out;
    }, py::arg("text"), "[(start, end)) code-point spans of the sentence units (densephrases_amd/sentencizer.py is the same rule)");
    m.def("aggregate", &aggregate, py::arg("results"), py::arg("agg_strat"), py::arg("normalize"));
}
