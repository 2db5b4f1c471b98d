// host_stress.cpp -- TEST INFRASTRUCTURE ONLY: drives the library's host scheduler (the real opal_api / db_pack / planner /
// ladder / multi_gpu / options translation units, compiled with ThreadSanitizer) on the fake device of fake_shim.cpp and
// checks every answer against the scalar oracle. What it exercises is the CONCURRENT host code no GPU test can check for
// races: the thread pool, the gather beside the search and its chunk gates, the speculative search of a database seen again
// (unchanged, edited in place, with another length), zero-copy fills, the queue of asynchronous searches on a handle, two
// caller threads at once, a call sharded over several device slots, and the multi-GPU handle's queued searches.
// Exit code 0 = every answer right (ThreadSanitizer adds its own non-zero exit for a race: TSAN_OPTIONS=exitcode=66).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "opal.h"
#include "opal_hip.h"

extern "C" int oracle_search(const unsigned char* query, int Q, const unsigned char* const* db, int N, const int* lens, int gapOpen,
                             int gapExt, const int* M, int A, int searchType, int mode, int* score, int* endQ, int* endT);
extern "C" int opalx_set_option(const char* name, const char* value);

namespace {

constexpr int A = 24;
int g_fail = 0;

struct Db {
    std::vector<std::vector<unsigned char>> seqs;
    std::vector<unsigned char*> ptr;
    std::vector<int> len;
    void index() {
        ptr.resize(seqs.size()); len.resize(seqs.size());
        static unsigned char dummy = 0;
        for (size_t i = 0; i < seqs.size(); i++) { ptr[i] = seqs[i].empty() ? &dummy : seqs[i].data(); len[i] = (int)seqs[i].size(); }
    }
};

Db make_db(int n, unsigned seed, int longest) {
    std::mt19937 rng(seed);
    Db d;
    d.seqs.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        int L = 1 + (int)(rng() % 60);
        if (i % 97 == 5) L = 0;
        if (i % 501 == 7) L = longest - (int)(rng() % 50);
        d.seqs[(size_t)i].resize((size_t)L);
        for (auto& c : d.seqs[(size_t)i]) c = (unsigned char)(rng() % 20);
    }
    d.index();
    return d;
}

std::vector<unsigned char> make_query(int Q, unsigned seed) {
    std::mt19937 rng(seed);
    std::vector<unsigned char> q((size_t)Q);
    for (auto& c : q) c = (unsigned char)(rng() % 20);
    return q;
}

std::vector<int> make_matrix(int scale) {
    std::mt19937 rng(3);
    std::vector<int> m(A * A);
    for (int a = 0; a < A; a++)
        for (int b = 0; b <= a; b++) {
            const int v = a == b ? 4 + (int)(rng() % 8) : (int)(rng() % 9) - 5;
            m[a * A + b] = m[b * A + a] = v * scale;
        }
    return m;
}

struct Want { int rc; std::vector<int> s, eq, et; };
Want oracle(const std::vector<unsigned char>& q, const Db& d, int go, int ge, const std::vector<int>& m, int st, int mode) {
    Want w;
    const size_t n = d.seqs.size();
    w.s.resize(n); w.eq.resize(n); w.et.resize(n);
    w.rc = oracle_search(q.data(), (int)q.size(), (const unsigned char* const*)d.ptr.data(), (int)n, d.len.data(), go, ge, m.data(), A, st, mode,
                         w.s.data(), w.eq.data(), w.et.data());
    return w;
}

void check(const char* what, const Want& w, int rc, const int* s, const int* eq, const int* et, size_t n, int st, int mode) {
    if (rc != w.rc) { fprintf(stderr, "FAIL %s: rc %d, oracle %d\n", what, rc, w.rc); g_fail++; return; }
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) {
        bool ok = s[i] == w.s[i];
        if (ok && st && !(mode == OPAL_MODE_SW && w.s[i] == 0) && eq && et) ok = eq[i] == w.eq[i] && et[i] == w.et[i];
        if (!ok && bad++ < 3) fprintf(stderr, "FAIL %s: #%zu got %d (%d,%d) want %d (%d,%d)\n", what, i, s[i], eq ? eq[i] : -9, et ? et[i] : -9, w.s[i], w.eq[i], w.et[i]);
    }
    if (bad) g_fail++;
}

int api_search(const std::vector<unsigned char>& q, Db& d, int go, int ge, std::vector<int>& m, int st, int mode, std::vector<int>& s,
               std::vector<int>& eq, std::vector<int>& et) {
    const size_t n = d.seqs.size();
    std::vector<OpalSearchResult> store(n);
    std::vector<OpalSearchResult*> res(n);
    for (size_t i = 0; i < n; i++) { opalInitSearchResult(&store[i]); res[i] = &store[i]; }
    const int rc = opalSearchDatabase(const_cast<unsigned char*>(q.data()), (int)q.size(), d.ptr.data(), (int)n, d.len.data(), go, ge, m.data(), A,
                                      res.data(), st, mode, OPAL_OVERFLOW_BUCKETS);
    s.resize(n); eq.resize(n); et.resize(n);
    for (size_t i = 0; i < n; i++) { s[i] = store[i].scoreSet ? store[i].score : -123456789; eq[i] = store[i].endLocationQuery; et[i] = store[i].endLocationTarget; }
    return rc;
}

void api_round(const char* tag, const std::vector<unsigned char>& q, Db& d, int go, int ge, std::vector<int>& m, int st, int mode) {
    std::vector<int> s, eq, et;
    const int rc = api_search(q, d, go, ge, m, st, mode, s, eq, et);
    char what[160];
    snprintf(what, sizeof what, "%s Q=%zu mode=%d st=%d", tag, q.size(), mode, st);
    check(what, oracle(q, d, go, ge, m, st, mode), rc, s.data(), eq.data(), et.data(), s.size(), st, mode);
}

}  // namespace

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 17000;  // (>= 16 384: the ordering takes the device path)
    std::vector<int> m = make_matrix(1);
    std::vector<int> m3 = make_matrix(40);  // scores beyond the 16-bit rung: the re-run list
    Db d = make_db(n, 1, 400);
    Db d2 = make_db(n / 2 + 3, 2, 600);
    const auto q60 = make_query(30, 7), q300 = make_query(120, 8);

    // 1. the plain API: a new database, the same again (speculative search), edited in place (the speculation fails and the
    //    search is repeated), with one length changed (a new database), another database, every mode
    for (int mode : {OPAL_MODE_SW, OPAL_MODE_NW, OPAL_MODE_HW, OPAL_MODE_OV})
        for (int st : {0, 1}) {
            api_round("new", q60, d, 3, 1, m, st, mode);
            api_round("same", q60, d, 3, 1, m, st, mode);
            for (int k = 0; k < 40; k++) { auto& s = d.seqs[(size_t)(k * 131 % n)]; if (!s.empty()) s[s.size() / 2] = (unsigned char)((s[s.size() / 2] + 1) % 20); }
            api_round("edited", q60, d, 3, 1, m, st, mode);
            api_round("edited-again", q300, d, 11, 1, m, st, mode);
            d.seqs[11].push_back(3); d.index();
            api_round("length-changed", q60, d, 3, 1, m, st, mode);
            api_round("other-db", q60, d2, 3, 1, m, st, mode);
        }
    // scores beyond the 16-bit rung (copies of the query, a matrix scaled by 40): the re-run list and the 32-bit launches
    for (int k : {5, 900, 4001}) d2.seqs[(size_t)k] = q300;
    d2.index();
    api_round("saturating", q300, d2, 3, 1, m3, 1, OPAL_MODE_SW);
    api_round("saturating-same", q300, d2, 3, 1, m3, 0, OPAL_MODE_SW);

    // ... and with small upload chunks, so that these small databases arrive in many chunks: sliced launches behind chunk gates,
    // zero-copy fills, chunks copied on demand
    opalx_set_option("OPAL_HIP_CHUNK_BYTES", "65536");
    for (int mode : {OPAL_MODE_SW, OPAL_MODE_HW})
        for (int st : {0, 1}) {
            api_round("chunks-new", q60, d, 3, 1, m, st, mode);
            api_round("chunks-same", q300, d, 3, 1, m, st, mode);
            d.seqs[300][0] = (unsigned char)((d.seqs[300][0] + 1) % 20);
            api_round("chunks-edited", q60, d, 3, 1, m, st, mode);
            api_round("chunks-other", q60, d2, 3, 1, m, st, mode);
        }

    // ... and a call that finds a third of its results computed already (opal.h:118-122): those stay as they are, the look
    // through the result structs -- which runs while the device orders the lengths -- drops that ordering and the rest is searched
    {
        const size_t nn = d.seqs.size();
        std::vector<OpalSearchResult> store(nn);
        std::vector<OpalSearchResult*> res(nn);
        for (size_t i = 0; i < nn; i++) {
            opalInitSearchResult(&store[i]); res[i] = &store[i];
            if (i % 3 == 0) { opalSearchResultSetScore(&store[i], -5); store[i].endLocationQuery = 1; store[i].endLocationTarget = 1; }
        }
        const int rc = opalSearchDatabase(const_cast<unsigned char*>(q60.data()), (int)q60.size(), d.ptr.data(), (int)nn, d.len.data(), 3, 1, m.data(), A,
                                          res.data(), OPAL_SEARCH_SCORE_END, OPAL_MODE_SW, OPAL_OVERFLOW_BUCKETS);
        Want w = oracle(q60, d, 3, 1, m, 1, OPAL_MODE_SW);
        size_t bad = rc != w.rc;
        for (size_t i = 0; i < nn; i++) {
            if (i % 3 == 0) bad += !(store[i].score == -5 && store[i].endLocationQuery == 1 && store[i].endLocationTarget == 1);
            else bad += !(store[i].scoreSet && store[i].score == w.s[i] && (w.s[i] == 0 || (store[i].endLocationQuery == w.eq[i] && store[i].endLocationTarget == w.et[i])));
        }
        if (bad) { fprintf(stderr, "FAIL partial results: %zu wrong\n", bad); g_fail++; }
    }

    // 2. two caller threads at once (cached contexts, the shared pool)
    {
        std::thread t1([&] { for (int r = 0; r < 3; r++) api_round("thread-1", q60, d, 3, 1, m, 1, OPAL_MODE_SW); });
        std::thread t2([&] { std::vector<int> mm = m; for (int r = 0; r < 3; r++) api_round("thread-2", q300, d2, 11, 1, mm, 1, OPAL_MODE_HW); });
        t1.join(); t2.join();
    }

    // 3. a prepared database with three searches queued on its handle
    {
        OpalxDb* h = opalx_db_create(d.ptr.data(), (int)d.seqs.size(), d.len.data());
        if (!h) { fprintf(stderr, "FAIL opalx_db_create\n"); return 1; }
        const size_t nn = d.seqs.size();
        std::vector<std::vector<int>> out(3, std::vector<int>(3 * nn, -77));
        const std::vector<unsigned char>* qs[3] = {&q60, &q300, &q60};
        const int modes[3] = {OPAL_MODE_SW, OPAL_MODE_OV, OPAL_MODE_NW};
        int rcs[3];
        for (int rep = 0; rep < 2; rep++) {
            for (int k = 0; k < 3; k++)
                rcs[k] = opalx_search_device_async(h, qs[k]->data(), (int)qs[k]->size(), 3, 1, m.data(), A, 1, modes[k], out[k].data(), out[k].data() + nn, out[k].data() + 2 * nn);
            float ms;
            for (int k = 0; k < 3; k++) opalx_db_wait(h, &ms);
            for (int k = 0; k < 3; k++) {
                Want w = oracle(*qs[k], d, 3, 1, m, 1, modes[k]);
                // (zero-length sequences are answered by the host part of the search as well)
                check("queued", w, rcs[k], out[k].data(), out[k].data() + nn, out[k].data() + 2 * nn, nn, 1, modes[k]);
            }
        }
        opalx_db_destroy(h);
    }

    // 4. one call sharded over three device slots (OPAL_HIP_DEVICES: one persistent host worker per slot)
    opalx_set_option("OPAL_HIP_DEVICES", "0,0,0");
    for (int rep = 0; rep < 2; rep++) {
        api_round("sharded", q60, d, 3, 1, m, 1, OPAL_MODE_SW);
        api_round("sharded", q300, d2, 11, 1, m, 1, OPAL_MODE_OV);
    }
    opalx_set_option("OPAL_HIP_DEVICES", nullptr);

    // 5. the multi-GPU handle: queued searches, gathered on the "root" by peer copies
    {
        std::vector<unsigned char> flat;
        std::vector<int64_t> offs(1, 0);
        for (auto& s : d.seqs) { flat.insert(flat.end(), s.begin(), s.end()); offs.push_back((int64_t)flat.size()); }
        const int devs[3] = {0, 0, 0};
        OpalxMultiDb* mh = opalx_multi_create_flat(flat.data(), offs.data(), (int)d.seqs.size(), devs, 3);
        if (!mh) { fprintf(stderr, "FAIL opalx_multi_create_flat\n"); return 1; }
        const size_t nn = d.seqs.size();
        std::vector<std::vector<int>> out(2, std::vector<int>(3 * nn, -77));
        for (int rep = 0; rep < 2; rep++) {
            const int r0 = opalx_multi_search_device_async(mh, q60.data(), (int)q60.size(), 3, 1, m.data(), A, 1, OPAL_MODE_SW, out[0].data(), out[0].data() + nn, out[0].data() + 2 * nn);
            const int r1 = opalx_multi_search_device_async(mh, q300.data(), (int)q300.size(), 11, 1, m.data(), A, 1, OPAL_MODE_HW, out[1].data(), out[1].data() + nn, out[1].data() + 2 * nn);
            float ms;
            opalx_multi_wait(mh, &ms); opalx_multi_wait(mh, &ms);
            check("multi", oracle(q60, d, 3, 1, m, 1, OPAL_MODE_SW), r0, out[0].data(), out[0].data() + nn, out[0].data() + 2 * nn, nn, 1, OPAL_MODE_SW);
            check("multi", oracle(q300, d, 11, 1, m, 1, OPAL_MODE_HW), r1, out[1].data(), out[1].data() + nn, out[1].data() + 2 * nn, nn, 1, OPAL_MODE_HW);
        }
        opalx_multi_destroy(mh);
    }
    opalx_release_cached();
    if (g_fail) { fprintf(stderr, "host_stress: %d check(s) failed\n", g_fail); return 1; }
    printf("host_stress ok\n");
    return 0;
}
