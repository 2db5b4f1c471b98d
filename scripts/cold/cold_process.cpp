// cold_process.cpp -- what the FIRST opalSearchDatabase call of a process costs (the reference's usage is one call per process,
// reference src/opal_aligner.cpp:125-160): from the start of main() to the first result, split into
//   runtime   the first HIP call (device count + one allocation: runtime and device initialisation),
//   first     the first opalSearchDatabase call (code-object load on first launch, streams, pinned and device buffers, search),
//   second    the same call again (steady state of this process).
// The database comes from a binary file written by scripts/cold/cold_process.py (reading FASTA is excluded, as the reference
// excludes it from its own timings, aligner_comparison.md:16).   usage: cold_process <db.bin> <mode> <searchType>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "opal.h"
#include "opal_hip.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const double t0 = now_ms();
    if (argc < 2) return 2;
    const int mode = argc > 2 ? atoi(argv[2]) : OPAL_MODE_SW, st = argc > 3 ? atoi(argv[3]) : OPAL_SEARCH_SCORE;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[4];  // Q, N, A, reserved
    if (fread(hdr, sizeof(int), 4, f) != 4) return 2;
    const int Q = hdr[0], N = hdr[1], A = hdr[2];
    std::vector<unsigned char> query((size_t)Q);
    std::vector<int> matrix((size_t)A * A), lens((size_t)N);
    if (fread(query.data(), 1, (size_t)Q, f) != (size_t)Q || fread(matrix.data(), sizeof(int), (size_t)A * A, f) != (size_t)A * A ||
        fread(lens.data(), sizeof(int), (size_t)N, f) != (size_t)N) return 2;
    long long total = 0;
    for (int l : lens) total += l;
    std::vector<unsigned char> res((size_t)total + 1);
    if (fread(res.data(), 1, (size_t)total, f) != (size_t)total) return 2;
    fclose(f);
    std::vector<unsigned char*> db((size_t)N);
    std::vector<OpalSearchResult> store((size_t)N);
    std::vector<OpalSearchResult*> results((size_t)N);
    long long at = 0;
    for (int i = 0; i < N; i++) { db[(size_t)i] = res.data() + at; at += lens[(size_t)i]; results[(size_t)i] = &store[(size_t)i]; }
    const double t1 = now_ms();
    const int ndev = opal_hip_device_count();
    void* p = opal_hip_malloc(256);
    const double t2 = now_ms();
    // the code object of the library: loaded when the first kernel of it is asked about (a register-count query is enough)
    OpalHipLaunch probe;
    memset(&probe, 0, sizeof probe);
    probe.width = 17; probe.mode = OPAL_MODE_SW; probe.rows_per_lane = 38; probe.lanes_per_seq = 8;
    const int vg = opal_hip_kernel_vgprs(&probe);
    const double t3 = now_ms();
    double call[3];
    long long sum = 0;
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < N; i++) opalInitSearchResult(&store[(size_t)i]);
        const double a = now_ms();
        const int rc = opalSearchDatabase(query.data(), Q, db.data(), N, lens.data(), 3, 1, matrix.data(), A, results.data(), st, mode, OPAL_OVERFLOW_BUCKETS);
        call[rep] = now_ms() - a;
        if (rc) { fprintf(stderr, "rc %d\n", rc); return 1; }
        sum = 0;
        for (int i = 0; i < N; i++) sum += store[(size_t)i].score;
    }
    opal_hip_free(p);
    printf("devices %d  N %d Q %d  read_input %.1f ms | runtime_init %.1f ms | code_object_load %.1f ms (%d VGPRs) | first_call %.1f ms | second %.2f ms | third %.2f ms | "
           "main_to_first_result %.1f ms | score_sum %lld\n",
           ndev, N, Q, t1 - t0, t2 - t1, t3 - t2, vg, call[0], call[1], call[2], (t3 - t0) + call[0], sum);
    return 0;
}
