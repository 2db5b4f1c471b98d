/*
 * ref_bench.c -- times the reference's own SIMD library (oracle/_ref/libopal_ref.so, symbols
 * renamed ref_*) on all host cores with plain pthreads: the database is cut into contiguous shards,
 * one thread per shard, each calling the re-entrant reference entry point (SURVEY.md 8d "CPU
 * baseline"). TEST/BENCH INFRASTRUCTURE ONLY -- used by bench.py's cpu_baseline leg.
 *
 * double ref_bench_run(query, Q, residues, offsets, N, gapOpen, gapExt, matrix, A, mode, threads,
 *                      int* scores_out)   -> wall seconds
 * double ref_search_run(..., mode, searchType, threads, scores_out, endq_out, endt_out, int* rc_out)
 *      the same with the search type of the call (0 score, 1 score + end cells) and the end cells: what the full-size
 *      parity tests compare EVERY sequence of C2 / C3 with (tests/test_gpu_fullsize.py). *rc_out = the largest return code.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

struct OpalSearchResult {
    int scoreSet, score, endLocationTarget, endLocationQuery, startLocationTarget, startLocationQuery;
    unsigned char* alignment;
    int alignmentLength;
};
extern int ref_opalSearchDatabase(unsigned char query[], int queryLength, unsigned char* db[], int dbLength,
                                  int dbSeqLengths[], int gapOpen, int gapExt, int* scoreMatrix, int alphabetLength,
                                  struct OpalSearchResult* results[], const int searchType, int mode, int overflowMethod);
extern void ref_opalInitSearchResult(struct OpalSearchResult* result);

struct Shard {
    unsigned char* query; int Q;
    unsigned char** db; int* lens; struct OpalSearchResult** res; int n;
    int go, ge; int* m; int A; int mode; int rc; int searchType;
};

static void* worker(void* arg) {
    struct Shard* s = (struct Shard*)arg;
    s->rc = ref_opalSearchDatabase(s->query, s->Q, s->db, s->n, s->lens, s->go, s->ge, s->m, s->A, s->res,
                                   s->searchType, s->mode, 1 /* OPAL_OVERFLOW_BUCKETS */);
    return 0;
}

double ref_search_run(unsigned char* query, int Q, unsigned char* residues, const int64_t* offsets, int N, int gapOpen,
                      int gapExt, int* matrix, int A, int mode, int searchType, int threads, int* scores_out, int* endq_out,
                      int* endt_out, int* rc_out) {
    unsigned char** db = (unsigned char**)malloc(sizeof(*db) * (size_t)N);
    int* lens = (int*)malloc(sizeof(int) * (size_t)N);
    struct OpalSearchResult* store = (struct OpalSearchResult*)malloc(sizeof(*store) * (size_t)N);
    struct OpalSearchResult** res = (struct OpalSearchResult**)malloc(sizeof(*res) * (size_t)N);
    for (int i = 0; i < N; i++) {
        db[i] = residues + offsets[i];
        lens[i] = (int)(offsets[i + 1] - offsets[i]);
        ref_opalInitSearchResult(&store[i]);
        res[i] = &store[i];
    }
    if (threads < 1) threads = 1;
    struct Shard* sh = (struct Shard*)calloc((size_t)threads, sizeof(*sh));
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    /* shards of equal residue count */
    int64_t total = offsets[N], per = total / threads + 1;
    int start = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        int end = start;
        while (end < N && (t == threads - 1 || offsets[end] < (int64_t)(t + 1) * per)) end++;
        sh[t] = (struct Shard){query, Q, db + start, lens + start, res + start, end - start, gapOpen, gapExt, matrix, A, mode, 0, searchType};
        pthread_create(&th[t], 0, worker, &sh[t]);
        start = end;
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (scores_out)
        for (int i = 0; i < N; i++) scores_out[i] = store[i].scoreSet ? store[i].score : -2147483647 - 1;
    if (endq_out)
        for (int i = 0; i < N; i++) endq_out[i] = store[i].endLocationQuery;
    if (endt_out)
        for (int i = 0; i < N; i++) endt_out[i] = store[i].endLocationTarget;
    if (rc_out) {
        *rc_out = 0;
        for (int t = 0; t < threads; t++) if (sh[t].rc > *rc_out) *rc_out = sh[t].rc;
    }
    free(db); free(lens); free(store); free(res); free(sh); free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

double ref_bench_run(unsigned char* query, int Q, unsigned char* residues, const int64_t* offsets, int N, int gapOpen,
                     int gapExt, int* matrix, int A, int mode, int threads, int* scores_out) {
    return ref_search_run(query, Q, residues, offsets, N, gapOpen, gapExt, matrix, A, mode, 0 /* OPAL_SEARCH_SCORE */, threads,
                          scores_out, 0, 0, 0);
}
