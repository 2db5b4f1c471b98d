/*
 * opal_oracle.c -- CPU restatement of the reference's database-search DP.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (opal_amd/csrc, the
 * C-ABI library) includes, links or calls this file. Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it, as the checker.
 *
 * Parity status: PINNED. tests/test_oracle_golden.py checks this file against
 * (1) golden vectors produced in the build container by the reference's own
 * SSE4.1/AVX2 library compiled from /root/reference/src/opal.cpp
 * (oracle/build_ref.sh -> oracle/_ref/libopal_ref.so; generator
 * tests/golden/make_golden.py), on the reference's bundled test_data and on
 * seeded synthetic sets, and (2) the known answers listed in SURVEY.md 8(c).
 *
 * What it restates (plain scalar 64-bit integer arithmetic, one sequence at a
 * time, column by column, exactly the recurrences of the reference):
 *   SW            reference src/opal.cpp:164-470 (core :280-327), ground truth
 *                 src/test.cpp:199-251
 *   NW / HW / OV  reference src/opal.cpp:594-977 (core :748-771, init :671-689,
 *                 result rules :844-903), ground truth src/test.cpp:253-328
 *   API rules     reference src/opal.cpp:1435-1519 (mode dispatch, end = -1 for
 *                 score-only search), :1522-1546 (8-bit-only SW variant)
 *
 * End-location tie-breaks (reference src/opal.h:44-45):
 *   SW  first column (smallest target index) holding the maximum, then the
 *       smallest row in that column         (opal.cpp:306-312, 385-401)
 *   NW  (Q-1, L-1)                          (opal.cpp:873-876)
 *   HW  endQ = Q-1, endT = first column whose last-row H reaches the maximum
 *                                           (opal.cpp:824-834, 877-880)
 *   OV  if max of last column > best last-row H of columns 0..L-2:
 *       endT = L-1, endQ = smallest row attaining it; else endQ = Q-1 and
 *       endT = first column attaining the last-row maximum (opal.cpp:881-904)
 *
 * Corners where the reference is undefined and this file fixes a value
 * (excluded from diffs against oracle/_ref):
 *   - SW with score 0 and an end-location search: end = (-1,-1), the value of
 *     the reference's scalar ground truth (test.cpp:242-243) and of its
 *     alignment path (opal.cpp:1479-1483).
 *   - 32-bit NW/HW/OV overflow (reference reads an uninitialised array,
 *     opal.cpp:792-816): OPAL_ERR_OVERFLOW is returned iff the a-priori bound
 *     below says a value could leave [-2^30, 2^30].
 *   - zero-length target or query: boundary-column values (see code).
 */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ERR_OVERFLOW 1
#define ERR_INVALID_MODE 3
#define MODE_NW 0
#define MODE_HW 1
#define MODE_OV 2
#define MODE_SW 3
#define SEARCH_SCORE 0

typedef int64_t i64;
#define NEG_INF (INT64_MIN / 4)

static inline i64 max2(i64 a, i64 b) { return a > b ? a : b; }

/* reference opal.cpp:178-198 / :610-630, 32-bit instantiation: gap penalties and
 * every matrix entry must lie strictly inside (INT_MIN/2, INT_MAX/2). */
static int args_fit_int32(int gapOpen, int gapExt, const int* M, int A) {
    const int lo = INT_MIN / 2, hi = INT_MAX / 2;
    if (gapOpen <= lo || hi <= gapOpen || gapExt <= lo || hi <= gapExt) return 0;
    for (int i = 0; i < A * A; i++)
        if (M[i] <= lo || hi <= M[i]) return 0;
    return 1;
}

/* Local alignment of one pair. Returns 1 if a 32-bit wrap would have been
 * detected by the reference (some ulH + P > INT_MAX, opal.cpp:302-303,339-351). */
static int sw_one(const unsigned char* q, int Q, const unsigned char* t, int L,
                  int Go, int Ge, const int* M, int A,
                  int* score, int* endQ, int* endT, i64* Hc, i64* Ec) {
    i64 best = 0;
    int bq = -1, bt = -1, wrapped = 0;
    for (int r = 0; r < Q; r++) Hc[r] = Ec[r] = 0;
    for (int c = 0; c < L; c++) {
        i64 uF = 0, uH = 0, ulH = 0;
        for (int r = 0; r < Q; r++) {
            i64 E = max2(Hc[r] - Go, Ec[r] - Ge);
            i64 F = max2(uH - Go, uF - Ge);
            i64 d = ulH + M[q[r] * A + t[c]];
            if (d > INT_MAX) wrapped = 1;
            i64 H = max2(max2(0, E), max2(F, d));
            if (H > best) { best = H; bq = r; bt = c; } /* column-major scan, strict > */
            uF = F; uH = H; ulH = Hc[r];
            Hc[r] = H; Ec[r] = E;
        }
    }
    *score = (int)best; *endQ = bq; *endT = bt;
    return wrapped;
}

/* Global / semi-global alignment of one pair (mode NW, HW or OV). */
static void global_one(const unsigned char* q, int Q, const unsigned char* t, int L,
                       int Go, int Ge, const int* M, int A, int mode,
                       int* score, int* endQ, int* endT, i64* Hc, i64* Ec) {
    /* H[r][-1], E[r][-1]   (opal.cpp:671-682) */
    for (int r = 0; r < Q; r++) {
        Hc[r] = (mode == MODE_OV) ? 0 : -(i64)Go - (i64)r * Ge;
        Ec[r] = NEG_INF;
    }
    if (L == 0 || Q == 0) { /* degenerate: only the boundary exists */
        if (mode == MODE_NW) {
            int n = Q + L; /* one gap of length n (n may be 0) */
            *score = n ? (int)(-(i64)Go - (i64)(n - 1) * Ge) : 0;
        } else if (mode == MODE_HW) {
            *score = (Q && !L) ? (int)(-(i64)Go - (i64)(Q - 1) * Ge) : 0;
        } else {
            *score = 0;
        }
        *endQ = Q - 1; *endT = L - 1;
        return;
    }
    i64 bestLastRow = NEG_INF; /* max over columns of H[Q-1][c], first column wins */
    int bestLastRowCol = -1;
    i64 lastColMax = NEG_INF;  /* OV: max over rows of H[r][L-1], smallest row wins */
    int lastColRow = -1;
    i64 prevBestLastRow = NEG_INF;
    i64 H = NEG_INF;
    for (int c = 0; c < L; c++) {
        i64 uF = NEG_INF, uH, ulH;
        if (mode == MODE_NW) { /* H[-1][c] = -Go - c*Ge, H[-1][-1] = 0 (opal.cpp:686-689,717-729) */
            uH = -(i64)Go - (i64)c * Ge;
            ulH = c ? uH + Ge : 0;
        } else {
            uH = ulH = 0;
        }
        prevBestLastRow = bestLastRow;
        for (int r = 0; r < Q; r++) {
            i64 E = max2(Hc[r] - Go, Ec[r] - Ge);
            i64 F = max2(uH - Go, uF - Ge);
            H = max2(max2(E, F), ulH + M[q[r] * A + t[c]]);
            if (c == L - 1 && H > lastColMax) { lastColMax = H; lastColRow = r; }
            uF = F; uH = H; ulH = Hc[r];
            Hc[r] = H; Ec[r] = E;
        }
        if (H > bestLastRow) { bestLastRow = H; bestLastRowCol = c; }
    }
    if (mode == MODE_NW) {
        *score = (int)H; *endQ = Q - 1; *endT = L - 1;
    } else if (mode == MODE_HW) {
        *score = (int)bestLastRow; *endQ = Q - 1; *endT = bestLastRowCol;
    } else { /* OV (opal.cpp:844-846, 881-904) */
        if (lastColMax > prevBestLastRow) {
            *score = (int)lastColMax; *endT = L - 1; *endQ = lastColRow;
        } else {
            *score = (int)bestLastRow; *endT = bestLastRowCol; *endQ = Q - 1;
        }
    }
}

/* A-priori 32-bit safety bound for NW/HW/OV: every DP value of the pair lies in
 * [-(3*|Go| + (Q+L+2)*|Ge|), min(Q,L)*max(0,maxS)]; "overflow" iff that interval
 * is not inside [-2^30, 2^30]. (Our definition; the reference has none.) */
static int global_may_overflow32(int Q, int L, int Go, int Ge, const int* M, int A) {
    i64 maxS = 0;
    for (int i = 0; i < A * A; i++) if (M[i] > maxS) maxS = M[i];
    i64 lo = 3 * llabs((i64)Go) + ((i64)Q + L + 2) * llabs((i64)Ge);
    i64 hi = (i64)(Q < L ? Q : L) * maxS;
    return lo >= (1LL << 30) || hi >= (1LL << 30);
}

/*
 * Whole-database search. Outputs are plain arrays (score/endQ/endT per
 * sequence). Returns 0, ERR_OVERFLOW or ERR_INVALID_MODE with the meaning of the
 * reference's opalSearchDatabase return value.
 */
int oracle_search(const unsigned char* query, int Q,
                  const unsigned char* const* db, int N, const int* lens,
                  int gapOpen, int gapExt, const int* M, int A,
                  int searchType, int mode,
                  int* score, int* endQ, int* endT) {
    if (mode != MODE_NW && mode != MODE_HW && mode != MODE_OV && mode != MODE_SW)
        return ERR_INVALID_MODE;
    if (!args_fit_int32(gapOpen, gapExt, M, A)) return ERR_OVERFLOW;
    i64* Hc = (i64*)malloc(sizeof(i64) * (size_t)(Q > 0 ? Q : 1));
    i64* Ec = (i64*)malloc(sizeof(i64) * (size_t)(Q > 0 ? Q : 1));
    int rc = 0;
    for (int j = 0; j < N; j++) {
        int s = 0, eq = -1, et = -1;
        if (mode == MODE_SW) {
            if (sw_one(query, Q, db[j], lens[j], gapOpen, gapExt, M, A, &s, &eq, &et, Hc, Ec))
                rc = ERR_OVERFLOW;
        } else {
            if (global_may_overflow32(Q, lens[j], gapOpen, gapExt, M, A)) { rc = ERR_OVERFLOW; continue; }
            global_one(query, Q, db[j], lens[j], gapOpen, gapExt, M, A, mode, &s, &eq, &et, Hc, Ec);
        }
        score[j] = s;
        if (searchType == SEARCH_SCORE) { endQ[j] = -1; endT[j] = -1; } /* opal.cpp:423-426, 869-871 */
        else { endQ[j] = eq; endT[j] = et; }
    }
    free(Hc); free(Ec);
    return rc;
}

/*
 * 8-bit-only local search (reference opal.cpp:1522-1546 over :164-470 with
 * SimdSW<char>): the 8-bit pass represents 0..127; a sequence overflows iff some
 * ulH + P reaches 128, which (non-negative gap penalties) is equivalent to its
 * exact score being >= 128. Overflowed sequences: score -1, scoreSet 0.
 * Arguments that do not fit a signed char make the pass return 1 with no result
 * written (opal.cpp:178-198): scoreSet is then left 0 for every sequence.
 */
int oracle_char_sw(const unsigned char* query, int Q,
                   const unsigned char* const* db, int N, const int* lens,
                   int gapOpen, int gapExt, const int* M, int A,
                   int* score, int* scoreSet) {
    int fits = !(gapOpen < SCHAR_MIN || gapOpen > SCHAR_MAX || gapExt < SCHAR_MIN || gapExt > SCHAR_MAX);
    for (int i = 0; fits && i < A * A; i++)
        if (M[i] < SCHAR_MIN || M[i] > SCHAR_MAX) fits = 0;
    if (!fits) {
        for (int j = 0; j < N; j++) { score[j] = -1; scoreSet[j] = 0; }
        return ERR_OVERFLOW;
    }
    i64* Hc = (i64*)malloc(sizeof(i64) * (size_t)(Q > 0 ? Q : 1));
    i64* Ec = (i64*)malloc(sizeof(i64) * (size_t)(Q > 0 ? Q : 1));
    int rc = 0;
    for (int j = 0; j < N; j++) {
        int s, eq, et;
        sw_one(query, Q, db[j], lens[j], gapOpen, gapExt, M, A, &s, &eq, &et, Hc, Ec);
        if (s >= 128) { score[j] = -1; scoreSet[j] = 0; rc = ERR_OVERFLOW; }
        else { score[j] = s; scoreSet[j] = 1; }
    }
    free(Hc); free(Ec);
    return rc;
}

/*
 * Replay an alignment and re-score it (restates reference src/test.cpp:348-422).
 * Returns 0 if the operation string is a valid alignment from (startQ,startT)
 * to (endQ,endT) with exactly `score`, else a small positive code.
 */
int oracle_check_alignment(const unsigned char* q, int Q, const unsigned char* t, int L,
                           int gapOpen, int gapExt, const int* M, int A,
                           int score, int startQ, int startT, int endQ, int endT,
                           const unsigned char* ops, int nops) {
    i64 s = 0;
    int qi = startQ, ti = startT, prev = -1;
    for (int i = 0; i < nops; i++) {
        int op = ops[i];
        if ((op != 1 && ti >= L) || (op != 2 && qi >= Q)) return 1; /* left the matrix */
        switch (op) {
        case 0: if (q[qi] != t[ti]) return 2; s += M[q[qi] * A + t[ti]]; qi++; ti++; break;
        case 3: if (q[qi] == t[ti]) return 3; s += M[q[qi] * A + t[ti]]; qi++; ti++; break;
        case 1: s -= (prev == 1 ? gapExt : gapOpen); qi++; break; /* deletion from query */
        case 2: s -= (prev == 2 ? gapExt : gapOpen); ti++; break; /* insertion to query */
        default: return 4;
        }
        prev = op;
    }
    if (qi - 1 != endQ || ti - 1 != endT) return 5;
    if (s != score) return 6;
    return 0;
}
