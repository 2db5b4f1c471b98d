/*
 * opal.h -- public C API of the MI355X-native database-search aligner.
 *
 * This header is the drop-in boundary. It re-declares, name for name and value
 * for value, the interface of the reference library so that a caller built
 * against the reference header links against this library unchanged:
 *
 *   constants            reference src/opal.h:17-40
 *   OpalSearchResult     reference src/opal.h:47-74   (40 bytes, offsets 0/4/8/12/16/20/24/32)
 *   opalInitSearchResult reference src/opal.h:82,  src/opal.cpp:1549-1555
 *   opalSearchResultIsEmpty  reference src/opal.h:87,  src/opal.cpp:1557-1559 (struct BY VALUE)
 *   opalSearchResultSetScore reference src/opal.h:89,  src/opal.cpp:1561-1564
 *   opalSearchDatabase   reference src/opal.h:150-154, src/opal.cpp:1435-1519
 *   opalSearchDatabaseCharSW reference src/opal.h:162-165, src/opal.cpp:1522-1546
 *
 * Behaviour notes that differ from a CPU SIMD build are in DESIGN.md; results
 * (score, end locations) are bit-exact integers.
 */
#ifndef OPAL_H
#define OPAL_H

#ifdef __cplusplus
extern "C" {
#endif

/* Return codes (0 = success). */
#define OPAL_ERR_OVERFLOW 1        /* a score does not fit a 32-bit int, or gap/matrix values out of range */
#define OPAL_ERR_NO_SIMD_SUPPORT 2 /* here: no usable gfx950 device / HIP runtime failure (loud, never a CPU fallback) */
#define OPAL_ERR_INVALID_MODE 3    /* `mode` is not one of OPAL_MODE_* */

/* Alignment modes. */
#define OPAL_MODE_NW 0 /* global */
#define OPAL_MODE_HW 1 /* semi-global: gaps before/after the query are free */
#define OPAL_MODE_OV 2 /* semi-global: gaps before/after query and target are free */
#define OPAL_MODE_SW 3 /* local */

/* Overflow handling. Both produce identical results; see DESIGN.md "precision ladder". */
#define OPAL_OVERFLOW_SIMPLE 0
#define OPAL_OVERFLOW_BUCKETS 1

/* Search types. */
#define OPAL_SEARCH_SCORE 0     /* score only */
#define OPAL_SEARCH_SCORE_END 1 /* score + end location */
#define OPAL_SEARCH_ALIGNMENT 2 /* score + end + start location + alignment operations */

/* Alignment operations. */
#define OPAL_ALIGN_MATCH 0
#define OPAL_ALIGN_DEL 1 /* deletion from query (insertion to target) */
#define OPAL_ALIGN_INS 2 /* insertion to query (deletion from target) */
#define OPAL_ALIGN_MISMATCH 3

/*
 * One result per database sequence. Among equally scoring alignments the one
 * whose end has the smallest target position, then the smallest query position,
 * is reported.
 */
struct OpalSearchResult {
    int scoreSet;            /* 1 if `score` (at least) is valid, 0 if the result is empty */
    int score;
    int endLocationTarget;   /* 0-based, -1 if not set */
    int endLocationQuery;    /* 0-based, -1 if not set */
    int startLocationTarget; /* 0-based, -1 if not set */
    int startLocationQuery;  /* 0-based, -1 if not set */
    unsigned char* alignment; /* malloc()ed array of OPAL_ALIGN_* (caller free()s), NULL if none */
    int alignmentLength;
};
#ifndef __cplusplus
typedef struct OpalSearchResult OpalSearchResult;
#endif

/* Reset a result to "empty". Does not free `alignment`. */
void opalInitSearchResult(OpalSearchResult* result);

/* 1 if the result holds no score, else 0. Takes the struct by value, like the reference. */
int opalSearchResultIsEmpty(const OpalSearchResult result);

void opalSearchResultSetScore(OpalSearchResult* result, int score);

/*
 * Align `query` against every sequence of `db` with affine gaps (a gap of
 * length n costs gapOpen + (n-1)*gapExt). Sequences are arrays of alphabet
 * indices; scoreMatrix[q * alphabetLength + t] scores query letter q against
 * target letter t. results[i] must be initialised; a non-empty result (with end
 * location, if one is asked for) is kept and not recomputed.
 * Returns 0 or an OPAL_ERR_* code.
 */
int opalSearchDatabase(
    unsigned char query[], int queryLength, unsigned char* db[], int dbLength,
    int dbSeqLengths[], int gapOpen, int gapExt, int* scoreMatrix,
    int alphabetLength, OpalSearchResult* results[],
    const int searchType, int mode, int overflowMethod);

/*
 * Local alignment restricted to the 8-bit score range: sequences whose score
 * does not fit get score = -1 / scoreSet = 0 and the call returns
 * OPAL_ERR_OVERFLOW. Score only.
 */
int opalSearchDatabaseCharSW(
    unsigned char query[], int queryLength, unsigned char** db, int dbLength,
    int dbSeqLengths[], int gapOpen, int gapExt, int* scoreMatrix,
    int alphabetLength, OpalSearchResult* results[]);

#ifdef __cplusplus
}
#endif

#endif /* OPAL_H */
