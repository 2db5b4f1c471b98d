// opal_traceback.cpp -- host traceback for OPAL_SEARCH_ALIGNMENT (SURVEY.md 8f "next #1").
// Placeholder until the traceback lands: fails loudly instead of pretending.
#include <cstdio>

#include "opal.h"

int opal_find_alignments(unsigned char[], int, unsigned char*[], int, int[], int, int, int*, int, OpalSearchResult*[], int) {
    fprintf(stderr, "[opal] OPAL_SEARCH_ALIGNMENT is not implemented yet in this build\n");
    return OPAL_ERR_NO_SIMD_SUPPORT;
}
