// opal_cli.cpp -- command-line front end of the MI355X aligner.
//
// Same interface and the same output lines as the reference's opal_aligner (reference
// src/opal_aligner.cpp:19-99 options, :121-160 database chunks, :169-189 result lines, :303-339 alignment
// blocks), so that the two programs' outputs diff textually:
//     opal_hip_aligner [-a SW|NW|HW|OV] [-o N] [-e N] [-m Blosum50] [-f FILE] [-x 0|1|2] [-s] <query.fasta> <db.fasta>
// What differs on purpose: residue and sequence totals are 64-bit (the reference's `int` wraps beyond 2^31
// residues, opal_aligner.cpp:123), the search is timed with a wall clock next to the reference's CPU clock
// (the GPU does the work while the calling thread waits), and the FASTA reader works on whole buffers.
// Own code throughout; only the printed text is shared with the reference.
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "opal.h"

namespace {

// BLOSUM50 over ARNDCQEGHILKMFPSTWYVBZX*, lower triangle (published table; the matrix is symmetric)
const char kBlosumAlphabet[] = "ARNDCQEGHILKMFPSTWYVBZX*";
const signed char kBlosum50Lower[] = {
    5,
    -2, 7,
    -1, -1, 7,
    -2, -2, 2, 8,
    -1, -4, -2, -4, 13,
    -1, 1, 0, 0, -3, 7,
    -1, 0, 0, 2, -3, 2, 6,
    0, -3, 0, -1, -3, -2, -3, 8,
    -2, 0, 1, -1, -3, 1, 0, -2, 10,
    -1, -4, -3, -4, -2, -3, -4, -4, -4, 5,
    -2, -3, -4, -4, -2, -2, -3, -4, -3, 2, 5,
    -1, 3, 0, -1, -3, 2, 1, -2, 0, -3, -3, 6,
    -1, -2, -2, -4, -2, 0, -2, -3, -1, 2, 3, -2, 7,
    -3, -3, -4, -5, -2, -4, -3, -4, -1, 0, 1, -4, 0, 8,
    -1, -3, -2, -1, -4, -1, -1, -2, -2, -3, -4, -1, -3, -4, 10,
    1, -1, 1, 0, -1, 0, -1, 0, -1, -3, -3, 0, -2, -3, -1, 5,
    0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 2, 5,
    -3, -3, -4, -5, -5, -1, -3, -3, -3, -3, -2, -3, -1, 1, -4, -4, -3, 15,
    -2, -1, -2, -3, -3, -1, -2, -3, 2, -1, -1, -2, 0, 4, -3, -2, -2, 2, 8,
    0, -3, -3, -4, -1, -3, -3, -4, -4, 4, 1, -3, 1, -1, -3, -2, 0, -3, -1, 5,
    -2, -1, 5, 6, -3, 0, 1, -1, 0, -4, -4, 0, -3, -4, -2, 0, 0, -5, -3, -3, 6,
    -1, 0, 0, 1, -3, 4, 5, -2, 0, -3, -3, 1, -1, -4, -1, 0, -1, -2, -2, -3, 1, 5,
    -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
    -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, -5, 1};

struct Scoring {
    std::string alphabet;
    std::vector<int> matrix;  // alphabet.size()^2, row = query letter
};

Scoring blosum50() {
    Scoring s;
    s.alphabet = kBlosumAlphabet;
    const int n = (int)s.alphabet.size();
    s.matrix.assign((size_t)n * n, 0);
    int k = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++, k++) s.matrix[(size_t)i * n + j] = s.matrix[(size_t)j * n + i] = kBlosum50Lower[k];
    return s;
}

// Matrix file format of the reference (src/ScoreMatrix.cpp:17-35): first line the letters, then the scores.
bool load_scoring(const char* path, Scoring& s) {
    std::ifstream in(path);
    if (!in) return false;
    std::string line, tok;
    if (!std::getline(in, line)) return false;
    std::istringstream letters(line);
    s.alphabet.clear();
    while (letters >> tok) s.alphabet.push_back(tok[0]);
    s.matrix.clear();
    int v;
    while (in >> v) s.matrix.push_back(v);
    return !s.alphabet.empty() && s.matrix.size() == s.alphabet.size() * s.alphabet.size();
}

// FASTA records as one flat residue buffer + offsets (indices into the alphabet). A '*' in the alphabet is the
// catch-all for letters outside it (reference opal_aligner.cpp:250-258).
struct Records {
    std::vector<unsigned char> residues;
    std::vector<int64_t> offsets{0};
    size_t count() const { return offsets.size() - 1; }
};

class FastaReader {
public:
    FastaReader(FILE* f, const std::string& alphabet) : f_(f) {
        unsigned char fallback = 0;
        const size_t star = alphabet.find('*');
        if (star != std::string::npos) fallback = (unsigned char)star;
        for (int c = 0; c < 256; c++) table_[c] = fallback;
        for (size_t i = 0; i < alphabet.size(); i++) table_[(unsigned char)alphabet[i]] = (unsigned char)i;
    }
    // Reads records until the end of the file or until more than `limit` residues of COMPLETE earlier records
    // have been read (the reference cuts its database into chunks of 2^30 residues the same way,
    // opal_aligner.cpp:283-290). Returns true at the end of the file.
    bool read(Records& out, int64_t limit) {
        out.residues.clear();
        out.offsets.assign(1, 0);
        bool open = false;  // a record is being filled
        for (;;) {
            if (pos_ == len_) {
                len_ = fread(buf_, 1, sizeof buf_, f_);
                pos_ = 0;
                if (len_ == 0) break;
            }
            const unsigned char c = buf_[pos_];
            if (inHeader_) {
                pos_++;
                if (c == '\n') inHeader_ = false;
                continue;
            }
            if (c == '>') { pos_++; inHeader_ = true; if (open) { close(out); open = false; } continue; }
            if (c == '\r' || c == '\n') { pos_++; continue; }
            if (!open) {
                // a new record starts: stop BEFORE it if the chunk is full (this byte is seen again next time)
                if ((int64_t)out.residues.size() > limit && out.count() > 0) return false;
                open = true;
            }
            out.residues.push_back(table_[c]);
            pos_++;
        }
        if (open) close(out);
        return true;
    }

private:
    static void close(Records& out) { out.offsets.push_back((int64_t)out.residues.size()); }
    FILE* f_;
    unsigned char table_[256];
    unsigned char buf_[1 << 16];
    size_t pos_ = 0, len_ = 0;
    bool inHeader_ = false;
};

// 50 columns per block: target line, query line, blank line (reference opal_aligner.cpp:303-339)
void print_alignment(const unsigned char* query, const unsigned char* target, const OpalSearchResult& r, const std::string& alphabet) {
    int t = r.startLocationTarget, q = r.startLocationQuery;
    for (int from = 0; from < r.alignmentLength; from += 50) {
        const int to = from + 50 < r.alignmentLength ? from + 50 : r.alignmentLength;
        const int t0 = t, q0 = q;
        fputs("T: ", stdout);
        for (int j = from; j < to; j++) putchar(r.alignment[j] == OPAL_ALIGN_DEL ? '_' : alphabet[target[t++]]);
        printf(" (%d - %d)\n", t0 > 0 ? t0 : 0, t - 1);
        fputs("Q: ", stdout);
        for (int j = from; j < to; j++) putchar(r.alignment[j] == OPAL_ALIGN_INS ? '_' : alphabet[query[q++]]);
        printf(" (%d - %d)\n\n", q0 > 0 ? q0 : 0, q - 1);
    }
}

int usage() {
    fprintf(stderr, "\n");
    fprintf(stderr, "Usage: opal_hip_aligner [options...] <query.fasta> <db.fasta>\n");
    fprintf(stderr, "Options:\n");
    fprintf(stderr, "  -o N  N is gap opening penalty. [default: 3]\n");
    fprintf(stderr, "  -e N  N is gap extension penalty. [default: 1]\n"
                    "    Gap of length n will have penalty of o + (n - 1) * e.\n");
    fprintf(stderr, "  -m Blosum50  Score matrix to be used. [default: Blosum50]\n");
    fprintf(stderr, "  -f FILE  FILE contains score matrix and some additional data. Overrides -m.\n");
    fprintf(stderr, "  -s  If set, there will be no score output (silent mode).\n");
    fprintf(stderr, "  -a SW|NW|HW|OV  Alignment mode that will be used. [default: SW]\n");
    fprintf(stderr, "  -x search_level  Following search levels are available [default: %d]:\n"
                    "    %d - score\n    %d - score, end location\n    %d - score, end and start location and alignment\n",
            OPAL_SEARCH_SCORE, OPAL_SEARCH_SCORE, OPAL_SEARCH_SCORE_END, OPAL_SEARCH_ALIGNMENT);
    return 1;
}

}  // namespace

int main(int argc, char* const argv[]) {
    int gapOpen = 3, gapExt = 1, searchType = OPAL_SEARCH_SCORE;
    std::string mode = "SW", matrixName = "Blosum50", matrixFile;
    bool silent = false;
    for (int opt; (opt = getopt(argc, argv, "a:o:e:m:f:x:s")) >= 0;) {
        switch (opt) {
        case 'a': mode = optarg; break;
        case 'o': gapOpen = atoi(optarg); break;
        case 'e': gapExt = atoi(optarg); break;
        case 'm': matrixName = optarg; break;
        case 'f': matrixFile = optarg; break;
        case 'x': searchType = atoi(optarg); break;
        case 's': silent = true; break;
        default: return usage();
        }
    }
    if (optind + 2 != argc) return usage();

    Scoring sc;
    if (matrixName == "Blosum50") sc = blosum50();
    else { fprintf(stderr, "Given score matrix name is not valid\n"); return 1; }
    if (!matrixFile.empty() && !load_scoring(matrixFile.c_str(), sc)) {
        fprintf(stderr, "Could not read score matrix file %s\n", matrixFile.c_str());
        return 1;
    }
    const int A = (int)sc.alphabet.size();

    int modeCode;
    if (mode == "SW") modeCode = OPAL_MODE_SW;
    else if (mode == "HW") modeCode = OPAL_MODE_HW;
    else if (mode == "NW") modeCode = OPAL_MODE_NW;
    else if (mode == "OV") modeCode = OPAL_MODE_OV;
    else { printf("Invalid mode!\n"); return 1; }
    printf("Using %s alignment mode.\n", mode.c_str());

    FILE* qf = fopen(argv[optind], "r");
    if (!qf) { printf("Error: There is no file with name %s\n", argv[optind]); return 1; }
    printf("Reading query fasta file...\n");
    Records qrec;
    FastaReader(qf, sc.alphabet).read(qrec, INT64_MAX);
    fclose(qf);
    if (qrec.count() == 0) { printf("Error: no sequence in %s\n", argv[optind]); return 1; }
    std::vector<unsigned char> query(qrec.residues.begin() + qrec.offsets[0], qrec.residues.begin() + qrec.offsets[1]);
    const int queryLength = (int)query.size();
    printf("Read query sequence, %d residues.\n", queryLength);

    FILE* df = fopen(argv[optind + 1], "r");
    if (!df) { printf("Error: There is no file with name %s\n", argv[optind + 1]); return 1; }
    FastaReader dbReader(df, sc.alphabet);
    double cpuSeconds = 0, wallSeconds = 0;
    int64_t totalResidues = 0, totalSeqs = 0;
    int status = 0;
    for (bool done = false; !done;) {
        Records rec;
        printf("\nReading database fasta file...\n");
        done = dbReader.read(rec, (int64_t)1 << 30);
        const int n = (int)rec.count();
        std::vector<unsigned char*> db((size_t)n);
        std::vector<int> lens((size_t)n);
        for (int i = 0; i < n; i++) {
            db[(size_t)i] = rec.residues.data() + rec.offsets[(size_t)i];
            lens[(size_t)i] = (int)(rec.offsets[(size_t)i + 1] - rec.offsets[(size_t)i]);
        }
        const int64_t chunkResidues = rec.offsets[(size_t)n];
        printf("Read %d database sequences, %lld residues total.\n", n, (long long)chunkResidues);
        totalResidues += chunkResidues;
        totalSeqs += n;
        if (done)
            printf("Whole database read: %lld database sequences, %lld residues in total.\n", (long long)totalSeqs, (long long)totalResidues);

        std::vector<OpalSearchResult> store((size_t)n);
        std::vector<OpalSearchResult*> results((size_t)n);
        for (int i = 0; i < n; i++) { opalInitSearchResult(&store[(size_t)i]); results[(size_t)i] = &store[(size_t)i]; }
        printf("\nComparing query to database...");
        fflush(stdout);
        const clock_t c0 = clock();
        const auto w0 = std::chrono::steady_clock::now();
        const int rc = opalSearchDatabase(query.data(), queryLength, db.data(), n, lens.data(), gapOpen, gapExt, sc.matrix.data(), A,
                                          results.data(), searchType, modeCode, OPAL_OVERFLOW_BUCKETS);
        wallSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        cpuSeconds += (double)(clock() - c0) / CLOCKS_PER_SEC;
        if (rc) { printf("\nDatabase search failed with error code: %d\n", rc); status = rc; }
        printf("\nFinished!\n");

        if (!silent) {
            printf("\n#<i>: <score> (<query start>, <target start>) (<query end>, <target end>)\n");
            for (int i = 0; i < n; i++) {
                const OpalSearchResult& r = store[(size_t)i];
                printf("#%lld: %d", (long long)(totalSeqs - n + i), r.score);
                if (r.startLocationQuery >= 0) printf(" (%d, %d)", r.startLocationQuery, r.startLocationTarget);
                else printf(" (?, ?)");
                if (r.endLocationQuery >= 0) printf(" (%d, %d)", r.endLocationQuery, r.endLocationTarget);
                else printf(" (?, ?)");
                printf("\n");
                if (r.alignment) print_alignment(query.data(), db[(size_t)i], r, sc.alphabet);
            }
        }
        for (int i = 0; i < n; i++) free(store[(size_t)i].alignment);
    }
    fclose(df);

    printf("\nCpu time of searching: %.2lf\n", cpuSeconds);
    printf("Wall time of searching: %.4lf\n", wallSeconds);
    if (searchType != OPAL_SEARCH_ALIGNMENT && wallSeconds > 0)
        printf("GCUPS (giga cell updates per second): %.2lf\n", (double)totalResidues / 1e9 * queryLength / wallSeconds);
    return status ? 1 : 0;
}
