// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// Thin multi-call front end over the *reference's own* object code (compiled in
// place from /root/reference by oracle/Makefile.ref).  It plays the role that
// src/carpedeam.cpp + lib/mmseqs/src/commons/Application.cpp play in the
// reference: it defines the application globals every MMseqs2-derived tool must
// define (src/carpedeam.cpp:5-13) and a command table restricted to the four
// hot-path modules (src/carpedeam.cpp:31-53, lib/mmseqs/src/MMseqsBase.cpp:562,613).
// Extra "probe" sub-commands call individual reference functions so that
// function-level known answers (damage tables, per-base calls, overlap
// likelihoods, E-values) can be dumped into tests/golden/ by
// tests/golden/make_golden.py.
#include "Command.h"
#include "LocalCommandDeclarations.h"
#include "LocalParameters.h"
#include "nuclassembleUtil.h"
#include "EvalueComputation.h"
#include "NucleotideMatrix.h"
#include "Timer.h"
#include "FileUtil.h"

#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>

// lib/mmseqs/src/CommandDeclarations.h:57,86 (that header also declares `map`, which clashes with libgab's `using namespace std`)
extern int kmermatcher(int argc, const char **argv, const Command &command);
extern int rescorediagonal(int argc, const char **argv, const Command &command);
// lib/mmseqs/src/CommandDeclarations.h:33,29 - the steps either side of the hot path
extern int createdb(int argc, const char **argv, const Command &command);
extern int convert2fasta(int argc, const char **argv, const Command &command);

const char* binary_name = "carpedeam_ref";
const char* tool_name = "CarpeDeam (reference objects, oracle driver)";
const char* tool_introduction = "oracle driver";
const char* main_author = "n/a";
const char* version = "oracle";
const char* show_extended_help = NULL;
const char* show_bash_info = NULL;
bool hide_base_commands = true;
void (*validatorUpdate)(void) = 0;
LocalParameters& localPar = LocalParameters::getLocalInstance();

std::vector<struct Command> commands = {
    {"ancient_read_assemble", ancientReadsResults, &localPar.assembleresults, COMMAND_HIDDEN, "", NULL, "", "<i:sequenceDB> <i:alnResult> <o:reprSeqDB>", 0,
        {{"sequenceDB", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, &DbValidator::nuclDb},
         {"alnResult", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, &DbValidator::alignmentDb},
         {"reprSeqDB", DbType::ACCESS_MODE_OUTPUT, DbType::NEED_DATA, &DbValidator::sequenceDb}}},
    {"ancient_contig_merge", ancientContigsResults, &localPar.assembleresults, COMMAND_HIDDEN, "", NULL, "", "<i:sequenceDB> <i:alnResult> <o:reprSeqDB>", 0,
        {{"sequenceDB", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, &DbValidator::nuclDb},
         {"alnResult", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, &DbValidator::alignmentDb},
         {"reprSeqDB", DbType::ACCESS_MODE_OUTPUT, DbType::NEED_DATA, &DbValidator::sequenceDb}}},
    {"ancient_correction", correction, &localPar.assembleresults, COMMAND_MAIN, "", NULL, "", "<i:sequenceDB> <i:alnResult> <o:reprSeqDB>", 0,
        {{"sequenceDB", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, &DbValidator::nuclDb},
         {"alnResult", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, &DbValidator::alignmentDb},
         {"reprSeqDB", DbType::ACCESS_MODE_OUTPUT, DbType::NEED_DATA, &DbValidator::sequenceDb}}},
};

std::vector<struct Command> baseCommands = {
    {"createdb", createdb, &localPar.createdb, COMMAND_DATABASE_CREATION, "", NULL, "", "<i:fastaFile1[.gz]> ... <o:sequenceDB>", 0,
        {{"fast[a|q]File[.gz|bz2]|stdin", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA | DbType::VARIADIC, &DbValidator::flatfileStdinAndGeneric},
         {"sequenceDB", DbType::ACCESS_MODE_OUTPUT, DbType::NEED_DATA, &DbValidator::flatfile}}},
    {"convert2fasta", convert2fasta, &localPar.convert2fasta, COMMAND_FORMAT_CONVERSION, "", NULL, "", "<i:sequenceDB> <o:fastaFile>", 0,
        {{"sequenceDB", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA | DbType::NEED_HEADER, &DbValidator::allDb},
         {"fastaFile", DbType::ACCESS_MODE_OUTPUT, DbType::NEED_DATA, &DbValidator::flatfile}}},
    {"createhdb", createhdb, &localPar.createhdb, COMMAND_HIDDEN, "", NULL, "", "<i:sequenceDB> [<i:sequenceDBcycle>] <o:headerDB>", 0,
        {{"sequenceDB", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, NULL}}},
    {"cyclecheck", cyclecheck, &localPar.cyclecheck, COMMAND_HIDDEN, "", NULL, "", "<i:sequenceDB> <o:sequenceDBcycle>", 0,
        {{"sequenceDB", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, &DbValidator::nuclDb},
         {"cycleResult", DbType::ACCESS_MODE_OUTPUT, DbType::NEED_DATA, &DbValidator::nuclDb}}},
    {"kmermatcher", kmermatcher, &localPar.kmermatcher, COMMAND_PREFILTER, "", NULL, "", "<i:sequenceDB> <o:prefilterDB>", 0,
        {{"sequenceDB", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, &DbValidator::sequenceDb},
         {"prefilterDB", DbType::ACCESS_MODE_OUTPUT, DbType::NEED_DATA, &DbValidator::prefilterDb}}},
    {"rescorediagonal", rescorediagonal, &localPar.rescorediagonal, COMMAND_ALIGNMENT, "", NULL, "", "<i:queryDB> <i:targetDB> <i:prefilterDB> <o:resultDB>", 0,
        {{"queryDB", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, &DbValidator::sequenceDb},
         {"targetDB", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, &DbValidator::sequenceDb},
         {"resultDB", DbType::ACCESS_MODE_INPUT, DbType::NEED_DATA, &DbValidator::resultDb},
         {"alignmentDB", DbType::ACCESS_MODE_OUTPUT, DbType::NEED_DATA, &DbValidator::alignmentDb}}},
};

// src/assembler/correction.cpp:7 (not declared in any header)
int mostLikeliBaseRead(const int baseInQuery, const unsigned int qIter, const std::vector<countDeamCov> & deamVec,
                       const std::vector<countDeamCov> & countRevs, const std::vector<diNucleotideProb> & subDeamDiNuc,
                       const std::vector<diNucleotideProb> & subDeamDiNucRev, const diNucleotideProb & seqErrMatch,
                       bool wasCorr, unsigned int querySeqLen);
// src/assembler/ancientReadsResults.cpp:48 (not declared in any header)
scorePerRes r_s_pair(Matcher::result_t res, std::string & consensus, char* targetSeq, unsigned int querySeqLen,
                     std::vector<diNucleotideProb> &subDeamDiNuc, std::vector<diNucleotideProb> &subDeamDiNucRev,
                     unsigned int & maxLeft, unsigned int & maxRight, float randAlnPenal, diNucleotideProb & seqErrMatch, float excessPenal);

static void loadDamage(const std::string &prefix, std::vector<diNucleotideProb> &d, std::vector<diNucleotideProb> &drev) {
    std::vector<substitutionRates> sub5p, sub3p;
    d.assign(11, diNucleotideProb());
    initDeamProbabilities(prefix + "5p.prof", prefix + "3p.prof", sub5p, sub3p, d, drev);
}

// probe damage <prefix>: 2 x 11 x 16 long doubles as C99 hex floats
static int probeDamage(int argc, const char **argv) {
    if (argc < 1) return 2;
    std::vector<diNucleotideProb> d, drev;
    loadDamage(argv[0], d, drev);
    for (int r = 0; r < 2; r++)
        for (int i = 0; i < 11; i++) {
            const diNucleotideProb &m = r ? drev[i] : d[i];
            printf("%s %d", r ? "rev" : "fwd", i);
            for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) printf(" %La", m.p[a][b]);
            printf("\n");
        }
    return 0;
}

// probe seqerr <err-as-decimal-string>
static int probeSeqErr(int argc, const char **argv) {
    if (argc < 1) return 2;
    long double err = strtold(argv[0], NULL);
    diNucleotideProb e;
    getSeqErrorProf(e, err);
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) printf("%La%c", e.p[a][b], (a == 3 && b == 3) ? '\n' : ' ');
    return 0;
}

// probe mostlikeli <prefix> : stdin lines
//   qBase qIter qLen wasCorr  88 x count  88 x rev     -> one line "newBase" each
static int probeMostLikeli(int argc, const char **argv) {
    if (argc < 1) return 2;
    std::vector<diNucleotideProb> d, drev;
    loadDamage(argv[0], d, drev);
    diNucleotideProb seqErr;
    long double e = 0.01;
    getSeqErrorProf(seqErr, e);
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        std::istringstream in(line);
        int qBase, wasCorr; unsigned int qIter, qLen;
        in >> qBase >> qIter >> qLen >> wasCorr;
        std::vector<countDeamCov> cnt(qLen), rev(qLen);
        memset(cnt.data(), 0, sizeof(countDeamCov) * qLen);
        memset(rev.data(), 0, sizeof(countDeamCov) * qLen);
        for (int t = 0; t < 4; t++) for (int l = 0; l < 11; l++) in >> cnt[qIter].count[t][l];
        for (int t = 0; t < 4; t++) for (int l = 0; l < 11; l++) in >> rev[qIter].count[t][l];
        int r = mostLikeliBaseRead(qBase, qIter, cnt, rev, d, drev, seqErr, wasCorr != 0, qLen);
        printf("%d\n", r);
    }
    return 0;
}

// probe overlap <prefix> : stdin lines
//   consensus targetSeq qLen dbKey qStart qEnd dbStart dbEnd dbLen alnLen isRev maxLeft maxRight randAlnPenal excessPenal
//   -> "sLenNorm(hex) sRatio(hex)"
static int probeOverlap(int argc, const char **argv) {
    if (argc < 1) return 2;
    std::vector<diNucleotideProb> d, drev;
    loadDamage(argv[0], d, drev);
    diNucleotideProb seqErr;
    long double e = 0.001;
    getSeqErrorProf(seqErr, e);
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        std::istringstream in(line);
        std::string consensus, target;
        unsigned int qLen, dbKey, dbLen, alnLen, maxLeft, maxRight; int qs, qe, ds, de, isRev; float rnd, exc;
        in >> consensus >> target >> qLen >> dbKey >> qs >> qe >> ds >> de >> dbLen >> alnLen >> isRev >> maxLeft >> maxRight >> rnd >> exc;
        Matcher::result_t res(dbKey, 0, 0, 0, 0, 0, alnLen, qs, qe, qLen, ds, de, dbLen, "");
        res.isRevToAlignment = isRev != 0;
        scorePerRes s = r_s_pair(res, consensus, (char *) target.c_str(), qLen, d, drev, maxLeft, maxRight, rnd, seqErr, exc);
        printf("%a %a\n", s.sLenNorm, s.sRatio);
    }
    return 0;
}

// probe evalue <dbResidues> <path of nucleotide.out> : stdin lines "score qLen" -> "evalue(hex) bitscore(hex) %.3E"
static int probeEvalue(int argc, const char **argv) {
    if (argc < 1) return 2;
    if (argc < 2) return 2;
    size_t dbRes = strtoull(argv[0], NULL, 10);
    NucleotideMatrix subMat(argv[1], 1.0, 0.0);
    EvalueComputation evaluer(dbRes, &subMat);
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        std::istringstream in(line);
        int score, qLen;
        in >> score >> qLen;
        double ev = evaluer.computeEvalue(score, qLen);
        double bs = evaluer.computeBitScore(score);
        printf("%a %a %.3E %a\n", ev, bs, ev, evaluer.computeRawScoreFromBitScore(static_cast<int>(bs + 0.5)));
    }
    return 0;
}

// probe alp : the gapless Gumbel parameters ALP derives for nucleotide.out, obtained exactly the way
// EvalueComputation::init does for an ungapped matrix (M/alignment/EvalueComputation.h:96-128)
static int probeAlp(int argc, const char **argv) {
    if (argc < 1) return 2;
    NucleotideMatrix subMat(argv[0], 1.0, 0.0);  // path of lib/mmseqs/data/nucleotide.out
    long **tmpMat = new long *[subMat.alphabetSize];
    long *tmpMatData = new long[subMat.alphabetSize * subMat.alphabetSize];
    for (int i = 0; i < subMat.alphabetSize; i++) {
        tmpMat[i] = &tmpMatData[i * subMat.alphabetSize];
        for (int j = 0; j < subMat.alphabetSize; j++) tmpMat[i][j] = subMat.subMatrix[i][j];
    }
    Sls::AlignmentEvaluer ev;
    // probe alp <matrix> <gapOpen> <gapExtend>: the GAPPED parameters, obtained exactly the way EvalueComputation::init does for gap
    // costs it has no table entry for (M/alignment/EvalueComputation.h:96-110: tolerances 0.01 / 0.05, 60 s, 500 MB, seed 42)
    if (argc >= 3) ev.initGapped(subMat.alphabetSize - 1, (const long *const *) tmpMat, subMat.pBack, subMat.pBack, atoi(argv[1]), atoi(argv[2]), atoi(argv[1]), atoi(argv[2]),
                                 false, 0.01, 0.05, 60.0, 500, 42);
    else
    ev.initGapless(subMat.alphabetSize - 1, (const long *const *) tmpMat, subMat.pBack, subMat.pBack, 60.0);
    const Sls::ALP_set_of_parameters &p = ev.parameters();
    printf("lambda %a\nK %a\na_I %a\na_J %a\nalpha_I %a\nalpha_J %a\nsigma %a\nb_I %a\nb_J %a\nbeta_I %a\nbeta_J %a\ntau %a\n",
           p.lambda, p.K, p.a_I, p.a_J, p.alpha_I, p.alpha_J, p.sigma, p.b_I, p.b_J, p.beta_I, p.beta_J, p.tau);
    printf("vi_y_thr %a\nvj_y_thr %a\nc_y_thr %a\n", p.vi_y_thr, p.vj_y_thr, p.c_y_thr);
    printf("matrix");
    for (int i = 0; i < subMat.alphabetSize; i++) for (int j = 0; j < subMat.alphabetSize; j++) printf(" %d", (int) subMat.subMatrix[i][j]);
    printf("\npBack");
    for (int i = 0; i < subMat.alphabetSize; i++) printf(" %a", subMat.pBack[i]);
    printf("\n");
    return 0;
}

static Command *find(const char *s) {
    for (size_t i = 0; i < commands.size(); i++) if (!strcmp(s, commands[i].cmd)) return &commands[i];
    for (size_t i = 0; i < baseCommands.size(); i++) if (!strcmp(s, baseCommands[i].cmd)) return &baseCommands[i];
    return NULL;
}

int main(int argc, const char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: carpedeam_ref <module|probe> ...\n"); return 2; }
    if (!strcmp(argv[1], "probe")) {
        if (argc < 3) return 2;
        if (!strcmp(argv[2], "damage")) return probeDamage(argc - 3, argv + 3);
        if (!strcmp(argv[2], "seqerr")) return probeSeqErr(argc - 3, argv + 3);
        if (!strcmp(argv[2], "mostlikeli")) return probeMostLikeli(argc - 3, argv + 3);
        if (!strcmp(argv[2], "overlap")) return probeOverlap(argc - 3, argv + 3);
        if (!strcmp(argv[2], "evalue")) return probeEvalue(argc - 3, argv + 3);
        if (!strcmp(argv[2], "alp")) return probeAlp(argc - 3, argv + 3);
        return 2;
    }
    FileUtil::fixRlimitNoFile();
    setenv("MMSEQS", argv[0], true);
    Command *c = find(argv[1]);
    if (c == NULL) { fprintf(stderr, "unknown module %s\n", argv[1]); return 2; }
    Timer timer;
    int status = c->commandFunction(argc - 2, argv + 2, *c);
    Debug(Debug::INFO) << "Time for processing: " << timer.lap() << "\n";
    EXIT(status);
}
